"""ORACLE side of the plugin models of examples/plugins/models.py: the same discrete updates as float-or-dual Python
functions step(x, u, p, dt) -> list, for oracle.models_np.Model.custom (test infrastructure; examples/ never imports oracle/)."""
from oracle import dual as D


def chain_step(nq):
    def step(x, u, p, dt):
        ks, c, kc = p[0], p[1], p[2]
        out = [None] * (2 * nq)
        for i in range(nq):
            qi, vi = x[i], x[nq + i]
            a = -ks * D.sin(qi) - c * vi
            if i < nq - 1:
                a = a + kc * D.sin(x[i + 1] - qi)
            if i > 0:
                a = a - kc * D.sin(qi - x[i - 1])
            if i >= nq - 12:
                a = a + u[i - (nq - 12)]
            vn = vi + dt * a
            out[nq + i] = vn
            out[i] = qi + dt * vn
        return out
    return step


def chainx_step(nq, m, ne):
    def step(x, u, p, dt):
        ks, c, kc, ae = p[0], p[1], p[2], p[3]
        out = [None] * (2 * nq + ne)
        for i in range(nq):
            qi, vi = x[i], x[nq + i]
            a = -ks * D.sin(qi) - c * vi
            if i < nq - 1:
                a = a + kc * D.sin(x[i + 1] - qi)
            if i > 0:
                a = a - kc * D.sin(qi - x[i - 1])
            if i >= nq - m:
                a = a + u[i - (nq - m)]
            vn = vi + dt * a
            out[nq + i] = vn
            out[i] = qi + dt * vn
        for j in range(ne):
            e = x[2 * nq + j]
            out[2 * nq + j] = e + dt * (D.sin(x[j % nq]) - ae * e + u[j % m])
        return out
    return step


def vdp_step(x, u, p, dt):
    q, v = x[0], x[1]
    a = p[0] * (1.0 - q * q) * v - q + u[0]
    vn = v + dt * a
    return [q + dt * vn, vn]


def chain3_step(x, u, p, dt):
    ks, c, kc = p[0], p[1], p[2]
    l01, l12 = D.sin(x[1] - x[0]), D.sin(x[2] - x[1])
    a0 = -ks * D.sin(x[0]) - c * x[3] + kc * l01 + u[0]
    a1 = -ks * D.sin(x[1]) - c * x[4] + kc * l12 - kc * l01
    a2 = -ks * D.sin(x[2]) - c * x[5] - kc * l12 + u[1]
    v0, v1, v2 = x[3] + dt * a0, x[4] + dt * a1, x[5] + dt * a2
    return [x[0] + dt * v0, x[1] + dt * v1, x[2] + dt * v2, v0, v1, v2]


def kink2_step(x, u, p, dt):
    c, k = p[0], p[1]
    q, v = x[0], x[1]
    a = u[0] - c * v - 2.0 * D.sin(q)
    qv = q.v if isinstance(q, D.Dual) else q
    if qv < 0.0:
        a = a - k * q
    vn = v + dt * a
    return [q + dt * vn, vn]
