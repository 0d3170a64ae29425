"""Forward-mode dual numbers for the ORACLE (test infrastructure only).

Stands in for Drake's AutoDiffXd on the CPU side: the reference obtains exact
dynamics Jacobians by seeding (x,u) with unit derivatives
(/root/reference/ilqr.py:253-256 ``InitializeAutoDiff``), pushing them through
the discrete update (:259-265) and reading the gradient back (:268-270
``ExtractGradient``).  ``Dual`` reproduces that arithmetic (value + gradient
row, product/chain rule per primitive) so the oracle's ``fx``/``fu`` are exact
to round-off, like AutoDiff.

Nothing in the product path (drake_ddp_amd/) may import this file.
"""
import numpy as np

# numpy ufuncs (not math.*): sin(inf)=nan and exp(1e3)=inf instead of raising, so a
# diverging line-search rollout yields L=nan/inf and is rejected.
np.seterr(all="ignore")


class math:  # noqa: N801 - local shim with IEEE (non-raising) semantics
    sin = staticmethod(lambda a: float(np.sin(a)))
    cos = staticmethod(lambda a: float(np.cos(a)))
    exp = staticmethod(lambda a: float(np.exp(a)))
    log1p = staticmethod(lambda a: float(np.log1p(a)))
    sqrt = staticmethod(lambda a: float(np.sqrt(a)))


class Dual:
    """Scalar value ``v`` carrying a gradient row ``d`` (numpy 1-D array)."""

    __slots__ = ("v", "d")
    __array_priority__ = 1000  # make numpy defer to our reflected operators

    def __init__(self, v, d):
        self.v = float(v)
        self.d = d

    # -- helpers ---------------------------------------------------------
    @staticmethod
    def _lift(o, like):
        if isinstance(o, Dual):
            return o
        return Dual(o, np.zeros_like(like.d))

    # -- arithmetic ------------------------------------------------------
    def __add__(self, o):
        o = Dual._lift(o, self)
        return Dual(self.v + o.v, self.d + o.d)

    __radd__ = __add__

    def __sub__(self, o):
        o = Dual._lift(o, self)
        return Dual(self.v - o.v, self.d - o.d)

    def __rsub__(self, o):
        o = Dual._lift(o, self)
        return Dual(o.v - self.v, o.d - self.d)

    def __mul__(self, o):
        o = Dual._lift(o, self)
        return Dual(self.v * o.v, self.d * o.v + o.d * self.v)

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = Dual._lift(o, self)
        q = self.v / o.v
        return Dual(q, (self.d - q * o.d) / o.v)

    def __rtruediv__(self, o):
        o = Dual._lift(o, self)
        return o.__truediv__(self)

    def __neg__(self):
        return Dual(-self.v, -self.d)

    def __pos__(self):
        return self

    def __repr__(self):
        return f"Dual({self.v!r}, {self.d!r})"


def seed(values):
    """Unit-seeded duals for a flat vector (the InitializeAutoDiff analogue)."""
    values = np.asarray(values, dtype=float).ravel()
    k = values.size
    eye = np.eye(k)
    return [Dual(values[i], eye[i].copy()) for i in range(k)]


def gradient(duals):
    """Stack gradient rows (the ExtractGradient analogue) -> (len, k)."""
    return np.vstack([q.d for q in duals])


# -- primitives that work on float or Dual ------------------------------

def sin(a):
    if isinstance(a, Dual):
        return Dual(math.sin(a.v), math.cos(a.v) * a.d)
    return math.sin(a)


def cos(a):
    if isinstance(a, Dual):
        return Dual(math.cos(a.v), -math.sin(a.v) * a.d)
    return math.cos(a)


def exp(a):
    if isinstance(a, Dual):
        e = math.exp(a.v)
        return Dual(e, e * a.d)
    return math.exp(a)


def log1p(a):
    if isinstance(a, Dual):
        return Dual(math.log1p(a.v), a.d / (1.0 + a.v))
    return math.log1p(a)


def sqrt(a):
    if isinstance(a, Dual):
        r = math.sqrt(a.v)
        return Dual(r, a.d / (2.0 * r))
    return math.sqrt(a)


def softplus(z):
    """log(1+exp(z)) in the overflow-safe form max(z,0)+log1p(exp(-|z|)).

    d/dz = logistic(z).  Same branch structure as the device code
    (drake_ddp_amd/csrc/models.hpp: mi_softplus)."""
    zv = z.v if isinstance(z, Dual) else z
    if zv > 0.0:
        return z + log1p(exp(-z))
    return log1p(exp(z))
