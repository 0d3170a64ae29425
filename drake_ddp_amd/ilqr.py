"""Host-side mirror of the reference solver class over the C ABI.

``IterativeLinearQuadraticRegulator`` keeps the class surface of
/root/reference/ilqr.py:12-733 (same constructor signature, setters, ``Solve``
return tuple, ``SaveSolution`` npz keys, attribute names and shapes) so the
solver section of any example script can import it as a drop-in; the `system`
argument is a model descriptor (drake_ddp_amd.models.ModelSystem).  All compute
is in libmi_ilqr.so (HIP, gfx950) — this file only moves arrays across the
boundary.  ``BatchedIterativeLQR`` is the same surface with a leading batch axis.
"""
import ctypes as C
import time
import warnings
import weakref

import numpy as np

from . import _capi
from . import utils_derivs_interpolation
from .models import ModelSystem

_KP_IDS = {"setInterval": _capi.KP_SET_INTERVAL, "adaptiveJerk": _capi.KP_ADAPTIVE_JERK,
           "iterativeError": _capi.KP_ITERATIVE_ERROR}
_JAC_IDS = {"fd": _capi.JAC_FD_CENTRAL, "fd_central": _capi.JAC_FD_CENTRAL,
            "autodiff": _capi.JAC_AUTODIFF, "ad": _capi.JAC_AUTODIFF}


from .dist import shard_range, allreduce_min, allreduce_min_async  # noqa: E402,F401


class _PinnedBlock:
    """One page-locked allocation of a solver's result pool and whether an array handed out over it is still alive."""
    __slots__ = ("raw", "addr", "ctype", "count", "busy", "ref", "__weakref__")

    def __init__(self, raw, nbytes, count):
        self.raw, self.addr, self.count, self.busy, self.ref = raw, raw.value, count, False, None
        self.ctype = C.c_char * nbytes

    def _released(self, _ref):
        self.busy = False


class BatchedIterativeLQR:
    """B independent iLQR problems sharing model, horizon and cost, solved on one GPU.

    Same method names as the reference class; array arguments/attributes carry a
    leading batch axis: x0 (B,n), u_guess (B,m,N-1) [or (m,N-1), broadcast],
    x_bar (B,n,N), u_bar (B,m,N-1), K (B,m,n,N-1), kappa (B,m,N-1) ...
    """

    def __init__(self, system, num_timesteps, batch, input_port_index=0, delta=1e-2, beta=0.95, gamma=0.0,
                 derivs_keypoint_method=None, jacobian_mode="fd", fd_step=1e-5, device=0,
                 max_iters=1000, hist_cap=64, kernel_mode="auto", pinned_results=True, on_indefinite="stop"):
        assert isinstance(system, ModelSystem), \
            "system must be a drake_ddp_amd.models.ModelSystem (Drake systems cannot run on the GPU)"
        assert system.IsDifferenceEquationSystem()[0], "must be a discrete-time system"   # ilqr.py:37
        self._lib = _capi.load()
        self.system = system
        self.N = int(num_timesteps)
        self.B = int(batch)
        self.delta, self.beta, self.gamma = delta, beta, gamma
        self.n, self.m = system.n, system.m
        # device-side number of controls: more than m when the model carries padding controls (n > 32 with m % 4 != 0 on the
        # workgroup-per-problem kernels, plugin.device_controls).  Everything a caller passes or receives has m controls;
        # R gets a unit block, guesses zeros, results are cut back - the padding's gains and inputs are exact zeros.
        self._md = int(getattr(system, "m_dev", system.m))
        # ilqr.py:97-100: default = derivatives at every time step
        if derivs_keypoint_method is None:
            derivs_keypoint_method = utils_derivs_interpolation.derivs_interpolation('setInterval', 1, 0, 0, 0)
        self.derivs_interpolation = derivs_keypoint_method
        if derivs_keypoint_method.keypoint_method not in _KP_IDS:
            raise Exception('unknown interpolation method')                               # ilqr.py:404
        d = _capi.Desc()
        d.n, d.m, d.N, d.B = self.n, self._md, self.N, self.B
        d.model_id = system.model_id
        d.n_params = system.params.size
        for i, v in enumerate(system.params):
            d.model_params[i] = float(v)
        d.dt = system.dt
        d.delta, d.beta, d.gamma = float(delta), float(beta), float(gamma)
        d.keypoint_method = _KP_IDS[derivs_keypoint_method.keypoint_method]
        d.minN, d.maxN = int(derivs_keypoint_method.minN), int(derivs_keypoint_method.maxN)
        d.jerk_threshold = float(derivs_keypoint_method.jerk_threshold)
        d.iterative_error_threshold = float(derivs_keypoint_method.iterative_error_threshold)
        d.jacobian_mode = _JAC_IDS[jacobian_mode]
        d.fd_step = float(fd_step)
        d.max_iters, d.hist_cap, d.device_id = int(max_iters), int(hist_cap), int(device)
        d.kernel_mode = {"auto": _capi.KERNEL_AUTO, "latency": _capi.KERNEL_LATENCY,
                         "throughput": _capi.KERNEL_THROUGHPUT}[kernel_mode]
        # a Quu that is not positive definite (workgroup-per-problem kernels): "stop" the problem with STATUS_NOT_PD (this class's
        # default: one such problem does not spoil a batch - it is reported, never raised), or "continue" like the reference, which
        # inverts whatever comes out with np.linalg.inv and carries on (ilqr.py:655; the single-problem drop-in's default)
        d.on_indefinite = {"stop": 0, "continue": 1}[on_indefinite]
        self.on_indefinite = on_indefinite
        self._desc = d
        self.hist_cap = int(hist_cap)
        h = C.c_void_p()
        _capi.check(self._lib.mi_ilqr_create(C.byref(d), C.byref(h)), "mi_ilqr_create")
        self._h = h
        # pinned_results (the default): the arrays the state attributes / Solve() return are views of page-locked buffers
        # (direct DMA, no page faults of freshly allocated arrays; the wave-per-problem kernels write x_bar / u_bar / cost
        # into them themselves).  The reference REBINDS its result arrays on every forward pass and never mutates one it
        # has handed out (ilqr.py:375-376) - so does this class: a buffer is reused only once the caller has
        # dropped every reference to the array it got (a small pool per attribute; `x, u, t, L = ilqr.Solve()` in a loop
        # alternates between two blocks), otherwise a new block is taken.  False: plain pageable arrays and blocking copies.
        self._pinned = {} if pinned_results else None
        self._into = None
        self._result_fields = ((_capi.F_X_BAR, (self.B, self.n, self.N)), (_capi.F_U_BAR, (self.B, self._md, self.N - 1)), (_capi.F_COST, (self.B,)))
        self._sink = None              # pinned_results: whether the kernels of this handle can write the results into host buffers themselves
        # reference defaults (ilqr.py:61-67); NOTE x_nom is undefined until SetTargetState (F12)
        self.x0 = np.zeros((self.B, self.n))
        self.Q, self.R, self.Qf = np.eye(self.n), np.eye(self.m), np.eye(self.n)
        self._u_guess = None
        self.stats = None
        self.time_getDerivs = self.time_backwardsPass = self.time_fp = 0.0
        self.solve_wall_s = 0.0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.mi_ilqr_destroy(h)           # (synchronizes the stream: no kernel writes a result sink afterwards)

    # ------------------------------------------------------------- setters (ilqr.py:102-159)
    def SetInitialState(self, x0):
        self.x0 = x0

    def SetTargetState(self, x_nom):
        self.x_nom = np.asarray(x_nom).reshape((self.n,))

    def SetRunningCost(self, Q, R):
        assert Q.shape == (self.n, self.n)
        assert R.shape == (self.m, self.m)
        self.Q = Q
        self.R = R

    def SetTerminalCost(self, Qf):
        assert Qf.shape == (self.n, self.n)
        self.Qf = Qf

    def SetInitialGuess(self, u_guess):
        assert u_guess.shape in ((self.m, self.N - 1), (self.B, self.m, self.N - 1))
        self._u_guess = u_guess        # aliased like the reference; copied in at Solve()

    def SetControlLimits(self, u_min, u_max):
        pass                           # no-op stub in the reference too (ilqr.py:158-159)

    # ------------------------------------------------------------- boundary traffic
    def _push_problem(self):
        x_nom = self.x_nom             # AttributeError if SetTargetState was never called, as in the reference
        Q, R, Qf = (_capi.as_f64(a) for a in (self.Q, self._pad_u(self.R, (0, 1), diag=1.0), self.Qf))
        xn = _capi.as_f64(x_nom, (self.n,))
        _capi.check(self._lib.mi_ilqr_set_cost(self._h, _capi.ptr(Q), _capi.ptr(R), _capi.ptr(Qf), _capi.ptr(xn)),
                    "mi_ilqr_set_cost")
        x0 = np.asarray(self.x0, dtype=np.float64).reshape(-1, self.n)
        x0 = np.ascontiguousarray(x0 if len(x0) == self.B else np.broadcast_to(x0, (self.B, self.n)))
        ug = None
        shared = False
        if self._u_guess is not None:
            ug = np.asarray(self._u_guess, dtype=np.float64)
            # one sequence for the whole batch - (m,N-1) or (1,m,N-1), the reference's own argument - crosses the
            # bus once; the device writes the batch's copies
            if ug.shape not in ((self.m, self.N - 1), (1, self.m, self.N - 1), (self.B, self.m, self.N - 1)):
                raise AssertionError(f"initial guess must be (m,N-1) or (B,m,N-1), got {ug.shape}")
            shared = ug.ndim == 2 or (ug.shape[0] == 1 and self.B > 1)
            if shared:
                ug = np.ascontiguousarray(self._pad_u(ug.reshape(self.m, self.N - 1), (0,)))
            else:
                ug = np.ascontiguousarray(self._pad_u(np.broadcast_to(ug, (self.B, self.m, self.N - 1)), (1,)))
            self._u_guess = None       # u_bar is rebound by the forward pass (ilqr.py:375)
        if shared:
            _capi.check(self._lib.mi_ilqr_set_initial_shared(self._h, _capi.ptr(x0), _capi.ptr(ug)), "mi_ilqr_set_initial_shared")
        else:
            _capi.check(self._lib.mi_ilqr_set_initial(self._h, _capi.ptr(x0), _capi.ptr(ug)), "mi_ilqr_set_initial")

    _POOL_CAP = 4

    def _out(self, which, shape, dtype):
        return self._out_addr(which, shape, dtype)[0]

    def _out_addr(self, which, shape, dtype):
        """Destination of a field read and its address: a fresh array, or (pinned_results) a page-locked block nobody else holds.

        Every hand-out wraps the block in a NEW owner array (np.frombuffer); every view a caller derives from what it got has
        that owner as its base, so the owner dies exactly when the last of them does - a weak reference's callback then marks
        the block free.  No reference counts are read (a debugger, a profiler or another interpreter may hold extra ones): a
        block whose owner has not been collected yet simply is not reused."""
        if self._pinned is None:
            a = np.empty(shape, dtype=dtype)
            return a, _capi.ptr(a)
        key = (which, shape if type(shape) is tuple else tuple(shape), dtype)
        pool = self._pinned.get(key)
        if pool is None:
            pool = self._pinned[key] = []
        blk = None
        for b_ in pool:
            if not b_.busy:
                blk = b_
                break
        if blk is None:
            if len(pool) >= self._POOL_CAP:         # the caller keeps many results alive: those stay theirs, this one is pageable
                a = np.empty(shape, dtype=dtype)
                return a, _capi.ptr(a)
            count = int(np.prod(shape))
            nbytes = max(count * np.dtype(dtype).itemsize, 8)
            raw = C.c_void_p()
            _capi.check(self._lib.mi_ilqr_host_alloc(nbytes, C.byref(raw)), "mi_ilqr_host_alloc")
            blk = _PinnedBlock(raw, nbytes, count)
            # the memory lives as long as the block object does: the pool's reference, or a buffer a caller's array still wraps
            weakref.finalize(blk, self._lib.mi_ilqr_host_free, raw).atexit = False
            pool.append(blk)
        buf = blk.ctype.from_address(blk.addr)
        buf._block = blk                            # (an array over the buffer keeps the buffer, the buffer keeps the block - and its memory)
        owner = np.frombuffer(buf, dtype=dtype, count=blk.count)
        blk.busy = True
        blk.ref = weakref.ref(owner, blk._released)
        return owner.reshape(shape), blk.addr

    # axis (from the end) along which a field carries the controls
    _U_AXIS = {_capi.F_U_BAR: -2, _capi.F_KAPPA: -2, _capi.F_U_TRIAL: -2, _capi.F_K: -3, _capi.F_FU: -2}

    def _pad_u(self, a, axes, diag=0.0):
        """`a` with its control axes grown from m to the device's count (zeros; `diag` on the new diagonal of a square block)."""
        if self._md == self.m:
            return a
        a = np.asarray(a, dtype=np.float64)
        shape = list(a.shape)
        for ax in axes:
            shape[ax] = self._md
        out = np.zeros(shape)
        out[tuple(slice(0, self.m) if i in [ax % a.ndim for ax in axes] else slice(None) for i in range(a.ndim))] = a
        if diag and len(axes) == 2:
            for k in range(self.m, self._md):
                out[k, k] = diag
        return out

    def _dev_shape(self, which, shape):
        if self._md == self.m or which not in self._U_AXIS:
            return tuple(shape), None
        ax = len(shape) + self._U_AXIS[which]
        dev = list(shape)
        dev[ax] = self._md
        return tuple(dev), ax

    def _cut_u(self, out, ax):
        return out if ax is None else out[tuple(slice(0, self.m) if i == ax else slice(None) for i in range(out.ndim))]

    def _get(self, which, shape):
        dev, ax = self._dev_shape(which, shape)
        out = self._out(which, dev, np.float64)
        _capi.check(self._lib.mi_ilqr_get(self._h, which, _capi.ptr(out), out.nbytes), "mi_ilqr_get")
        return self._cut_u(out, ax)

    def _get_int(self, which, shape):
        out = self._out(which, shape, np.int32)
        _capi.check(self._lib.mi_ilqr_get_int(self._h, which, _capi.ptr(out), out.nbytes), "mi_ilqr_get_int")
        return out

    def _set(self, which, arr, shape):
        dev, ax = self._dev_shape(which, shape)
        if ax is not None:
            arr, shape = self._pad_u(np.asarray(arr, dtype=np.float64).reshape(shape), (ax,)), dev
        arr = _capi.as_f64(arr, shape)
        _capi.check(self._lib.mi_ilqr_set(self._h, which, _capi.ptr(arr), arr.nbytes), "mi_ilqr_set")

    # state attributes with the reference's names (batched)
    x_bar = property(lambda s: s._get(_capi.F_X_BAR, (s.B, s.n, s.N)))
    u_bar = property(lambda s: s._get(_capi.F_U_BAR, (s.B, s.m, s.N - 1)))
    K = property(lambda s: s._get(_capi.F_K, (s.B, s.m, s.n, s.N - 1)))
    kappa = property(lambda s: s._get(_capi.F_KAPPA, (s.B, s.m, s.N - 1)))
    dV_coeff = property(lambda s: s._get(_capi.F_DV, (s.B, s.N - 1)))
    fx = property(lambda s: s._get(_capi.F_FX, (s.B, s.n, s.n, s.N - 1)))
    fu = property(lambda s: s._get(_capi.F_FU, (s.B, s.n, s.m, s.N - 1)))
    cost = property(lambda s: s._get(_capi.F_COST, (s.B,)))
    history = property(lambda s: s._get(_capi.F_HIST, (s.B, s.hist_cap, 4)))
    iteration_cycles = property(lambda s: s._get(_capi.F_ITER_CYCLES, (s.B, s.hist_cap, 4)))
    iterations = property(lambda s: s._get_int(_capi.I_ITERS, (s.B,)))
    status = property(lambda s: s._get_int(_capi.I_STATUS, (s.B,)))
    ls_trials = property(lambda s: s._get_int(_capi.I_LS_TRIALS, (s.B,)))
    keypoint_count = property(lambda s: s._get_int(_capi.I_KP_COUNT, (s.B,)))
    keypoint_list = property(lambda s: s._get_int(_capi.I_KP_LIST, (s.B, s.N - 1)))

    @property
    def percentage_derivs(self):
        return self.keypoint_count / (self.N - 1) * 100.0          # ilqr.py:406

    def set_state(self, **fields):
        """Overwrite persistent state arrays (stage-level tests / checkpoint restore)."""
        table = {"x_bar": (_capi.F_X_BAR, (self.B, self.n, self.N)), "u_bar": (_capi.F_U_BAR, (self.B, self.m, self.N - 1)),
                 "K": (_capi.F_K, (self.B, self.m, self.n, self.N - 1)), "kappa": (_capi.F_KAPPA, (self.B, self.m, self.N - 1)),
                 "dV_coeff": (_capi.F_DV, (self.B, self.N - 1)), "fx": (_capi.F_FX, (self.B, self.n, self.n, self.N - 1)),
                 "fu": (_capi.F_FU, (self.B, self.n, self.m, self.N - 1))}
        for k, v in fields.items():
            self._set(table[k][0], v, table[k][1])

    @property
    def stage_cycles(self):
        """(B,4) in-kernel shader-clock cycles of the last solve: line search, linearization,
        backward pass, whole loop (device counterpart of time_fp/time_getDerivs/time_backwardsPass)."""
        out = np.empty((self.B, 4), dtype=np.int64)
        _capi.check(self._lib.mi_ilqr_get_cycles(self._h, _capi.ptr(out), out.nbytes), "mi_ilqr_get_cycles")
        return out

    @property
    def cluster_stats(self):
        """Workgroup-per-problem kernels, diagnostic: per problem of the last launch - helpers that took part, linearizations shared
        with them the regular way, rounds in which every helper sat on the leader's own XCD, early rounds opened (the helpers
        linearize the line search's first trial while it is rolled out), early rounds whose trial was accepted, candidate-group rounds
        (mid-size kernels: the helpers roll out line-search candidates 4 .. beside the leader's four) - (B,6) int64;
        zeros when the launch was not clustered (mi_ilqr.h: MI_I64_CLUSTER_WORDS)."""
        w = np.empty((self.B, _capi.CLUSTER_WORDS), dtype=np.uint64)
        _capi.check(self._lib.mi_ilqr_get_int(self._h, _capi.I64_CLUSTER_WORDS, _capi.ptr(w), w.nbytes), "mi_ilqr_get_int")
        u = np.uint64
        ea_open, ea_hit, groups = w[:, 5] >> u(32), w[:, 5] & u(0xffffffff), w[:, 7]
        return np.stack([w[:, 2] & u(0xffff), (w[:, 3] >> u(32)) - ea_open - groups, (w[:, 3] >> u(8)) & u(0xffffff), ea_open, ea_hit, groups], axis=1).astype(np.int64)

    def last_kernel_ms(self):
        ms = C.c_float()
        _capi.check(self._lib.mi_ilqr_last_kernel_ms(self._h, C.byref(ms)), "mi_ilqr_last_kernel_ms")
        return float(ms.value)

    def set_timing(self, every=1):
        """HIP events on one pipelined solve in `every` (1 = all, the default; 0 = none): see mi_ilqr_set_timing."""
        _capi.check(self._lib.mi_ilqr_set_timing(self._h, int(every)), "mi_ilqr_set_timing")

    def Reset(self):
        """Forget warm-start state: equivalent to constructing a new solver (ilqr.py:70-83)."""
        _capi.check(self._lib.mi_ilqr_reset(self._h), "mi_ilqr_reset")

    def _check_internal(self, stats):
        """A solve aborted inside the kernel (MI_STATUS_INTERNAL: a cluster helper stopped answering) left x_bar / u_bar
        that are not a solution: never hand them out as one.  Problems that met a Quu which is not positive definite do NOT
        abort the batch - the other B - 1 results are valid: they are reported per problem (`status`: STATUS_NOT_PD = stopped
        there with on_indefinite="stop", STATUS_FLAG_INDEFINITE OR-ed onto the outcome with "continue"), counted in
        stats.n_not_pd, and announced by a warning."""
        if stats is not None and stats.n_internal > 0:
            raise RuntimeError(f"{stats.n_internal} problem(s) aborted inside the device kernel (status {_capi.STATUS_INTERNAL}); "
                               "their results are not a solution")
        if stats is not None and stats.n_max_iters > 0:
            # (the reference's loop has no cap, ilqr.py:692: a problem stopped by this class's `max_iters` is not a converged one)
            warnings.warn(f"{stats.n_max_iters} of {self.B} problem(s) stopped at max_iters = {int(self._desc.max_iters)} with the improvement "
                          f"still above delta (status {_capi.STATUS_MAX_ITERS}): not converged", RuntimeWarning, stacklevel=3)
        if stats is not None and stats.n_not_pd > 0:
            how = ("stopped there (status %d): their gains are not to be used" % _capi.STATUS_NOT_PD if self.on_indefinite == "stop"
                   else "inverted it like the reference's np.linalg.inv (ilqr.py:655) and carried on (status flag %d)" % _capi.STATUS_FLAG_INDEFINITE)
            warnings.warn(f"{stats.n_not_pd} of {self.B} problem(s) met a Quu that is not positive definite in a backward pass and {how}",
                          RuntimeWarning, stacklevel=3)

    # ------------------------------------------------------------- Solve (ilqr.py:669-710)
    def Solve(self):
        st = time.time()
        self._push_problem()
        if self._pinned is not None:
            res = self._solve_into_pinned()
            self._check_internal(self.stats)
            self.solve_wall_s = time.time() - st
            return res[0], res[1], self.solve_wall_s, res[2]
        stats = _capi.Stats()
        _capi.check(self._lib.mi_ilqr_solve(self._h, C.byref(stats)), "mi_ilqr_solve")
        self.stats = stats
        self._check_internal(stats)
        self.solve_wall_s = time.time() - st
        return self.x_bar, self.u_bar, self.solve_wall_s, self.cost

    def _solve_into_pinned(self, extra=()):
        """One solve from the inputs just pushed, x_bar / u_bar / cost into page-locked arrays nobody else holds, ONE host
        synchronization.  Wave-per-problem kernels write them themselves as each problem finishes (mi_ilqr_set_result_sink:
        the copy-out overlaps the launch's stragglers) - the sink is set for THIS solve only, so no later kernel of the
        handle (pipelined solves, MPCRun, stage calls) touches arrays a caller holds; the other kernel families enqueue
        three copy-outs behind the solve.  `extra`: (field, destination, its address) triples copied out behind the solve as
        well.  ONE call across the boundary (mi_ilqr_solve_into)."""
        (x, ax), (u, au), (c, ac) = (self._out_addr(which, shp, np.float64) for which, shp in self._result_fields)
        res = [x, u, c]
        pinned = self._sink is not False and x.base is not None and u.base is not None and c.base is not None
        k = len(extra)
        if self._into is None or len(self._into[0]) != k:
            self._into = ((C.c_int32 * k)(), (C.c_void_p * k)(), (C.c_size_t * k)())
        wh, ds, by = self._into
        for i, (which, out, addr) in enumerate(extra):     # further fields of this solve, behind it on the stream
            wh[i], ds[i], by[i] = which, addr, out.nbytes
        stats, used = _capi.Stats(), C.c_int32()
        rc = self._lib.mi_ilqr_solve_into(self._h, ax, au, ac, 1 if pinned else 0, k, wh, ds, by, C.byref(stats), C.byref(used))
        if rc != _capi.OK:
            _capi.check(rc, "mi_ilqr_solve_into")
        self.stats = stats
        if pinned:
            self._sink = bool(used.value)
        if self._md != self.m:
            res[1] = res[1][:, :self.m, :]
        return res

    def solve_resident(self):
        """Solve again from the inputs already resident on the device (no host traffic
        except the aggregate stats): used by bench.py and the on-device MPC loop."""
        stats = _capi.Stats()
        _capi.check(self._lib.mi_ilqr_solve(self._h, C.byref(stats)), "mi_ilqr_solve")
        self.stats = stats
        return stats

    def solve_resident_async(self):
        """Enqueue a solve from the resident inputs on the handle's stream and return at once; up to
        32 may be in flight before `collect(count)` (each keeps its own events and statistics)."""
        _capi.check(self._lib.mi_ilqr_solve_async(self._h), "mi_ilqr_solve_async")

    def collect(self, count=1):
        """Wait for the stream and return the statistics of the last `count` enqueued solves, oldest first."""
        arr = (_capi.Stats * int(count))()
        _capi.check(self._lib.mi_ilqr_collect_stats_n(self._h, int(count), arr), "mi_ilqr_collect_stats_n")
        self.stats = arr[count - 1]
        return list(arr)

    def rearm(self, cold=True):
        if cold:
            _capi.check(self._lib.mi_ilqr_reset(self._h), "mi_ilqr_reset")
        _capi.check(self._lib.mi_ilqr_rearm_initial_guess(self._h), "mi_ilqr_rearm_initial_guess")

    def MPCShift(self, replan_steps):
        """On-device warm start of the receding-horizon loop (acrobot.py:147-152)."""
        _capi.check(self._lib.mi_ilqr_mpc_shift(self._h, int(replan_steps)), "mi_ilqr_mpc_shift")

    def MPCRun(self, num_resolves, replan_steps, target_step=None):
        """The receding-horizon loop (acrobot.py:145-155, mini_cheetah.py:190-201) kept on the device:
        num_resolves x { shift warm start; x_nom += target_step; Solve }.  One launch for the
        wave-per-problem and workgroup-per-problem models (the lane-per-problem "throughput" kernels loop on
        the host).  Returns the aggregate stats; `mpc_log` has the per-re-solve record."""
        ts = None
        if target_step is not None:
            ts = _capi.as_f64(target_step, (self.n,))
            self.x_nom = np.asarray(self.x_nom, dtype=np.float64) + num_resolves * ts
        stats = _capi.Stats()
        _capi.check(self._lib.mi_ilqr_mpc_run(self._h, int(num_resolves), int(replan_steps), _capi.ptr(ts), C.byref(stats)),
                    "mi_ilqr_mpc_run")
        self.stats = stats
        self._mpc_resolves = int(num_resolves)
        self._check_internal(stats)
        return stats

    @property
    def mpc_log(self):
        """(B, num_resolves, n+2): x0 of each re-solve | cost | iterations (written by the single-launch kernels, or after each re-solve of the host loop)."""
        out = np.empty((self.B, self._mpc_resolves, self.n + 2), dtype=np.float64)
        _capi.check(self._lib.mi_ilqr_get_mpc_log(self._h, _capi.ptr(out), out.nbytes), "mi_ilqr_get_mpc_log")
        return out

    def SetTargetStateResident(self, x_nom):
        """Moving target between resident re-solves (mini_cheetah.py:151-156)."""
        self.SetTargetState(x_nom)
        xn = _capi.as_f64(self.x_nom, (self.n,))
        _capi.check(self._lib.mi_ilqr_set_cost(self._h, None, None, None, _capi.ptr(xn)), "mi_ilqr_set_cost")

    # ------------------------------------------------------------- stage-level entries
    def stage_rollout(self, eps):
        self._push_problem()
        eps = np.ascontiguousarray(np.broadcast_to(np.asarray(eps, dtype=np.float64), (self.B,)))
        _capi.check(self._lib.mi_ilqr_rollout(self._h, _capi.ptr(eps)), "mi_ilqr_rollout")
        tc = self._get(_capi.F_TRIAL_COST, (self.B, 2))
        return (self._get(_capi.F_X_TRIAL, (self.B, self.n, self.N)),
                self._get(_capi.F_U_TRIAL, (self.B, self.m, self.N - 1)), tc[:, 0], tc[:, 1])

    def stage_forward(self, L_last):
        self._push_problem()
        L_last = np.ascontiguousarray(np.broadcast_to(np.asarray(L_last, dtype=np.float64), (self.B,)))
        _capi.check(self._lib.mi_ilqr_forward(self._h, _capi.ptr(L_last)), "mi_ilqr_forward")
        h = self.history[:, 0, :]
        return h[:, 0], h[:, 1], h[:, 2].astype(int)

    def stage_linearize(self):
        self._push_problem()
        _capi.check(self._lib.mi_ilqr_linearize(self._h), "mi_ilqr_linearize")

    def stage_backward(self):
        self._push_problem()
        _capi.check(self._lib.mi_ilqr_backward(self._h), "mi_ilqr_backward")

    # ------------------------------------------------------------- multi-GPU helper
    def best_cost_allreduce(self):
        """min over all ranks of the best converged cost — the ONE collective of the
        path: a single RCCL all-reduce(min) per batched solve, off
        the per-iteration path.  No-op without an initialized process group."""
        best = float(self.stats.best_cost) if self.stats is not None else float(np.min(self.cost))
        return allreduce_min(best, self._desc.device_id)

    def best_cost_allreduce_async(self):
        """Same collective, non-blocking: returns a handle whose .wait() yields the global best
        cost, so the 8-byte RCCL reduction overlaps the next batched solve."""
        best = float(self.stats.best_cost) if self.stats is not None else float(np.min(self.cost))
        return allreduce_min_async(best, self._desc.device_id)


class IterativeLinearQuadraticRegulator(BatchedIterativeLQR):
    """Drop-in for the reference class (single problem): same constructor signature
    (ilqr.py:21-22), same setters, Solve() -> (x_bar (n,N), u_bar (m,N-1), solve_time,
    cost), SaveSolution(fname)."""

    def __init__(self, system, num_timesteps, input_port_index=0, delta=1e-2, beta=0.95, gamma=0.0,
                 derivs_keypoint_method=None, **device_options):
        self.verbose = device_options.pop("verbose", True)
        # the reference inverts every Quu with np.linalg.inv and never looks at its definiteness (ilqr.py:655): so does the drop-in.
        # `status` then carries STATUS_FLAG_INDEFINITE; on_indefinite="stop" raises RuntimeError at the first such Quu instead.
        device_options.setdefault("on_indefinite", "continue")
        # The reference's loop has no iteration cap (`while improvement > delta`, ilqr.py:692) and prints a row per iteration
        # (:704): so does the drop-in - no cap unless the caller gives `max_iters` (a solve that then runs into it RAISES: it is
        # not a converged one), and a log of 4096 rows (128 KB; `hist_cap`), of which the first 64 ride with the solve and the
        # rest is fetched only after a solve that took more iterations (cart_pole.py's problem takes 108).
        self._capped = "max_iters" in device_options
        device_options.setdefault("max_iters", 2 ** 31 - 1)
        device_options.setdefault("hist_cap", 4096)
        hc = min(int(device_options["hist_cap"]), self._LOG_ROWS_WITH_SOLVE)
        # what Solve() copies out behind the solve besides x_bar / u_bar / cost (field, shape, dtype)
        self._single_extra = ((_capi.I_ITERS, (1,), np.int32), (_capi.I_STATUS, (1,), np.int32), (_capi.F_HIST, (1, hc, 4), np.float64),
                              (_capi.F_ITER_CYCLES, (1, hc, 4), np.float64), (_capi.I64_STAGE_CYCLES, (1, 4), np.int64))
        super().__init__(system, num_timesteps, 1, input_port_index=input_port_index, delta=delta, beta=beta,
                         gamma=gamma, derivs_keypoint_method=derivs_keypoint_method, **device_options)
        self.x0 = np.zeros(self.n)

    _LOG_ROWS_WITH_SOLVE = 64

    def SetInitialGuess(self, u_guess):
        assert u_guess.shape == (self.m, self.N - 1)          # ilqr.py:155
        self._u_guess = u_guess

    def _log_rows(self, which, rows):
        """The leading `rows` rows of a per-iteration record of the last solve (mi_ilqr.h: partial reads of MI_F_HIST / MI_F_ITER_CYCLES)."""
        out = np.empty((rows, 4), dtype=np.float64)
        _capi.check(self._lib.mi_ilqr_get(self._h, which, _capi.ptr(out), out.nbytes), "mi_ilqr_get")
        return out

    x_bar = property(lambda s: s._get(_capi.F_X_BAR, (s.n, s.N)))
    u_bar = property(lambda s: s._get(_capi.F_U_BAR, (s.m, s.N - 1)))
    K = property(lambda s: s._get(_capi.F_K, (s.m, s.n, s.N - 1)))
    kappa = property(lambda s: s._get(_capi.F_KAPPA, (s.m, s.N - 1)))
    dV_coeff = property(lambda s: s._get(_capi.F_DV, (s.N - 1,)))
    fx = property(lambda s: s._get(_capi.F_FX, (s.n, s.n, s.N - 1)))
    fu = property(lambda s: s._get(_capi.F_FU, (s.n, s.m, s.N - 1)))

    @property
    def percentage_derivs(self):
        return float(self.keypoint_count[0]) / (self.N - 1) * 100.0

    def Solve(self):
        st = time.time()
        self._push_problem()
        res = None
        if self._pinned is not None:
            # ONE host synchronization for the whole call: results into page-locked arrays (by the kernel itself where it
            # can), the iteration log and the stopwatches copied out behind the solve on the same stream
            small = [(k,) + self._out_addr(k, shp, dt) for k, shp, dt in self._single_extra]
            res = self._solve_into_pinned(extra=small)
            stats = self.stats
            iters, status, hist, iter_cyc, loop_cycles = int(small[0][1][0]), int(small[1][1][0]), small[2][1][0], small[3][1][0], float(small[4][1][0, 3])
        else:
            stats = _capi.Stats()
            _capi.check(self._lib.mi_ilqr_solve(self._h, C.byref(stats)), "mi_ilqr_solve")
            self.stats = stats
            iters, status = int(self.iterations[0]), int(self.status[0])
            r0 = max(1, min(iters, self.hist_cap))
            hist, iter_cyc = self._log_rows(_capi.F_HIST, r0), self._log_rows(_capi.F_ITER_CYCLES, r0)
            loop_cycles = float(self.stage_cycles[0, 3])
        total_time = time.time() - st
        self.solve_wall_s = total_time
        # the reference's stopwatches (ilqr.py:364-372,696-702) from the in-kernel cycle counters, PER ITERATION:
        # cycles of the whole solve loop (stage_cycles[3]) span the kernel's HIP-event time
        sec_per_cycle = stats.kernel_ms * 1e-3 / max(loop_cycles, 1.0)
        rows = min(iters, self.hist_cap)
        if rows > len(hist):                                          # a long solve: the rest of the log (the reference prints every row, ilqr.py:704)
            hist, iter_cyc = self._log_rows(_capi.F_HIST, rows), self._log_rows(_capi.F_ITER_CYCLES, rows)
        t_iter = iter_cyc[:rows] * sec_per_cycle                      # columns: fp (line search), derivs, bp, iteration
        if rows:                                                      # like the reference: the LAST iteration's stopwatches
            self.time_fp, self.time_getDerivs, self.time_backwardsPass = (float(v) for v in t_iter[-1, :3])
        if self.verbose:
            # same table as ilqr.py:685-704
            print("----------------------------------------------------------------------------------------------------------------------------------")
            print("|    iter    |    cost    |    eps    |    ls    | derivs time | derivs '%'  | bp time  | fp time  |   iter time    |    time    |")
            print("----------------------------------------------------------------------------------------------------------------------------------")
            elapsed = 0.0
            for i in range(rows):
                L_new, eps, ls, pct = hist[i]
                t_fp, t_derivs, t_bp, t_it = t_iter[i]
                elapsed += t_it
                print(f"{i + 1:^14}{L_new:11.4f}  {eps:^12.4f}{int(ls):^11}   {t_derivs:1.5f}         {pct:.1f}       "
                      f"{t_bp:1.5f}    {t_fp:1.5f}      {t_it:1.5f}          {elapsed:4.2f}")
            if iters > rows:
                print(f"note: iterations {rows + 1} .. {iters} are not in the log (hist_cap = {self.hist_cap} rows)")
        self.met_indefinite_quu = bool(status & _capi.STATUS_FLAG_INDEFINITE)
        status &= ~_capi.STATUS_FLAG_INDEFINITE
        if self.met_indefinite_quu and self.verbose:
            print("note: a Quu of this solve was not positive definite; inverted like np.linalg.inv (ilqr.py:655) and carried on")
        if status == _capi.STATUS_LINESEARCH_FAILED:
            # ilqr.py:337 reports the trials of the FAILING line search (not the solve's running total): every step
            # size eps = beta^k >= 1e-8 was tried, counted with the reference's own float recurrence (ilqr.py:299-335)
            n_trials, eps = 0, 1.0
            while eps >= 1e-8:
                n_trials += 1
                eps *= self.beta
            raise RuntimeError("linesearch failed after %s iterations" % n_trials)
        if status == _capi.STATUS_INTERNAL:
            raise RuntimeError("solve aborted inside the device kernel (cluster hand-shake lost); results are not a solution")
        if status == _capi.STATUS_MAX_ITERS:
            # the reference has no cap (ilqr.py:692): what a capped solve returns is NOT what the reference would have returned
            raise RuntimeError(f"no convergence after max_iters = {int(self._desc.max_iters)} iterations (improvement still above delta = {self.delta}); "
                               "the state attributes hold the last iterate")
        if status == _capi.STATUS_NOT_PD:
            raise RuntimeError("Quu is not positive definite in the backward pass (indefinite cost expansion, or round-off) and "
                               "on_indefinite=\"stop\" was asked for; the default, \"continue\", inverts it like the reference (ilqr.py:655)")
        if res is not None:
            return res[0].reshape(self.n, self.N), res[1][0], total_time, float(res[2][0])
        return self.x_bar, self.u_bar, total_time, float(self.cost[0])

    def SaveSolution(self, fname):
        """ilqr.py:712-733: npz with t, x_bar (last step dropped), u_bar, K."""
        dt = self.system.GetSubsystemByName("plant").time_step()
        T = (self.N - 1) * dt
        t = np.arange(0, T, dt)
        x_bar = self.x_bar[:, :-1]
        np.savez(fname, t=t, x_bar=x_bar, u_bar=self.u_bar, K=self.K)
