"""CPU-side checks (no GPU): the C-ABI library loads, exports every symbol that
include/mi_ilqr.h declares, fails loudly (no CPU fallback) without a device, and the
host-side mirror validates arguments like the reference class does."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from drake_ddp_amd import build, _capi
    build.build(force=False, verbose=False)
    return _capi.load()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mi_ilqr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_ilqr_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from drake_ddp_amd import _capi
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/mi_ilqr.h but not exported"
    assert sorted(_capi.EXPORTS) == syms       # the ctypes binding covers the whole header


def test_abi_version_and_strerror(lib):
    from drake_ddp_amd import _capi
    assert lib.mi_ilqr_abi_version() == _capi.ABI_VERSION
    assert b"linesearch failed" in lib.mi_ilqr_strerror(_capi.E_LINESEARCH)
    assert b"no CPU fallback" in lib.mi_ilqr_strerror(_capi.E_NO_DEVICE)


def test_model_registry_matches_python_side(lib):
    from drake_ddp_amd import models
    for mid, (n, m) in models._DIMS.items():
        nn, mm, npar = C.c_int32(), C.c_int32(), C.c_int32()
        buf = (C.c_double * 16)()
        assert lib.mi_ilqr_model_info(mid, C.byref(nn), C.byref(mm), C.byref(npar), buf) == 0
        assert (nn.value, mm.value) == (n, m)
        assert list(buf)[:npar.value] == models._DEFAULTS[mid]
    assert lib.mi_ilqr_model_info(99, None, None, None, None) != 0


def test_bytes_per_iteration_formula(lib):
    # SURVEY.md §8d per-unit figures
    assert lib.mi_ilqr_bytes_per_iteration(2, 1, 200, 1) == 49424
    assert lib.mi_ilqr_bytes_per_iteration(4, 1, 40, 1) == 22288
    assert lib.mi_ilqr_bytes_per_iteration(4, 1, 200, 1) == 113168
    assert lib.mi_ilqr_bytes_per_iteration(36, 12, 40, 1) == 1416704


def test_create_validates_and_fails_loudly_without_gpu(lib):
    import torch
    from drake_ddp_amd import _capi
    d = _capi.Desc()
    d.n, d.m, d.N, d.B, d.model_id, d.minN, d.fd_step = 2, 1, 200, 4, 0, 1, 1e-5
    h = C.c_void_p()
    bad = _capi.Desc.from_buffer_copy(d); bad.n = 3
    assert lib.mi_ilqr_create(C.byref(bad), C.byref(h)) == _capi.E_BAD_SHAPE
    bad = _capi.Desc.from_buffer_copy(d); bad.keypoint_method = 7
    assert lib.mi_ilqr_create(C.byref(bad), C.byref(h)) == _capi.E_BAD_METHOD
    bad = _capi.Desc.from_buffer_copy(d); bad.minN = 0
    assert lib.mi_ilqr_create(C.byref(bad), C.byref(h)) == _capi.E_BAD_ARG
    if not torch.cuda.is_available():
        assert lib.mi_ilqr_create(C.byref(d), C.byref(h)) == _capi.E_NO_DEVICE   # never a CPU fallback
        assert not h.value


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "drake_ddp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def test_host_mirror_argument_checks():
    """Same assertion behaviour as the reference setters (ilqr.py:130-131,145,155)."""
    import torch
    from drake_ddp_amd.models import ModelSystem, Pendulum
    from drake_ddp_amd.utils_derivs_interpolation import derivs_interpolation, index_tuple
    assert derivs_interpolation('setInterval', 1, 0, 0, 0).keypoint_method == 'setInterval'
    assert index_tuple(1, 2).end_index == 2
    sysm = Pendulum(1e-2)
    assert sysm.IsDifferenceEquationSystem()[0] and sysm.GetSubsystemByName("plant").time_step() == 1e-2
    assert isinstance(sysm, ModelSystem) and (sysm.n, sysm.m) == (2, 1)
    if torch.cuda.is_available():
        return
    from drake_ddp_amd.ilqr import IterativeLinearQuadraticRegulator
    from drake_ddp_amd._capi import MiIlqrError
    with pytest.raises(MiIlqrError):          # constructing the solver needs the device: loud failure
        IterativeLinearQuadraticRegulator(sysm, 200)


def test_shard_range_partitions_exactly():
    from drake_ddp_amd.ilqr import shard_range
    for B in (1, 7, 64, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [shard_range(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_mpc_shift_matches_reference_callers():
    """acrobot.py:147-152 / mini_cheetah.py:193-198 warm start."""
    from drake_ddp_amd.workloads import mpc_shift
    rng = np.random.default_rng(0)
    x, u = rng.normal(size=(4, 40)), rng.normal(size=(1, 39))
    replan = 2
    last_u = u[:, -1]
    ref_guess = np.block([u[:, replan:], np.repeat(last_u[np.newaxis].T, replan, axis=1)])
    x0, ug = mpc_shift(x, u, replan)
    assert np.array_equal(ug, ref_guess) and np.array_equal(x0, x[:, replan])


def test_struct_sizes_match_the_ctypes_mirror(lib):
    """mi_ilqr_struct_sizes: the library's sizeof(mi_ilqr_desc / _stats / _model_plugin) against the ctypes restatements of
    drake_ddp_amd/_capi.py and plugin.py - a field added on one side only must fail at load time, not shift fields silently."""
    from drake_ddp_amd import _capi, plugin
    d, s, p = C.c_int32(), C.c_int32(), C.c_int32()
    lib.mi_ilqr_struct_sizes(C.byref(d), C.byref(s), C.byref(p))
    assert (d.value, s.value, p.value) == (C.sizeof(_capi.Desc), C.sizeof(_capi.Stats), C.sizeof(plugin._Plugin))
    lib.mi_ilqr_struct_sizes(None, None, None)            # (any pointer may be NULL)


def test_no_spill_copy_under_a_zero_exec_mask_in_the_built_kernels(lib):
    """A miscompile of this hipcc that round 5 tracked down with rocgdb: the copy that spills a VGPR to an
    accumulation register placed at the top of a control-flow join block BEFORE the s_or_b64 that re-activates the lanes - run
    with EXEC = 0 it saves nothing and the reload returns a stale value (the coupled arm's receding-horizon kernel hung on it).
    tools/check_exec_spill.py walks the ISA of every kernel object of the build the library was linked from: no such site -
    and it does recognize one (a synthetic block of the shape the compiler emitted)."""
    import glob
    import subprocess
    import sys
    objs = sorted(o for o in glob.glob(os.path.join(ROOT, "drake_ddp_amd", "lib", "obj", "k_*.o")) if "-" not in os.path.basename(o))
    assert len(objs) >= 10, objs
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_exec_spill.py")] + objs, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert re.search(r"(\d+) kernel object\(s\), 0 spill copies", r.stdout), r.stdout[-500:]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_exec_spill
    bad = """
	s_and_saveexec_b64 s[0:1], vcc                             // 000000001000: BE80206A
	s_cbranch_execz 2                                          // 000000001004: BF880002
	ds_write_b32 v1, v0                                        // 000000001008: D81A0000 00000001
	v_add_u32_e32 v0, 0x100, v0                                // 00000000100C: 680000FF
	v_writelane_b32 v250, s24, 49                              // 000000001010: D28A00FA 00016218
	v_accvgpr_write_b32 a8, v8                                 // 000000001018: D3D94008 18000108
	s_or_b64 exec, exec, s[0:1]                                // 000000001020: 87FE007E
	s_barrier                                                  // 000000001024: BF8A0000
"""
    found = check_exec_spill.sites(bad)
    assert len(found) == 1 and found[0][1].startswith("v_accvgpr_write_b32 a8"), found
    good = bad.replace("v_accvgpr_write_b32 a8, v8                                 // 000000001018: D3D94008 18000108\n\ts_or_b64 exec, exec, s[0:1]                                // 000000001020: 87FE007E",
                       "s_or_b64 exec, exec, s[0:1]                                // 000000001018: 87FE007E\n\tv_accvgpr_write_b32 a8, v8                                 // 00000000101C: D3D94008 18000108")
    assert good != bad and check_exec_spill.sites(good) == []


def test_no_dpp_read_after_write_hazard_in_the_built_kernels(lib):
    """The Gauss-Jordan elimination's column update is inline assembly (v_fmac_f64_dpp with a row_newbcast source,
    csrc/ilqr_large.hpp) - invisible to the compiler's hazard recognizer, which otherwise guarantees two wait states between a
    VALU write of a register and a DPP read of it.  tools/check_dpp_hazard.py walks the ISA of every kernel object of the build
    the library was linked from and must find the elimination's instructions and no violation."""
    import glob
    import subprocess
    import sys
    objs = sorted(o for o in glob.glob(os.path.join(ROOT, "drake_ddp_amd", "lib", "obj", "k_*.o")) if "-" not in os.path.basename(o))
    assert len(objs) >= 10, objs                                           # (the fixture built them)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazard.py")] + objs, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    m = re.search(r"(\d+) DPP instructions in (\d+) objects: 0 hazard violation", r.stdout)
    assert m and int(m.group(1)) > 5000, r.stdout[-500:]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_mix
    text = isa_mix.device_asm(os.path.join(ROOT, "drake_ddp_amd", "lib", "obj", "k_synth36.o"))
    assert text.count("v_fmac_f64_dpp") >= 12 * 11                          # (the 12 x 12 elimination, at least once)
