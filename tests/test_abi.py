"""CPU checks of the drop-in boundary: libglio_hip.so loads without a GPU, exports every symbol that
include/glio_hip.h declares, agrees on struct layouts with the ctypes mirror, and fails loudly (no CPU
fallback) when no HIP device is present."""
import ctypes as C
import os
import re

import pytest

from glio_amd import ctypes_types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from glio_amd import build, capi
    build.build()
    return capi.load()


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "glio_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(glio_[a-z_0-9]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/glio_hip.h but not exported"


def test_debug_hooks_the_gpu_tests_rely_on_are_exported(lib):
    """Undeclared test hooks (not part of the boundary): a stale library without them should fail here, on the CPU, not in the GPU suite."""
    for n in ("glio_debug_set_solver", "glio_debug_solver_path", "glio_debug_chain_fast", "glio_debug_chain_fronts", "glio_debug_chain_fronts_used", "glio_debug_chain_f4_layout", "glio_debug_wave_reduce_check", "glio_debug_chol_solve",
              "glio_debug_arrow_stamps", "glio_debug_read_vec"):
        assert hasattr(lib, n), n


def test_struct_layouts_match(lib):
    out = (C.c_int32 * 9)()
    assert lib.glio_struct_sizes(out, 9) == 9
    mine = [C.sizeof(x) for x in (T.GlioOpts, T.GlioState, T.GlioPreint, T.GlioPrior, T.GlioDdPsr, T.GlioDoppler, T.GlioGnssFrame, T.GlioSummary, T.GlioBatchTrOpts)]
    assert list(out) == mine


def test_defaults_are_the_reference_yaml(lib):
    o = T.GlioOpts()
    lib.glio_opts_default(C.byref(o))
    assert (o.window, o.max_iterations, o.jacobi_scaling) == (5, 15, 1)
    assert (o.huber_delta, o.lidar_const, o.surf_dist_thres) == (1.0, 7.5, 0.18)
    assert o.kd_max_radius == 1.5 and o.weight_gate == 0.3
    assert list(o.t_lb) == [0.0, 0.0, 0.28] and list(o.q_lb) == [1.0, 0.0, 0.0, 0.0]
    assert o.gravity == 9.80511 and o.initial_trust_region_radius == 1e4 and o.function_tolerance == 1e-6


def test_no_cpu_fallback(lib):
    """On a box without a HIP device the product path must refuse, not fall back."""
    if lib.glio_device_count() >= 1:
        pytest.skip("HIP device present")
    from glio_amd import capi, synth
    with pytest.raises(capi.GlioError):
        capi.Context(synth.default_opts())
    h = C.c_void_p()
    o = synth.default_opts()
    assert lib.glio_create(0, C.byref(o), C.byref(h)) != 0
    assert b"no HIP device" in lib.glio_last_error() or b"hip" in lib.glio_last_error().lower()


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "glio_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "glio_oracle" not in txt and "orc_" not in txt, f
                # ... nor the reference's own factor code built for the checker (oracle/_ref, oracle/ref_shim, oracle/pyref.py)
                assert "pyref" not in txt and "libglio_ref" not in txt and "ref_shim" not in txt and "ref_eval_" not in txt and "/root/reference" not in txt, f


def test_product_library_does_not_link_the_checkers():
    """libglio_hip.so and the C++ host demos need neither the oracle nor the reference build: no such DT_NEEDED entry, no such symbol"""
    import subprocess
    so = os.path.join(ROOT, "glio_amd", "lib", "libglio_hip.so")
    dyn = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "libglio_oracle" not in dyn and "libglio_ref" not in dyn
    syms = subprocess.run(["nm", "-D", so], capture_output=True, text=True).stdout
    assert " orc_" not in syms and " ref_eval_" not in syms and " ref_marginalize" not in syms
