"""The cross-lane exchanges of glio_device.h (V_PERMLANE32/16_SWAP + DPP) against the __shfl_xor forms they replace: wave sums and maxima,
single butterfly stages at every distance, and the select-free reduce-scatter step of the value-splitting butterflies -- bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_lane_exchanges_equal_the_shuffle_forms_bit_for_bit():
    from glio_amd import capi, synth
    win = synth.make_window(W=5, pts_per_scan=64)
    ctx = capi.Context(win.opts)
    rng = np.random.default_rng(11)
    rounds = 16
    vals = rng.normal(0, 1, 64 * rounds) * 10.0 ** rng.integers(-8, 9, 64 * rounds)      # mixed magnitudes: the association of every sum matters
    vals[:64] = np.arange(64)                                                          # and one round where a wrong partner is obvious
    bad = C.c_int(-1)
    rc = capi.load().glio_debug_wave_reduce_check(ctx._h, vals.ctypes.data_as(C.POINTER(C.c_double)), rounds, C.byref(bad))
    assert rc == 0 and bad.value == 0, (rc, bad.value)
    ctx.close()
