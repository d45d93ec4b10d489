"""Host arithmetic of k_chain_step's four-front elimination (chain_f4_split / chain_f4_layout, glio_amd/csrc/solver_kernels.hip), swept on the CPU through a
test hook: the split covers every keyframe exactly once, and the LDS regions the panels are laid into -- the copy of the clock-drift blocks and the space
behind the gather index tables -- never overlap what lies between them nor exceed what the launch asks for."""
import ctypes as C

import pytest

from glio_amd import capi


def _layout(W, nd, mirrors):
    lib = capi.load()
    out = (C.c_longlong * 17)()
    assert lib.glio_debug_chain_f4_layout(W, nd, mirrors, out) == 0
    keys = ("regular", "off_dds", "dds_bytes", "k0", "off_r1", "total", "n_slots", "four", "s", "mL", "mR", "nA", "nB", "nC", "nD", "slot_bytes", "tile_bytes")
    return dict(zip(keys, list(out)))


@pytest.mark.parametrize("W", list(range(12, 41)))
def test_split_covers_every_keyframe_once(W):
    L = _layout(W, 0, 1)
    s, mL, mR = L["s"], L["mL"], L["mR"]
    a = list(range(0, L["nA"]))                       # front A eliminates 0 .. mL-1 upwards
    b = [s - 1 - k for k in range(L["nB"])]           # front B: s-1 downwards
    c = [s + 1 + k for k in range(L["nC"])]           # front C: s+1 upwards
    d = [W - 1 - k for k in range(L["nD"])]           # front D: W-1 downwards
    assert L["nB"] >= 1 and L["nC"] >= 1 and L["nA"] >= 1 and L["nD"] >= 1
    assert sorted(a + b + c + d + [mL, mR, s]) == list(range(W))
    assert a[-1] + 1 == mL == b[-1] - 1 and c[-1] + 1 == mR == d[-1] - 1
    assert L["n_slots"] == L["nB"] + L["nC"]
    # the inner fronts are not longer than the outer ones of their segment (their steps are the heavier ones)
    assert L["nB"] <= L["nA"] and L["nC"] <= L["nD"]


@pytest.mark.parametrize("mirrors", [0, 1])
def test_panels_fit_where_they_are_put(mirrors):
    limit = 158 * 1024
    took = 0
    for W in range(12, 27):
        for nd in range(0, 121, 3):
            L = _layout(W, nd, mirrors)
            # slots placed in the clock-drift copy stay inside it; the rest starts behind it (the descriptor tables lie in between)
            assert L["k0"] * L["slot_bytes"] <= L["dds_bytes"]
            assert L["off_r1"] >= L["off_dds"] + L["dds_bytes"]
            assert L["off_r1"] % 16 == 0 and L["off_dds"] % 8 == 0
            assert L["total"] == L["off_r1"] + (L["n_slots"] - L["k0"]) * L["slot_bytes"] + 2 * L["tile_bytes"]
            if L["four"]:
                took += 1
                assert max(L["total"], L["regular"]) + 256 <= limit
    assert took > 50            # (the sweep does exercise the four-front branch)
    # the headline window: 20 keyframes, 76 epochs, with the LDS mirrors -- four fronts, five of seven slots inside the clock-drift copy
    L = _layout(20, 76, 1)
    assert L["four"] == 1 and L["k0"] == 5 and L["n_slots"] == 7 and (L["s"], L["mL"], L["mR"]) == (10, 5, 14)
    # 22 keyframes with 84 epochs: no room for the mirrors, but the four-front panels fit the generic carve
    assert _layout(22, 84, 1)["four"] == 0 and _layout(22, 84, 0)["four"] == 1
