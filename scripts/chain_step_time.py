"""Phase stamps of k_chain_step on the steady-state C2 window (build with GLIO_DEV_STAMPS=1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from glio_amd import synth, capi
W = 20
stream = synth.make_window(W=W + 1, pts_per_scan=65536, with_gnss=True, seed=synth.SEED_BASE + 12)
first = synth.sub_window(stream, 0, W)
c0 = capi.Context(first.opts); c0.load_window(first, synth.analytic_correspondences(first))
s0, _ = c0.solve(first.init); prior = c0.marginalize(s0); c0.close()
win = synth.sub_window(stream, 1, W); win.prior = prior
corr = synth.analytic_correspondences(win)
ctx = capi.Context(win.opts); ctx.load_window(win, corr)
# GLIO_CHAIN_FAST masks: 0 = generic bodies through the global work vectors, 1 = tail from LDS, 2 = front from LDS, 3 = both (default)
for mask in (0, 1, 2, 3, 0, 3):
    capi.load().glio_debug_chain_fast(mask)
    sol, summ = ctx.solve(win.init)
    ms, _ = ctx.time_solve(win.init, 20)
    print("fast mask", mask, "path", capi.load().glio_debug_solver_path(ctx._h), "iterations", summ.iterations, "solve ms", round(ms, 4), "tr_step us",
          round(ctx.time_kernel(2, 40) * 1e3, 2), "linearize_all us", round(ctx.time_kernel(7, 20) * 1e3, 2), "trans checksum", float(sol.trans.sum()))
steady = os.environ.get("CST_STEADY", "0") == "1"      # phases of a LATER step (accepted candidate pending: the helpers' speculative build applies)
if steady:
    print("steady step us", round(ctx.time_kernel(8, 40) * 1e3, 2), "first step us", round(ctx.time_kernel(2, 40) * 1e3, 2))
ctx.time_kernel(8 if steady else 2, 1)
st = (C.c_longlong * 320)()
capi.load().glio_debug_arrow_stamps(ctx._h, st)
v = list(st)
names = ["tables", "gather diag/g/cost", "state machine", "epoch cols", "block gather+scale", "t = H u", "epoch corrections", "chain", "back subst + z", "factor body", "dogleg"]
print("k_chain_step phases (us):")
for k, nm in enumerate(names):
    print(f"  {nm:22s} {(v[41 + k] - v[40 + k]) / 100.0:7.2f}")
print(f"  total                  {(v[51] - v[40]) / 100.0:7.2f}")
if steady and v[300]:
    print(f"  (this step took the fat helpers' products: the four phases between the state machine and the chain are ONE load phase of {(v[47] - v[43]) / 100.0:.2f} us;")
    print("   their own lines above and the sub-phase lines of epoch columns / t = H u / epoch corrections below mix stamps of earlier launches)")
print("chain step phases, totals over the top half-chain of 10 steps (us): loads, 15 pivots, panel store, rank-15 update, correction:", [round(v[60 + k] / 100.0, 2) for k in range(5)])

def d(a, b):
    return round((v[a] - v[b]) / 100.0, 2)
print("front (fast): loads issued", d(74, 41), "K3 partial sums", d(75, 74), "barrier", d(76, 75), "finish", d(42, 76))
print("state machine (fast): status", d(80, 42), "step-norm sums", d(81, 80), "decision", d(82, 81), "gradient max", d(83, 82), "block max", d(84, 83),
      "loop top", d(85, 84), "work vectors", d(86, 85), "write back", d(43, 86))
print("tail (fast): loop 1 + five sums", d(70, 49), "coefficients + loop 2", d(71, 70), "three sums", d(72, 71), "candidate", d(73, 72), "status", d(50, 73))
print("epoch columns: round 1 + barrier", d(90, 43), "u / S, 1 / sqrt(m) + barrier", d(91, 90), "V = S c S / sqrt(m)", d(92, 91), "row mask", d(93, 92), "y_d", d(44, 93))
print("t = H u: rows", d(94, 45), "barrier", d(46, 94))
print("epoch corrections: row list", d(95, 46), "index lists + barrier", d(96, 95), "rank-one corrections + barrier", d(47, 96))
print("chain (wave 0): top front", d(100, 47), "wait for the bottom front", d(101, 100), "middle step", d(102, 101), "middle back substitution", d(103, 102), "barrier", d(48, 103))
if os.environ.get("GLIO_CHAIN_FRONTS", "4") != "2" and v[111]:
    print("four fronts, us after the start of the chain: front A done", d(100, 47), "B", d(113, 47), "C", d(116, 47), "D", d(114, 47), "| left meeting block done", d(111, 47),
          "right", d(115, 47), "| separator factored", d(102, 47), "its back substitution", d(103, 47))
print("back substitution: flag + barrier", d(97, 48), "half chains (wave 0)", d(98, 97), "barrier", d(99, 98), "epochs + z", d(49, 99))
print("epoch corrections on the matrix core (wave 0): entry", d(104, 96), "lane roles", d(105, 104), "offsets + accumulators", d(106, 105), "operands", d(107, 106), "MFMA", d(108, 107), "stores", d(109, 108), "return", d(110, 109), "barrier", d(47, 110))
print("shader clock over the step: %.0f MHz (clock64 ticks / wall-clock time between the first and the last stamp)" % ((v[121] - v[120]) / max(1, (v[73] - v[40])) * 100.0))

if steady:
    print("fat helper W/2 (us after the main workgroup's first stamp): start", d(304, 40), "barriers", [d(304 + k, 40) for k in range(1, 9)], "end", d(319, 40))
