"""A/B of the C++ keyframe stream (host_demo_stream): order of the batch association inside the call (0 default: enqueued after the solve, beside the
marginalization; 1 deferred to the next call; 2 after the marginalization) x stream priorities (GLIO_BASSOC_PRIORITY / GLIO_CTX_PRIORITY = 0: default)."""
import json, os, sys, tempfile
sys.path.insert(0, ".")
from glio_amd import synth
from glio_amd.host import window_io
W, pts, NK = 20, 65536, 8
long = synth.make_window(W=W + NK, pts_per_scan=pts, with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 12)
wins = [synth.sub_window(long, j, W) for j in range(NK + 1)]
opts = wins[0].opts
opts.max_ddt_epochs = max(w.init.n_ddt for w in wins) + 8
opts.max_map_points = 1 << 18
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "stream.bin")
    window_io.write_stream(path, long, wins, W, NK, pts)
    for prio in ("1", "0"):
        env = dict(os.environ, GLIO_BASSOC_PRIORITY=prio, GLIO_CTX_PRIORITY=prio)
        for mode in (0, 2, 1):
            r = min((window_io.run_demo_stream(path, env=env, defer=mode) for _ in range(2)), key=lambda x: x["cycle_ms"])
            print("priorities", prio, "mode", mode, "cycle_ms", r["cycle_ms"], {k: round(v, 3) for k, v in r["stages_ms"].items()}, flush=True)
