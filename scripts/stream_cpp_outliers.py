"""host_demo_stream on the C2 stream, N runs of one mode (SCO_MODE = default | host_draws | deferred | ahead | map_ahead): cycle and the slowest instance of every stage (stage_max_ms)."""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glio_amd import synth
from glio_amd.host import window_io
W, pts, NK = 20, 65536, 8
long = synth.make_window(W=W + NK, pts_per_scan=pts, with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 12)
wins = [synth.sub_window(long, j, W) for j in range(NK + 1)]
opts = wins[0].opts
opts.max_ddt_epochs = max(w.init.n_ddt for w in wins) + 8
opts.max_map_points = 1 << 18
mode = os.environ.get("SCO_MODE", "default")
kw = {"default": {}, "host_draws": {"stream_draws": False}, "deferred": {"defer": True}, "ahead": {"ahead": True}, "map_ahead": {"ahead": True, "map_ahead": True}}[mode]
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "s.bin")
    window_io.write_stream(path, long, wins, W, NK, pts)
    for rep in range(int(os.environ.get("SCO_N", "12"))):
        g = window_io.run_demo_stream(path, **kw)
        print(mode, "cycle", round(g["cycle_ms"], 4), "min/max", [round(v, 3) for v in g["cycle_ms_min_max"]], "stage max", [round(v, 3) for v in g["stage_max_ms"]], flush=True)
