"""Item 6 of the round-5 verdict: a 10 Hz caller leaves ~100 ms between calls.  host_demo_stream with the host sleeping inside every keyframe call, at three places;
prints the per-stage maxima next to the averages (a stall shows as a maximum far above the average)."""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glio_amd import synth
from glio_amd.host import window_io
W, pts, NK = 20, int(os.environ.get("STALL_PTS", "65536")), 8
long = synth.make_window(W=W + NK, pts_per_scan=pts, with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 12)
wins = [synth.sub_window(long, j, W) for j in range(NK + 1)]
opts = wins[0].opts
opts.max_ddt_epochs = max(w.init.n_ddt for w in wins) + 8
opts.max_map_points = 1 << 18
names = ["slide_and_new_scan", "local_map", "associate_enqueue", "factors+wait", "solve", "marginalize", "batch_assoc"]
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "s.bin")
    window_io.write_stream(path, long, wins, W, NK, pts)
    window_io.run_demo_stream(path)
    for sleep_ms, at in [(0, 0)] + [tuple(int(v) for v in a.split(':')) for a in os.environ.get('STALL_CASES', '100:0,100:1,100:2').split(',')] + [(0, 0)]:
        g = window_io.run_demo_stream(path, sleep_ms=sleep_ms, sleep_at=at)
        avg = [round(v, 3) for v in g["stages_ms"].values()]
        print(json.dumps({"sleep_ms": sleep_ms, "sleep_at": at, "cycle_ms": g["cycle_ms"], "avg": dict(zip(names, avg)), "max": dict(zip(names, [round(v, 3) for v in g["stage_max_ms"]]))}))
