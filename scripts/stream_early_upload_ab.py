"""A/B of the early factor uploads (GLIO_EARLY_UPLOAD=0: glio_set_imu / glio_set_gnss copy on the context's stream and wait for it -- i.e. for the window's
searches; default: a stream of their own into a device mirror, k_unstage stays on the context's stream) on the C++ keyframe stream."""
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
    for rep in range(2):
        for early in ("0", "1"):
            env = dict(os.environ, GLIO_EARLY_UPLOAD=early)
            for mode in (0, 1):
                r = min((window_io.run_demo_stream(path, env=env, defer=mode) for _ in range(2)), key=lambda x: x["cycle_ms"])
                print("early", early, "deferred" if mode else "default ", "cycle_ms", r["cycle_ms"], {k: round(v, 3) for k, v in r["stages_ms"].items()}, r.get("factors_stage_ms"), "checksum", r["trans_checksum"], flush=True)
