"""host_demo_stream on the C2 stream under its options (stream_draws, prepare_early, deferred): cycle and the two closing stages, 3 runs each."""
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
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "s.bin")
    window_io.write_stream(path, long, wins, W, NK, pts)
    window_io.run_demo_stream(path)
    for rep in range(3):
        for name, kw in (("default", {}), ("host_draws", {"stream_draws": False}), ("prepare_late", {"prepare_early": False}), ("deferred", {"defer": True})):
            g = window_io.run_demo_stream(path, **kw)
            st = g["stages_ms"]
            print(name, "cycle", round(g["cycle_ms"], 4), "marginalize", round(st["marginalize"], 3), "batch", round(st["batch_feature_association_enqueue_and_wait"], 3), "checksum", g["trans_checksum"], flush=True)
