"""host_demo_stream on the C2 stream: the default order and the deferred batch association, three runs each (cycle and stage table)."""
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
    for rep in range(int(os.environ.get("SCM_REPS", "3"))):
        for defer in (False, True):
            g = window_io.run_demo_stream(path, defer=defer, prepare_early=os.environ.get('SCM_PREP_EARLY', '1') == '1')
            print(json.dumps({"deferred": defer, "cycle_ms": g["cycle_ms"], "stages": {k: round(v, 3) for k, v in g["stages_ms"].items()}}))
