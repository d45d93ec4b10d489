#!/bin/bash
# host_demo_stream (default order, extra arguments "$@", e.g. group_early=1) under rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace: the LAST
# keyframe cycle's kernels and copies, and beside them every HIP call of the host thread that took more than 4 us (which call the host sat in)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
from glio_amd import synth
from glio_amd.host import window_io
W, pts, NK = 20, 65536, 4
long = synth.make_window(W=W + NK, pts_per_scan=pts, with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 12)
wins = [synth.sub_window(long, j, W) for j in range(NK + 1)]
opts = wins[0].opts
opts.max_ddt_epochs = max(w.init.n_ddt for w in wins) + 8
opts.max_map_points = 1 << 18
window_io.write_stream("/tmp/stream_tl.bin", long, wins, W, NK, pts)
window_io.build_demo_stream()
PY
OUT=/tmp/stl_api; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $OUT -o tl -- glio_amd/host/host_demo_stream /tmp/stream_tl.bin 0 6 0 "$@" > /tmp/stl_api.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
m=$(find $OUT -name "*memory_copy_trace.csv" | head -1)
h=$(find $OUT -name "*hip_api_trace.csv" | head -1)
python - "$f" "$m" "$h" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "    " + r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]) for r in rows]
for r in csv.DictReader(open(sys.argv[2])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "    copy:" + r.get("Direction", r.get("Name", ""))[:24]))
ev.sort()
starts = [i for i, t in enumerate(ev) if t[2].strip().startswith("k_lm_bbox_init")]
ta, tb = ev[starts[-2]][0], ev[starts[-1]][0]
api = []
for r in csv.DictReader(open(sys.argv[3])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 4000: api.append((s, e, "API " + r["Function"]))
sel = sorted([t for t in ev if ta <= t[0] < tb] + [t for t in api if ta - 150000 <= t[0] < tb])
for s, e, n in sel:
    print(f"{(s - ta) / 1e3:9.2f} us  dur {(e - s) / 1e3:8.2f}  {n}")
PY
tail -1 /tmp/stl_api.log | cut -c1-700
