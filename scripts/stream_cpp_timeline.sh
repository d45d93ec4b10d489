#!/bin/bash
# kernel timeline of the LAST keyframe cycle of the C++ stream demo (host_demo_stream, mode $1: 0 default / 1 deferred / 2 batch association after the
# marginalization; further arguments go to the program, e.g. ahead=1): every kernel and copy with start offset, duration and gap; GPU-busy time of the cycle
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
MODE=${1:-2}
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
from glio_amd import synth
from glio_amd.host import window_io
W, pts, NK = 20, 65536, 6
long = synth.make_window(W=W + NK, pts_per_scan=pts, with_gnss=True, with_prior=False, seed=synth.SEED_BASE + 12)
wins = [synth.sub_window(long, j, W) for j in range(NK + 1)]
opts = wins[0].opts
opts.max_ddt_epochs = max(w.init.n_ddt for w in wins) + 8
opts.max_map_points = 1 << 18
window_io.write_stream("/tmp/stream_tl.bin", long, wins, W, NK, pts)
window_io.build_demo_stream()
PY
OUT=/tmp/stl_cpp; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o tl -- glio_amd/host/host_demo_stream /tmp/stream_tl.bin 0 6 $MODE "${@:2}" > /tmp/stl_cpp.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
m=$(find $OUT -name "*memory_copy_trace.csv" | head -1)
python - "$f" "$m" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]) for r in rows]
try:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", r.get("Name", ""))[:24]))
except Exception as e:
    print("no copy trace", e)
ev.sort()
starts = [i for i, t in enumerate(ev) if t[2].startswith("k_lm_bbox_init")]     # one local-map push per keyframe
a, b = starts[-2], starts[-1]
sel = ev[a:b]
t0 = sel[0][0]
prev = t0
for s, e, n in sel:
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:8.2f}  gap {(s - prev) / 1e3:7.2f}  {n}")
    prev = max(prev, e)
busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]
for s, e, n in sel[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"events {len(sel)} cycle {(ev[b][0] - t0) / 1e3:.1f} us  gpu busy {busy / 1e3:.1f} us")
PY
tail -1 /tmp/stl_cpp.log | cut -c1-500
