#!/bin/bash
# GPU-busy time of one keyframe cycle of the moving stream: rocprofv3 --kernel-trace (+ memory copies) around scripts/stream_time.py, the kernels of the
# LAST keyframe listed with start offset / duration / gap, busy vs wall
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/stl
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o tl -- python scripts/stream_time.py 6 > gpurun_out/stl.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
m=$(find $OUT -name "*memory_copy_trace.csv" | head -1)
python - "$f" "$m" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:32]) for r in rows]
try:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", r.get("Name", ""))[:24]))
except Exception as e:
    print("no copy trace", e)
ev.sort()
# keyframe boundaries: the slide kernel starts a cycle
starts = [i for i, t in enumerate(ev) if t[2].startswith("k_slide_scans")]
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
tail -1 gpurun_out/stl.log | cut -c1-400
rm -rf $OUT
