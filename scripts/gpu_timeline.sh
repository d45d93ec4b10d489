#!/bin/bash
# kernel timeline of one solve: rocprofv3 --kernel-trace, then print start offset / duration of every kernel of the last solve
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python ${1:-scripts/solve_once.py} > gpurun_out/tl.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last solve = kernels after the last big gap (> 200 us)
ts = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:28]) for r in rows]
cut = 0
for i in range(1, len(ts)):
    if ts[i][0] - ts[i - 1][1] > 200000:
        cut = i
sel = ts[cut:]
t0 = sel[0][0]
prev_end = t0
busy = 0
for s, e, n in sel[:64]:
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:6.2f}  {n}")
    prev_end = e
tot = sel[-1][1] - t0
busy = sum(e - s for s, e, n in sel)
print(f"kernels {len(sel)} total {tot / 1e3:.1f} us busy {busy / 1e3:.1f} us gaps {(tot - busy) / 1e3:.1f} us")
PY
rm -rf $OUT
