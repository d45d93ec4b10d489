#!/bin/bash
# kernel timeline of ONE trust-region kernel group of the batch problem (full 15-state problem, C4 shape): rocprofv3 --kernel-trace around
# scripts/batch_tr_time.py, then every kernel between two consecutive k_bt_state_machine launches of the last solve with start offset / duration / gap
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/btl
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python scripts/batch_tr_time.py > gpurun_out/btl.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ts = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:30]) for r in rows]
pairs = [i for i, t in enumerate(ts) if t[2].startswith("k_bt_state_machine")]
a, b = pairs[-4], pairs[-3]          # a complete group in the middle of the last solve (from one state machine to the next)
sel = ts[a:b]
t0 = sel[0][0]
prev_end = t0
for s, e, n in sel:
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:6.2f}  {n}")
    prev_end = e
tot = ts[b][0] - t0
busy = sum(e - s for s, e, n in sel)
print(f"kernels {len(sel)} group {tot / 1e3:.1f} us busy {busy / 1e3:.1f} us gaps {(tot - busy) / 1e3:.1f} us")
agg = {}
for s, e, n in sel:
    agg.setdefault(n, [0, 0]); agg[n][0] += 1; agg[n][1] += e - s
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:32s} x{c:3d}  {d / 1e3:8.1f} us")
PY
rm -rf $OUT
