#!/bin/bash
# HBM traffic of the dominant kernel (K3) from the TCC fabric counters, one --pmc pass per counter
# (MI355X_MICROARCH.md: FETCH_SIZE costs 3 of 4 TCC slots; on gfx950 it reports 1/2 of a wide coalesced read).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for CNT in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$CNT
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $CNT --output-format csv -d $OUT -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$CNT.log 2>&1
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python - "$f" $CNT <<'PY'
import csv, sys, collections
f, cnt = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
with open(f) as fh:
    for row in csv.DictReader(fh):
        if row.get("Counter_Name") == cnt:
            acc[row["Kernel_Name"][:40]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:8]:
    big = [x for x in v if x > 0.5 * max(v)]
    print(f"{cnt} {k:40s} launches {len(v):4d} max {max(v):12.1f} mean_of_full_launches {sum(big)/len(big):12.1f}")
PY
  rm -rf $OUT
done
