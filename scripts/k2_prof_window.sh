#!/bin/bash
# per-kernel times of the one-call window association for library variants: scripts/k2_prof_window.sh prod r03 ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "prod" ]; then unset GLIO_HIP_LIB; else export GLIO_HIP_LIB=glio_amd/lib/libglio_hip_$v.so; fi
  OUT=/tmp/knnprofw_$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python scripts/knn_prof_window.py > /tmp/k2w.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out; cp $f $GRAFT_REPO_ROOT/gpurun_out/k2_window_kernel_stats_$v.csv      # (copy into profiles/rNN_k2_window_kernel_stats.csv: bench.py reads the newest)
  echo "== $v"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 10: print("%-20s calls %4s avg %9.1f us" % (r["Name"][:20], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
