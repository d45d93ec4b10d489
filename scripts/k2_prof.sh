#!/bin/bash
# per-kernel times of 30 associations of one C2 scan under rocprofv3 (-> gpurun_out/r04/k2_kernel_stats.csv)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
OUT=/tmp/knnprof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python scripts/knn_prof.py 0 > gpurun_out/r04/k2_prof.log 2>&1
for f in $(find $OUT -name "*kernel_stats.csv"); do cp $f gpurun_out/r04/k2_kernel_stats.csv; done
head -8 gpurun_out/r04/k2_kernel_stats.csv | cut -c1-60,200-400
tail -2 gpurun_out/r04/k2_prof.log
