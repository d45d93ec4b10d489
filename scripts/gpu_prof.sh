#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; summaries copied to gpurun_out/prof_*
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps ${1:-5} --warmup 2 > gpurun_out/prof_bench.log 2>&1
find $OUT -name "*stats*" | head
for f in $(find $OUT -name "*kernel_stats.csv"); do cp $f gpurun_out/prof_kernel_stats.csv; done
head -30 gpurun_out/prof_kernel_stats.csv
tail -5 gpurun_out/prof_bench.log
