#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; the kernel_stats summary lands in gpurun_out/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python bench.py --steps ${1:-10} --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
for f in $(find $OUT -name "*kernel_stats.csv"); do cp $f gpurun_out/prof_kernel_stats.csv; done
rm -rf $OUT
cat gpurun_out/prof_kernel_stats.csv | cut -c1-220 | head -30
grep '^{' gpurun_out/prof_bench.log | cut -c1-400
