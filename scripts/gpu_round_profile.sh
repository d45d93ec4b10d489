#!/bin/bash
# end-of-step evidence: GPU tests, bench line, rocprofv3 kernel stats of the same bench command, PMC traffic passes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 50 --warmup 5 > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err
tail -c 400 gpurun_out/bench_line.err
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
# (--no-c5: the C5 leg launches the same kernel templates on a 10x larger window and would blur the per-kernel averages of the
#  headline workload; its own trace is taken by scripts/c5_pmc.sh)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-c5 > gpurun_out/prof_bench.log 2>&1
for f in $(find $OUT -name "*kernel_stats.csv"); do cp $f gpurun_out/prof_kernel_stats.csv; done
rm -rf $OUT
bash scripts/gpu_pmc.sh > gpurun_out/pmc_run.log 2>&1
tail -3 gpurun_out/pmc_run.log | cut -c1-600
cut -c1-1500 gpurun_out/bench_line.json
