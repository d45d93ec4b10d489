#!/bin/bash
# round-6 evidence in one call: the bench line + full record, rocprofv3 kernel stats of the same command, the K3 PMC passes, the chain step's phase stamps
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 50 --warmup 5 > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err
cp bench_full.json gpurun_out/r06_bench_full.json      # (the profiled runs below write their own bench_full.json)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-c5 > gpurun_out/prof_bench.log 2>&1
for f in $(find $OUT -name "*kernel_stats.csv"); do cp $f gpurun_out/r06_bench_kernel_stats.csv; done
rm -rf $OUT
bash scripts/gpu_pmc.sh > gpurun_out/pmc_run.log 2>&1
tail -2 gpurun_out/pmc_run.log | cut -c1-700
bash scripts/build_variant.sh stamps -DGLIO_DEV_STAMPS > /dev/null 2>&1
CST_STEADY=1 GLIO_HIP_LIB=glio_amd/lib/libglio_hip_stamps.so python scripts/chain_step_time.py > gpurun_out/r06_chain_step_phases.txt 2>&1
cat gpurun_out/r06_bench_line.json | cut -c1-400
grep -i "k_linearize_all\|k_chain_step\|k_marg" gpurun_out/r06_bench_kernel_stats.csv | head
