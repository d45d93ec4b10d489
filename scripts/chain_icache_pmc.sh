#!/bin/bash
# instruction-cache and issue counters of k_chain_step (one rocprofv3 --pmc pass per counter group, kernel trace only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
: > gpurun_out/chain_pmc.txt
for GROUP in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_chain
  rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --pmc $GROUP --output-format csv -d $OUT -o pmc -- python scripts/solve_chain_once.py > gpurun_out/pmc_chain.log 2>&1
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "group [$GROUP]: no output ($(tail -2 gpurun_out/pmc_chain.log | tr '\n' ' '))" >> gpurun_out/chain_pmc.txt; continue; fi
  python - "$f" <<'PY' >> gpurun_out/chain_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "k_chain_step" in k or "k_linearize_all" in k:
        acc[(k.split("(")[0][:24], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    v = sorted(v)
    print(f"{k:24s} {c:22s} launches {len(v):3d} median {v[len(v)//2]:14.1f} max {v[-1]:14.1f}")
PY
  rm -rf $OUT
done
cat gpurun_out/chain_pmc.txt
