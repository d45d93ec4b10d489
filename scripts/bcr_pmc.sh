#!/bin/bash
# hardware counters of the batch solve's kernels (one --pmc pass per group) around scripts/batch_tr_time.py: what bounds k_bcr_pre / elim2 / k_batch_pairs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PAT=${1:-k_bcr_pre}
for GRP in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT"; do
  OUT=/tmp/bcr_pmc; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $GRP --output-format csv -d $OUT -o pmc -- python scripts/batch_tr_time.py 2000 4096 > /tmp/bcr_pmc.log 2>&1
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python - "$f" "$PAT" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for row in csv.DictReader(open(sys.argv[1])):
        kn = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if kn.startswith(sys.argv[2]):
            acc[(kn, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (kn, k), v in sorted(acc.items()):
        print(f"{kn:20s} {k:32s} max over launches {max(v):16.1f}  (launches {len(v)})")
    if not acc: print("no rows:", open("/tmp/bcr_pmc.log").read()[-500:])
except Exception as e:
    print("no data:", e); print(open("/tmp/bcr_pmc.log").read()[-600:])
PY
done
