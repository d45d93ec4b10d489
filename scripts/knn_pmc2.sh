#!/bin/bash
# second counter set for the K2 search kernels (what the wavefronts wait for): KNN_WINDOW=1 -> the merged window call
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
[ -f /tmp/knn_once.py ] || { echo "run scripts/knn_pmc.sh first (it writes /tmp/knn_once.py)"; exit 1; }
for GRP in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  OUT=/tmp/knn_pmc2; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $GRP --output-format csv -d $OUT -o pmc -- python /tmp/knn_once.py > /tmp/knn_pmc2.log 2>&1
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for row in csv.DictReader(open(sys.argv[1])):
        kn = row["Kernel_Name"].split("(")[0]
        if "k_knn5" in kn or "k_plane" in kn:
            acc[(kn, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (kn, k), v in sorted(acc.items()):
        print(f"{kn:22s} {k:28s} last launch {v[-1]:16.1f}  (launches {len(v)})")
    if not acc: print("no rows:", open("/tmp/knn_pmc2.log").read()[-300:])
except Exception as e:
    print("no data:", e, open("/tmp/knn_pmc2.log").read()[-300:])
PY
done
