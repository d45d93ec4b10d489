#!/bin/bash
# MFMA utilisation and HBM traffic of the fp32-Jacobian / MFMA form of K3 at the C5 shape (BASELINE configs[4]):
# separate --pmc passes (MI355X_MICROARCH.md: SQ 8 slots, FETCH_SIZE 3 of 4 TCC slots, WRITE_SIZE 2), kernel trace for the
# duration.  Writes gpurun_out/c5_pmc.json (+ .txt); copy into profiles/ to have bench.py quote it.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/c5_pmc.txt
run_pass () {   # name, counters...
  local NAME=$1; shift
  local OUT=$GRAFT_REPO_ROOT/gpurun_out/c5_$NAME
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc "$@" --output-format csv -d $OUT -o pmc -- python scripts/c5_launch.py 6 > gpurun_out/c5_$NAME.log 2>&1
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  cp "$f" gpurun_out/c5_$NAME.csv 2>/dev/null
  rm -rf $OUT
}
run_pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
OUT=$GRAFT_REPO_ROOT/gpurun_out/c5_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python scripts/c5_launch.py 10 > gpurun_out/c5_trace.log 2>&1
for f in $(find $OUT -name "*kernel_stats.csv"); do cp $f gpurun_out/c5_kernel_stats.csv; done
rm -rf $OUT
python - <<'PY' | tee -a gpurun_out/c5_pmc.txt
import csv, json, collections
def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    except OSError:
        pass
    return acc
sq, fe, wr = load("gpurun_out/c5_sq.csv"), load("gpurun_out/c5_fetch.csv"), load("gpurun_out/c5_write.csv")
out = {"source": "rocprofv3 --pmc (separate passes: SQ counters | FETCH_SIZE | WRITE_SIZE) and --kernel-trace --stats of `python scripts/c5_launch.py`; "
                 "FETCH_SIZE x2 (gfx950 wide-read correction), KB -> bytes; MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / ((GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs)"}
for tag, key, bpr in (("f32_mfma", "k_lidar_linearize_f32", 32), ("f64", "k_lidar_linearize<", 40)):
    ks = [k for k in sq if key in k]
    if not ks:
        continue
    k = ks[0]
    mean = lambda d, c: (sum(d[k][c]) / len(d[k][c])) if d.get(k) and d[k].get(c) else None
    e = {"kernel": k[:80], "launches": len(sq[k].get("GRBM_GUI_ACTIVE", []))}
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_MFMA", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"):
        e[c] = mean(sq, c)
    if e.get("GRBM_GUI_ACTIVE"):
        # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (1.59 M for an 84.6 us launch = 8 x 199 k cycles at
        # 2.35 GHz); MFMA busy cycles are summed over the 1024 SIMDs
        cyc = e["GRBM_GUI_ACTIVE"] / 8.0
        e["gpu_cycles_per_launch"] = cyc
        e["mfma_util_pct"] = 100.0 * (e["SQ_VALU_MFMA_BUSY_CYCLES"] or 0.0) / (cyc * 1024)
        e["valu_insts_per_wave64_of_residuals"] = (e["SQ_INSTS_VALU"] or 0.0) / (50 * 262144 / 64.0)
        e["mfma_flop_from_counter"] = (e["SQ_INSTS_VALU_MFMA_MOPS_F32"] or 0.0) * 512
    f = mean(fe, "FETCH_SIZE"); w = mean(wr, "WRITE_SIZE")
    e["fetch_bytes_per_launch"] = 2.0 * f * 1024 if f else None
    e["write_bytes_per_launch"] = w * 1024 if w else None
    e["algorithmic_bytes_per_launch"] = 50 * 262144 * bpr
    out[tag] = e
try:
    for row in csv.DictReader(open("gpurun_out/c5_kernel_stats.csv")):
        for tag, key in (("f32_mfma", "k_lidar_linearize_f32"), ("f64", "k_lidar_linearize<")):
            if key in row["Name"] and tag in out:
                out[tag]["trace_avg_ns"] = float(row["AverageNs"]); out[tag]["trace_calls"] = int(row["Calls"])
except OSError:
    pass
json.dump(out, open("gpurun_out/c5_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
