#!/bin/bash
# HBM traffic of the dominant kernel (K3) from the TCC fabric counters, one --pmc pass per counter
# (MI355X_MICROARCH.md: FETCH_SIZE costs 3 of 4 TCC slots; on gfx950 it reports 1/2 of a wide coalesced read).
# Writes gpurun_out/k3_pmc.json + gpurun_out/k3_pmc.txt; copy both into profiles/ to have bench.py quote `traffic`.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
: > gpurun_out/k3_pmc.txt
for CNT in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$CNT
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc $CNT --output-format csv -d $OUT -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-c5 --no-bassoc > gpurun_out/pmc_$CNT.log 2>&1
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  python - "$f" $CNT <<'PY' | tee -a gpurun_out/k3_pmc.txt
import csv, sys, collections, json, os
f, cnt = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
with open(f) as fh:
    for row in csv.DictReader(fh):
        if row.get("Counter_Name") == cnt:
            acc[row["Kernel_Name"][:48]].append(float(row["Counter_Value"]))
res = {}
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:60]:
    big = [x for x in v if x > 0.5 * max(v)]
    print(f"{cnt} {k:48s} launches {len(v):4d} max {max(v):12.1f} KB  mean_of_full_launches {sum(big)/len(big):12.1f} KB")
    res[k] = v
path = "gpurun_out/k3_pmc_raw.json"
d = json.load(open(path)) if os.path.exists(path) else {}
d[cnt] = res
json.dump(d, open(path, "w"), indent=1)
PY
  rm -rf $OUT
done
python - <<'PY'
import json
raw = json.load(open("gpurun_out/k3_pmc_raw.json"))
line = None
for ln in open("gpurun_out/pmc_FETCH_SIZE.log"):
    if ln.startswith("{"):
        line = json.loads(ln)
alg = line["roofline"]["bytes_per_launch"] if line else None
# every launch of the solver's K3 variant (the bench also runs the kernel once on a 10x larger window: keep the launches whose
# corrected read size is within 25 % of the C2 workload's algorithmic bytes)
k3v = [x for k, v in raw["FETCH_SIZE"].items() if "k_lidar_linearize" in k for x in v if alg and 0.75 < 2.0 * x * 1024 / alg < 1.25]
wrv = [x for k, v in raw.get("WRITE_SIZE", {}).items() if "k_lidar_linearize" in k for x in v if x * 1024 < 1e6]
fetch = 2.0 * (sum(k3v) / len(k3v)) * 1024 if k3v else None          # KB -> B, x2: gfx950 reports half of a wide coalesced read
# the launch the solve itself uses (K3 workgroups + small factors): k_linearize_all at the C2 size
lav = [x for k, v in raw["FETCH_SIZE"].items() if "k_linearize_all" in k for x in v if alg and 0.75 < 2.0 * x * 1024 / alg < 1.6]
law = [x for k, v in raw.get("WRITE_SIZE", {}).items() if "k_linearize_all" in k for x in v if x * 1024 < 4e6]
la_fetch = 2.0 * (sum(lav) / len(lav)) * 1024 if lav else None
la_write = (sum(law) / len(law)) * 1024 if law else 0.0
wr = [sum(wrv) / len(wrv)] if wrv else []
out = {"lidar_residuals": line["config"]["lidar_residuals"] if line else None, "launches_averaged": len(k3v),
       "k3_fetch_bytes_per_launch": fetch, "k3_write_bytes_per_launch": (wr[0] * 1024 if wr else None),
       "k3_hbm_bytes_per_launch": (fetch + (wr[0] * 1024 if wr else 0.0)) if fetch else None,
       "algorithmic_bytes_per_launch": line["roofline"]["bytes_per_launch"] if line else None,
       "linearize_all_launches_averaged": len(lav), "linearize_all_fetch_bytes_per_launch": la_fetch,
       "linearize_all_hbm_bytes_per_launch": (la_fetch + la_write) if la_fetch else None,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 3 --warmup 1 "
                 "--no-cpu-baseline --no-batch --no-c5 --no-bassoc`; FETCH_SIZE x2 (gfx950 wide-read correction), KB -> bytes"}
json.dump(out, open("gpurun_out/k3_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
