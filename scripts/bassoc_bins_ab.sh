#!/bin/bash
# batch association with and without the shared query binning (GLIO_BASSOC_SHARED_BINS): the per-keyframe call (12 pairs of 64 k queries) and 192 pairs
cd "$(dirname "$0")/.."
for v in 1 0 1 0; do
  GLIO_BASSOC_SHARED_BINS=$v python - <<'PY'
import os, json, sys
sys.path.insert(0, os.getcwd())
import bench
r = bench.bench_batch_association(0)
print("shared_bins", os.environ["GLIO_BASSOC_SHARED_BINS"], json.dumps({k: r[k] for k in r if k not in ("workload", "rounds_with_reassociation")})[:600])
PY
done
