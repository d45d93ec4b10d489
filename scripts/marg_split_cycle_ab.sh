#!/bin/bash
# the keyframe cycle (default order and deferred batch association) with the three-launch marginalization and with the one-workgroup kernel (GLIO_MARG_SPLIT=0)
cd "$(dirname "$0")/.."
for i in 1 2; do for sp in 1 0; do
  SCM_REPS=4 GLIO_MARG_SPLIT=$sp python scripts/stream_cpp_modes.py 2>/dev/null > /tmp/scm_$sp.txt
  python - $sp <<'PY'
import sys, json
sp = sys.argv[1]
rows = [json.loads(l) for l in open(f"/tmp/scm_{sp}.txt") if l.startswith("{")]
for d in (False, True):
    v = [round(r["cycle_ms"], 3) for r in rows if r["deferred"] == d]; m = [r["stages"]["marginalize"] for r in rows if r["deferred"] == d]
    print("split", sp, "deferred", d, "cycle", v, "marginalize stage", m)
PY
done; done
