#!/bin/bash
# A/B of library variants on the K2 workloads: scripts/k2_variants.sh <variant> [<variant> ...]   ("" = the product library)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = "prod" ]; then unset GLIO_HIP_LIB; else export GLIO_HIP_LIB=glio_amd/lib/libglio_hip_$v.so; fi
  echo "== $v"; python scripts/knn_ab.py 2>&1 | grep "^0 " | tail -1
done
