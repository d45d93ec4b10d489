#!/bin/bash
# headline solves/s with and without the helper workgroups of k_chain_step (GLIO_CHAIN_HELPERS), alternating
cd $GRAFT_REPO_ROOT
for v in 1 0 1 0 1 0; do
  GLIO_CHAIN_HELPERS=$v python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-c5 --no-bassoc --no-batch 2>/dev/null > /tmp/l.json
  python -c "
import json; d=json.loads(open('/tmp/l.json').readline()); print('helpers $v', d['value'], d['ms_per_step'], d['kernels_us']['tr_step'])"
done
