#!/bin/bash
# round-5 evidence in one call: the round profile (tests, bench line, kernel stats, K3 traffic) + the K2 window-call profile, counters and statistics
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/gpu_round_profile.sh 2>&1 | tail -12
(cd /tmp; bash $GRAFT_REPO_ROOT/scripts/k2_prof_window.sh prod 2>&1 | tail -14)
KNN_WINDOW=1 bash scripts/knn_pmc.sh > gpurun_out/k2_window_pmc.txt 2>&1
KNN_WINDOW=1 bash scripts/knn_pmc2.sh >> gpurun_out/k2_window_pmc.txt 2>&1
python scripts/knn_stats.py > gpurun_out/k2_near_stats.txt 2>&1
python scripts/knn_ab.py > gpurun_out/k2_ab.txt 2>&1
tail -4 gpurun_out/k2_ab.txt; tail -5 gpurun_out/k2_near_stats.txt | cut -c1-300
