cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_assoc.py tests/test_hip_c3.py tests/test_hip_bassoc.py tests/test_golden.py -x -q 2>&1 | tail -3
python scripts/knn_ab.py 2>&1 | grep "^0 " | tail -2
GLIO_HIP_LIB=glio_amd/lib/libglio_hip_stamps.so python scripts/knn_wg_times.py 2>&1 | tail -9
