cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
bash scripts/knn_pmc.sh > gpurun_out/r04/k2_pmc.txt 2>&1
GLIO_HIP_LIB=glio_amd/lib/libglio_hip_stamps.so python scripts/knn_stamps.py > gpurun_out/r04/k2_stamps.txt 2>&1
OUT=/tmp/knnprof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python scripts/knn_prof.py 0 > gpurun_out/r04/k2_prof.log 2>&1
for f in $(find $OUT -name "*kernel_stats.csv"); do cp $f gpurun_out/r04/k2_kernel_stats.csv; done
python scripts/knn_ab.py > gpurun_out/r04/k2_ab.txt 2>&1
tail -5 gpurun_out/r04/k2_stamps.txt; head -20 gpurun_out/r04/k2_kernel_stats.csv; tail -12 gpurun_out/r04/k2_ab.txt
