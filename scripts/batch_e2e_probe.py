import sys, json
sys.path.insert(0, '.')
import torch, bench
for K, pts in ((200, 8192), (2000, 32768)):
    r = bench.bench_batch_end_to_end(0, torch, K, pts=pts, projection={"projected_ms_per_group": 0.74, "assumed_collective": "test"})
    print(json.dumps(r)[:1500])
    print(torch.cuda.max_memory_allocated() / 1e9, "GB torch;", flush=True)
