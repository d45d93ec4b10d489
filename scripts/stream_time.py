"""The moving-stream keyframe cycle of bench.py on its own (for rocprofv3 --kernel-trace --stats around it)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
print(json.dumps(bench.bench_keyframe_stream(0, 20, 65536, n_keyframes=n, cpu_keyframes=0)))
