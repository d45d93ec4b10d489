"""glio_bassoc_create at C4 size (2000 keyframes x 32768 points): wall time, with / without the low stream priority"""
import sys, time
sys.path.insert(0, ".")
from glio_amd import batch
for _ in range(2):
    t0 = time.perf_counter()
    ba = batch.BatchAssociation(2000, 32768, 24000 * 4096)
    t1 = time.perf_counter()
    ba.close()
    print("create ms", round((t1 - t0) * 1e3, 1), "destroy ms", round((time.perf_counter() - t1) * 1e3, 1), flush=True)
