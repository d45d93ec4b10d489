import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
mode = sys.argv[1]
from glio_amd import capi, synth, batch
if mode == "ctx_then_torch":
    o = synth.default_opts(4, pts=1024, map_pts=1024)
    c = capi.Context(o); c.close()
elif mode == "bassoc_then_torch":
    ba = batch.BatchAssociation(8, 4096, 400000); ba.close()
elif mode == "bassoc_open_then_torch":
    ba = batch.BatchAssociation(8, 4096, 400000)
import torch
try:
    x = torch.zeros(4, device="cuda:0"); print(mode, "torch ok", x.sum().item())
except Exception as e:
    print(mode, "torch FAILED:", str(e)[:80])
