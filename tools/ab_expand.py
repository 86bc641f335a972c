"""mask_expand under different edge densities / clusterings (run under rocprofv3 --kernel-trace)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poppunk_amd import engine
rows = 49995000
d = (torch.rand((rows, 2), device="cuda") * 0.3).contiguous()
cases = {"none": 1e-6, "scattered_0.2pct": 0.02, "scattered_5pct": 0.1, "half": 0.3}
for name, t in cases.items():
    for _ in range(5):
        e = engine.edge_threshold_dev(d, 2, t, t, cap=rows)
    torch.cuda.synchronize()
    print(name, "edges", e.shape[0])
# clustered: the first 1 % of rows all within
d2 = d.clone(); d2[:] = 1.0; d2[: rows // 100] = 0.0
for _ in range(5):
    e = engine.edge_threshold_dev(d2, 2, 0.02, 0.02, cap=rows)
torch.cuda.synchronize()
print("clustered_1pct edges", e.shape[0])
