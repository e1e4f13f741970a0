import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vipcup_amd  # noqa
from vipcup_amd import ops
M, C, hid = 2509056, 96, 384
x = torch.randn((M, C), dtype=torch.float16, device="cuda")
fc1 = ops.make_dense_weight(torch.randn(C, hid) * 0.1, torch.zeros(hid))
fc2 = ops.make_dense_weight(torch.randn(hid, C) * 0.05, torch.zeros(C))
for _ in range(4):
    ops.mlp(x, fc1, fc2, act="gelu", residual=x)
torch.cuda.synchronize()
