import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sparse_matrix_with_flops_amd import hipspgemm as hs
mode = sys.argv[1]
print("env", {k: v for k, v in os.environ.items() if "VISIBLE" in k or "HSA" in k or "HIP" in k})
if mode == "torch_first":
    import torch
    print("torch first:", torch.cuda.is_available(), torch.cuda.device_count())
h = hs.Handle(0)
h.selftest()
print("hs ok")
if mode == "close_then_torch":
    h.close()
import torch
try:
    print("torch after hs:", torch.cuda.is_available(), torch.cuda.device_count())
    torch.cuda.set_device(0)
    x = torch.zeros(4, device="cuda")
    print("alloc ok")
except Exception as e:
    print("torch failed:", repr(e)[:200])
