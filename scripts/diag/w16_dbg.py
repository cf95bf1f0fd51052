import os, sys
sys.path.insert(0, os.getcwd())
import torch
from nerf_sos_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (R, S) in [(1, 16), (2, 16), (16, 16), (64, 64), (700, 64), (8192, 64)]:
    P = R * S
    w = torch.rand(R, S, device=dev) / S
    g = torch.randn(R, 2, device=dev) * 1e-4
    w2 = torch.randn(2, 128, device=dev) * 0.1
    hid = torch.relu(torch.randn(P, 128, device=dev))
    x = torch.randn(P, 320, device=dev).to(torch.float16)
    a = ops.sem_head_wgrad(w, g, w2, hid, x, split_fp16=True)
    torch.cuda.synchronize()
    b = ops.sem_head_wgrad(w, g, w2, hid, x.float(), split_fp16=True)
    torch.cuda.synchronize()
    print(R, S, "ok", float((a[0] - b[0]).abs().max() / b[0].abs().max()), flush=True)
