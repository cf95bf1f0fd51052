"""nsos_wgrad against fp64 GEMMs for every (M, N) tile pair it accepts, on column slices of wider buffers, ragged point counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nerf_sos_amd
from nerf_sos_amd import ops

dev = "cuda:0"
torch.manual_seed(0)
worst = 0.0
for P in (1, 7, 384, 777, 5000):
    G = torch.randn(P, 992, device=dev)
    X = torch.randn(P, 992, device=dev)
    for M in (32, 64, 128, 256):
        for N in (32, 64, 128, 256):
            dW = torch.zeros(M + 32, N + 64, device=dev)
            db = torch.zeros(M, device=dev)
            g, x = G[:, 96:96 + M], X[:, 160:160 + N]
            ops.wgrad(g, x, dW[16:16 + M, 32:32 + N], db)
            want = (g.double().T @ x.double())
            e = float((dW[16:16 + M, 32:32 + N].double() - want).abs().max() / (want.abs().max() + 1e-30))
            eb = float((db.double() - g.double().sum(0)).abs().max() / (g.double().sum(0).abs().max() + 1e-30))
            worst = max(worst, e, eb)
            if e > 1e-5 or eb > 1e-5:
                print("BAD", P, M, N, e, eb)
print("worst", worst)
