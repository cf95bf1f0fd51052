import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda:0")
for i in range(3):
    print(bench.hbm_kernel_rooflines(torch, dev, 65536))
