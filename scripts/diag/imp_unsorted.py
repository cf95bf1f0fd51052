"""Which rays of the unsorted-depth importance test differ from torch.sort, and how."""
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nerf_sos_amd
from nerf_sos_amd import ops
DEV = "cuda:0"
rng = np.random.default_rng(23)
for S, n_imp in ((64, 128), (17, 40), (200, 64)):
    R = 37
    z = (1.2 + 13 * rng.random((R, S), dtype=np.float32))
    z[: R // 3] = np.sort(z[: R // 3], -1)[:, ::-1]
    z[R // 3: 2 * R // 3] = np.sort(z[R // 3: 2 * R // 3], -1)
    z[5, 3] = np.nan
    w = rng.random((R, S), dtype=np.float32) ** 4
    for uu in (None, rng.random((R, n_imp), dtype=np.float32)):
        zt, wt = torch.from_numpy(z).to(DEV), torch.from_numpy(w).to(DEV)
        zf, zs, _ = ops.importance_sample(zt, wt, n_imp, None if uu is None else torch.from_numpy(uu).to(DEV))
        want = torch.sort(torch.cat([zt, zs], -1), -1).values
        a, b = torch.nan_to_num(zf, nan=-1.0).cpu(), torch.nan_to_num(want, nan=-1.0).cpu()
        bad = (a != b).any(-1).nonzero().flatten().tolist()
        print(S, n_imp, uu is None, "bad rows", bad)
        for r in bad[:3]:
            k = (a[r] != b[r]).nonzero().flatten().tolist()
            print("  row", r, "cols", k[:10], "got", a[r][k[:6]].tolist(), "want", b[r][k[:6]].tolist(), "nan in zs", bool(torch.isnan(zs[r]).any()),
                  "sorted-got", bool((np.diff(a[r].numpy()) >= 0).all()))
