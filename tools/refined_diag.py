import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import harness as H, atom_amd
import test_gpu_refined as R
o, d, names = R._pair(H.HYD, 301)
o.op("hyd_rk3d"); d.op("hyd_rk3d")
for k in names:
    if k == "RESIDUUM": continue
    a, b = d.download(k), o.f[k]
    neq = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    print(k, "differ", int(neq.sum()), "nan dev/oracle", int(np.isnan(a).sum()), int(np.isnan(b).sum()), "inf", int(np.isinf(a).sum()), int(np.isinf(b).sum()),
          "max|oracle|", float(np.nanmax(np.abs(b))))
    if neq.any():
        idx = np.argwhere(neq)
        print("   levels", np.unique(idx[:, 0])[:10], "rows", idx[:, 1].min(), idx[:, 1].max(), "cols", idx[:, 2].min(), idx[:, 2].max())
        i, j, kk = idx[0]
        print("   first", (i, j, kk), a[i, j, kk], b[i, j, kk])
