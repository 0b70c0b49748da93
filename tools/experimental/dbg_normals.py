import os, sys
sys.path.insert(0, "mv-lm-icp_amd"); sys.path.insert(0, "tests")
import numpy as np, mvicp
K = np.load("tests/golden/bunny_knn.npz"); pts, gi, gd = K["pts"], K["knn_idx"], K["knn_d2"]
eng = mvicp.Engine(0); eng.set_frames([pts], None)
nrm, knn = eng.recompute_normals(0, 10, want_knn=True)
bad = np.where(~np.all(knn == gi, axis=1))[0]
print("rows", len(pts), "differing", len(bad), "set-differing", int((~np.all(np.sort(knn,1)==np.sort(gi,1),1)).sum()))
for i in bad[:6]:
    print(i, knn[i].tolist(), gi[i].tolist(), [float(x) for x in gd[i]])
    print("   pts", pts[i].tolist())
eng.close()
