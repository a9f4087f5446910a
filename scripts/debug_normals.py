import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_amd import backend, synthetic as syn
from oracle import pyoracle as po
from scipy.spatial import cKDTree
scene = syn.make_scene()
scan = syn.os128_scan(scene, np.eye(4), n_az=256)
pts = po.voxel_down_sample(scan, 0.1)
be = backend.Backend(0, backend.PRECISION_F64)
c = be.upload(pts)
radius, knn = 0.3, 30
be.estimate_normals(c, radius, knn)
_, got = be.download(c)
ref = po.estimate_normals(pts, radius, knn)
dots = np.einsum("ij,ij->i", got, ref)
bad = np.where(dots < 1 - 1e-9)[0]
print("n", len(pts), "bad", len(bad))
tree = cKDTree(pts)
for i in bad[:12]:
    d, j = tree.query(pts[i], k=knn, distance_upper_bound=radius)
    ok = np.isfinite(d) & (d * d < radius * radius)
    nb = pts[j[ok]]
    mu = nb.mean(0); cov = nb.T @ nb / len(nb) - np.outer(mu, mu)
    w = np.linalg.eigvalsh(cov) if len(nb) >= 3 else [0, 0, 0]
    print(i, "nb", len(nb), "eig", np.array(w), "dot", dots[i], "got", got[i], "ref", ref[i], "maxd", d[ok].max() if ok.any() else None)
