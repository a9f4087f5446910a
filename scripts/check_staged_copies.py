"""Round trips through the staged (pinned-ring) host <-> device copies: every value must come back exactly, for sizes around the
chunk boundaries, many times back to back (a ring-reuse race would show up as a few wrong bytes once in a while)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import backend
be = backend.Backend(0, backend.PRECISION_F32)
be64 = backend.Backend(0, backend.PRECISION_F64)
rng = np.random.default_rng(0)
bad = 0
for it in range(300):
    n = int(rng.choice([8191, 65536, 65537, 131072, 131073, 200001, 43690, 43691]))
    rec = rng.uniform(-50, 50, size=(n, 4)).astype(np.float32)
    c = be.upload_f32(rec, 0, 4, 8)
    back = be.download_f32(c, 16, 0, 4, 8, None)
    got = np.frombuffer(np.ascontiguousarray(back).tobytes(), dtype=np.float32).reshape(n, 4)[:, :3]
    if not np.array_equal(got, rec[:, :3]):
        bad += 1
        print("f32 mismatch it", it, "n", n, int((got != rec[:, :3]).sum()))
    be.free(c)
    pts = rng.uniform(-50, 50, size=(n, 3))
    nrm = rng.normal(size=(n, 3))
    c = be64.upload(pts, nrm)
    p2, n2 = be64.download(c)
    if not (np.array_equal(p2, pts) and np.array_equal(n2, nrm)):
        bad += 1
        print("f64 mismatch it", it, "n", n)
    be64.free(c)
print("staged copy round trips done, mismatches:", bad)
