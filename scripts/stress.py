"""Back-to-back calls of every preprocessing / map op with varying sizes: results must be identical run to run -- points and
sizes exactly, and since round 2 the normals too (the kernel ranks neighbours by a total order); the script still reports
last-bit differences of normals separately from real mismatches, which makes a regression of that property visible as such."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_amd import backend, synthetic as syn
scene = syn.make_scene()
be = backend.Backend(0)
poses = syn.figure_eight_poses(200, 0.1)
scans = [syn.os128_scan(scene, poses[k], frame=k, n_az=1024 if k % 2 == 0 else 700) for k in range(6)]
ref = {}
bad = 0
tiny = 0
for rep in range(8):
    for k, raw in enumerate(scans):
        c = be.upload(raw)
        cr = be.crop_cloud(c, backend.make_crop(backend.CROP_MIN_MAX_RADIUS, rmin=2.0, rmax=30.0))
        v = be.voxel_down_sample(cr, 0.1)
        be.estimate_normals(v, 3.0, 20)
        t = be.transform_cloud(v, poses[k])
        m = be.upload(np.zeros((0, 3)))
        be.map_insert_scan(m, v, poses[k], 0.1, backend.make_crop(backend.CROP_MIN_MAX_RADIUS, center=poses[k][:3, 3], rmin=2.0, rmax=30.0), 1.0)
        be.map_insert_scan(m, t, np.eye(4), 0.1, backend.make_crop(backend.CROP_MIN_MAX_RADIUS, center=poses[k][:3, 3], rmin=2.0, rmax=30.0), 1.0)
        outs = [be.download(x) for x in (cr, v, t, m)]
        key = k
        sig = [(a.shape, float(np.nansum(a)), None if b is None else float(np.nansum(b))) for a, b in outs]
        if key not in ref:
            ref[key] = (sig, outs)
        else:
            for j, ((a, b), (ra, rb)) in enumerate(zip(outs, ref[key][1])):
                same = a.shape == ra.shape and np.array_equal(a, ra) and ((b is None and rb is None) or np.array_equal(b, rb, equal_nan=True))
                close = (not same) and a.shape == ra.shape and np.array_equal(a, ra) and b is not None and rb is not None and b.shape == rb.shape \
                    and int((np.abs(b - rb).max(1) > 1e-6).sum()) <= 3
                if close:
                    tiny += 1
                elif not same:
                    bad += 1
                    dp = np.abs(a - ra).max() if a.shape == ra.shape else -1
                    dn = np.nanmax(np.abs(b - rb)) if (b is not None and b.shape == rb.shape) else -1
                    nbad = int((np.abs(b - rb).max(1) > 1e-6).sum()) if (b is not None and b.shape == rb.shape) else -1
                    print("MISMATCH rep", rep, "scan", k, "stage", ["crop", "voxel+normals", "transform", "map"][j], a.shape, "max|dpts|", dp, "max|dnrm|", dn, "#nrm>1e-6", nbad)
        for x in (c, cr, v, t, m):
            be.free(x)
print("done, mismatches:", bad, " clouds whose normals differ in the last bits (default mode, see docstring):", tiny)
