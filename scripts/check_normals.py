"""GPU normal estimation against the oracle on voxel-filtered OS-128-like scans: f64 storage must agree BIT FOR BIT on every
point (same neighbour set by (d2, index), cumulants summed in that order, the oracle's eigen-solver arithmetic); f32 storage is
compared by direction; every configuration is hashed twice around a device-pool disturbance (the index build's atomic scatter
then places the points of a cell in another order) to show that the result does not depend on it.  Prints time per call."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_amd import backend, synthetic as syn
from oracle import pyoracle as po

scene = syn.make_scene()
bad = 0
for name, pose in (("origin", syn.make_pose((0.0, 0.0, 0.5), (0.0, 0.0, 0.0))), ("offset", syn.make_pose((12.0, -7.0, 1.5), (1.0, -2.0, 40.0)))):
    scan = syn.os128_scan(scene, pose)
    for prec, pname in ((backend.PRECISION_F64, "f64"), (backend.PRECISION_F32, "f32")):
        be = backend.Backend(0, prec)
        c = be.upload(scan)
        v = be.voxel_down_sample(c, 0.1)
        pts = be.download(v)[0]
        for radius, knn in ((3.0, 20), (1.0, 5), (0.5, 30), (2.0, 48)):
            ts = []
            for rep in range(4):
                be.synchronize(); t0 = time.perf_counter(); be.estimate_normals(v, radius, knn); be.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            _, nrm = be.download(v)
            h1 = hashlib.sha1(np.ascontiguousarray(nrm).tobytes()).hexdigest()[:12]
            junk = [be.upload(np.random.default_rng(i).normal(size=(200_000 + 1000 * i, 3))) for i in range(3)]  # disturb the pool
            for j in junk:
                be.free(j)
            be.estimate_normals(v, radius, knn)
            _, nrm2 = be.download(v)
            h2 = hashlib.sha1(np.ascontiguousarray(nrm2).tobytes()).hexdigest()[:12]
            t0 = time.perf_counter(); ref = po.estimate_normals(pts, radius, knn); tc = (time.perf_counter() - t0) * 1e3
            neq = int(np.sum(np.any(nrm != ref, axis=1)))
            dots = np.einsum("ij,ij->i", nrm, ref)
            line = (f"{name} {pname} n={len(pts)} r={radius} knn={knn}: gpu min {min(ts):.3f} ms (cpu oracle {tc:.0f} ms)  sha {h1} / {h2} "
                    f"{'REPEATS' if h1 == h2 else 'DIFFERS'}  points != oracle: {neq}  |dot|<1-1e-6: {int(np.sum(np.abs(dots) < 1 - 1e-6))}  "
                    f"sign flips: {int(np.sum(dots < 0))}")
            print(line, flush=True)
            if h1 != h2 or (pname == "f64" and neq):
                bad += 1
                w = np.flatnonzero(np.any(nrm != ref, axis=1))[:5]
                for i in w:
                    print("   point", i, pts[i], "gpu", nrm[i], "ref", ref[i], "maxabs", np.abs(nrm[i] - ref[i]).max())
        be.close()
print("check_normals:", "OK" if bad == 0 else f"{bad} configurations FAILED")
