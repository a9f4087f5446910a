"""Development aid: one small registration per (precision, candidate sets on/off), poses and per-launch counters side by side."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import numpy as np
    from open3d_slam_amd import backend, synthetic as syn
    prec = int(sys.argv[1])
    src, tgt, nrm, _ = syn.config2_inputs(n_map=100_000, n_az=512)
    be = backend.Backend(0, prec)
    for it in (0, 1, 2, 3, 10):
        r = be.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=it, rel_fitness=0.0, rel_rmse=0.0)
        print("prec", prec, "sets", os.environ.get("O3DS_ICP_SETS"), "iters", it, "T[:3,3]", r["transformation"][:3, 3], "fit", r["fitness"], "rmse", r["inlier_rmse"], flush=True)
    be.close()
else:
    for prec in (0, 1):
        for sets in ("0", "1"):
            env = dict(os.environ, O3DS_ICP_SETS=sets, O3DS_ICP_STATS="1")
            subprocess.run([sys.executable, __file__, str(prec)], env=env)
