"""Phase trace (O3DS_FUSED_TRACE) and query statistics (O3DS_ICP_STATS) of the configs[1] registration, launches 0..L, on an A/B library
(O3DS_BACKEND_LIB, built with -DO3DS_AB_SWITCHES): python scripts/icp_trace.py [launches...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import numpy as np
    from open3d_slam_amd import backend, synthetic as syn
    scene = syn.make_scene(); T_gt = syn.ground_truth_pose()
    src = syn.vlp16_scan(scene, T_gt); tgt, nrm = syn.sample_map(scene, int(os.environ.get("N_MAP", "1000000")), seed=syn.SEED_MAP)
    be = backend.Backend(0)
    s, t = be.upload(src), be.upload(tgt, nrm)
    be.build_index(t, 1.0, 0.0)
    if os.environ.get("METHOD") == "gicp":
        be.estimate_normals(s, 3.0, 20)
    for _ in range(3):
        if os.environ.get("METHOD") == "gicp":
            r = be.icp_generalized_dev(s, t, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
        else:
            r = be.icp_point_to_plane_dev(s, t, 1.0, max_iter=10, rel_fitness=0.0, rel_rmse=0.0)
    be.close()
    sys.exit(0)
for L in (sys.argv[1:] or ["0", "1", "5"]):
    path = "/tmp/trace_%s.txt" % L
    env = dict(os.environ, O3DS_FUSED_TRACE=path, O3DS_FUSED_TRACE_LAUNCH=L, **({"O3DS_ICP_STATS": "1"} if os.environ.get("STATS") else {}))
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True)
    st = [l for l in p.stderr.splitlines() if l.startswith("icp stats")]
    print("==== launch", L, "lib", os.path.basename(os.environ.get("O3DS_BACKEND_LIB", "default")))
    if L == (sys.argv[1:] or ["0"])[0] and st:
        print(st[-1])
    if p.returncode != 0:
        print(p.stderr[-2000:])
        continue
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fused_trace.py"), path], capture_output=True, text=True).stdout
    print("\n".join(l for l in out.splitlines() if "inside the body" in l or "body->" in l or "stepped->" in l or "loaded->" in l or l.startswith("  end") or "start->" in l))
