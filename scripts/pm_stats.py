"""what the insertions of a persistent map leave behind per frame (O3DS_PM_STATS, the A/B library): list sizes, index growth"""
import os, sys
os.environ["O3DS_PM_STATS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open3d_slam_amd import backend
scans = bench.make_stream(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
be = backend.Backend(0, ab=True)
out = bench.run_stream(be, scans)
print(out["scans_per_sec"], out["map_points"])
