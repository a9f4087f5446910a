"""Why does the normal estimation of the shipped configuration run its density pilot every frame?  (A/B library, O3DS_NRM_DEBUG=1 prints every call.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["O3DS_NRM_DEBUG"] = "1"
import bench
from open3d_slam_amd import backend
_load = backend.load
backend.load = lambda ab=False: _load(True)
scans = bench.make_stream(12)
be = backend.Backend(0)
bench.run_stream(be, scans, shipped=("bench" not in sys.argv), profile=("profile" in sys.argv))
be.close()
