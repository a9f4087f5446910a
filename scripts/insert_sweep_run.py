import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
scans = bench.make_stream(200)
print(bench.run_insert_sweep(0, scans))
