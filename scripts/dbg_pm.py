import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_preprocess_map_gpu as t
from open3d_slam_amd import backend
which = sys.argv[1] if len(sys.argv) > 1 else "big"
try:
    if which == "big":
        be = backend.Backend(0, ab=True)
        print([x[-1] if isinstance(x, tuple) else x for x in t._big_map_inserts(be)][:3])
    else:
        be = backend.Backend(0, backend.PRECISION_F64)
        out = t._insert_sequence(be, 24, 12.0, 0.2, look_at=(3, 4, 11, 17, 23), carve_at=(14,))
        print([o[2] for o in out])
except Exception:
    traceback.print_exc()
