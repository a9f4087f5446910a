import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from open3d_slam_amd import backend, parameters as P, synthetic as syn
from open3d_slam_amd.mapper import Mapper
from open3d_slam_amd.odometry import LidarOdometry
from open3d_slam_amd.pointcloud import PointCloud
from oracle import pyoracle as po
from scipy.spatial import cKDTree
import test_pipeline_gpu as tp

be = backend.Backend(0, backend.PRECISION_F64)
mp, op = tp._params()
scene = syn.make_scene()
poses = syn.figure_eight_poses(200, 0.1)[:3]
odo = LidarOdometry(be); odo.setParameters(op)
mapper = Mapper(be, odo); mapper.setParameters(mp)
ref = tp._OracleLoop(po, mp, op)
for k in range(3):
    raw = syn.os128_scan(scene, poses[k], frame=k, n_az=tp.N_AZ)
    t = 0.1 * k
    cloud = PointCloud.from_numpy(be, raw)
    # device-side inputs of this frame's scan-to-map ICP, captured BEFORE the mapper mutates the map
    mapc = mapper.getActiveSubmap().getMapPointCloud()
    dev_map_p, dev_map_n = (mapc.points_, mapc.normals_) if len(mapc) else (np.zeros((0, 3)), np.zeros((0, 3)))
    odo.addRangeScan(cloud, t)
    ref.odometry(raw, t)
    if k > 0:
        d, j = cKDTree(dev_map_p).query(ref.map_p)
        print(f"frame {k}: map before ICP: sizes {len(dev_map_p)}/{len(ref.map_p)} max point diff {d.max():.3e}; normal diff "
              f"{np.abs(dev_map_n[j] - ref.map_n).max():.3e}; #normals differing >1e-9: {(np.abs(dev_map_n[j]-ref.map_n).max(1) > 1e-9).sum()}")
    mapper.addRangeMeasurement(cloud, t)
    ref.mapping(raw, t)
    if k > 0:
        print("   device iterations/converged:", mapper.lastResult_.iterations_, mapper.lastResult_.converged_, "fitness", mapper.lastResult_.fitness_)
    dt, dr = syn.se3_error(mapper.getMapToRangeSensor(), ref.T)
    print(f"frame {k}: pose diff {dt:.3e} {dr:.3e}")
