import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_amd import backend, parameters as P, synthetic as syn
from open3d_slam_amd.mapper import Mapper
from open3d_slam_amd.odometry import LidarOdometry
from open3d_slam_amd.pointcloud import PointCloud
mp = P.lua_default_mapper_parameters(); mp.scanMatcher_.icp_.maxNumIter_ = 50
op = P.OdometryParameters(); op.scanMatcher_.icp_ = P.IcpParameters(maxNumIter_=50, maxCorrespondenceDistance_=1.0, knn_=20, maxDistanceKnn_=3.0)
op.scanProcessing_.voxelSize_ = 0.1
op.scanProcessing_.cropper_ = P.ScanCroppingParameters(croppingMinRadius_=2.0, croppingMaxRadius_=30.0, cropperName_="MinMaxRadius")
scene = syn.make_scene(); poses = syn.figure_eight_poses(200, 0.1)
be = backend.Backend(0); odo = LidarOdometry(be); odo.setParameters(op); mapper = Mapper(be, odo); mapper.setParameters(mp)
for k in range(14):
    raw = syn.os128_scan(scene, poses[k], frame=k, n_az=1024)
    cloud = PointCloud.from_numpy(be, raw)
    ok1 = odo.addRangeScan(cloud, 0.1 * k); ok2 = mapper.addRangeMeasurement(cloud, 0.1 * k)
    r = mapper.lastResult_
    m = mapper.getActiveSubmap().getMapPointCloud()
    if os.environ.get("NODL"):
        mx = mn = np.zeros((len(m), 3))
    else:
        mx, mn = m.points_, m.normals_
    T_gt = np.linalg.inv(poses[0]) @ poses[k]
    print(k, ok1, ok2, None if r is None else (round(r.fitness_, 4), round(r.inlier_rmse_, 4), r.iterations_, r.converged_), "map", len(mx),
          "nan pts", int(np.isnan(mx).any(1).sum()), "nan nrm", int(np.isnan(mn).any(1).sum()), "err", syn.se3_error(mapper.getMapToRangeSensor(), T_gt))
    cloud.release()
