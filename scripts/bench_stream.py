#!/usr/bin/env python
"""M2 = scans/s of the full per-scan path (BASELINE.json configs[2]): OS-128-like 131 072-pt synthetic stream along a
figure-eight, LidarOdometry.addRangeScan + Mapper.addRangeMeasurement (crop -> voxel 0.1 -> normals knn 20 / r 3 -> narrow
crop -> scan-to-map ICP with the default convergence criteria -> transform + append + re-voxelize the submap + index rebuild),
on the device through the reference-named host classes, and the same loop on the CPU oracle for a few frames."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--cpu-frames", type=int, default=6)
    ap.add_argument("--n-az", type=int, default=1024)
    ap.add_argument("--max-iter", type=int, default=50)
    args = ap.parse_args()
    from open3d_slam_amd import backend, parameters as P, synthetic as syn
    from open3d_slam_amd.mapper import Mapper
    from open3d_slam_amd.odometry import LidarOdometry
    from open3d_slam_amd.pointcloud import PointCloud

    mp = P.lua_default_mapper_parameters()
    mp.scanMatcher_.icp_.maxNumIter_ = args.max_iter
    op = P.OdometryParameters()
    op.scanMatcher_.icp_ = P.IcpParameters(maxNumIter_=args.max_iter, maxCorrespondenceDistance_=1.0, knn_=20, maxDistanceKnn_=3.0)
    op.scanProcessing_.voxelSize_ = 0.1
    op.scanProcessing_.cropper_ = P.ScanCroppingParameters(croppingMinRadius_=2.0, croppingMaxRadius_=30.0, cropperName_="MinMaxRadius")
    scene = syn.make_scene()
    poses = syn.figure_eight_poses(200, 0.1)
    # a lidar driver delivers float32 x/y/z (sensor_msgs/PointCloud2): both loops see those values, the device ingests them directly
    scans32 = [syn.os128_scan(scene, poses[k], frame=k, n_az=args.n_az).astype(np.float32) for k in range(args.frames)]
    scans = [s.astype(np.float64) for s in scans32]

    be = backend.Backend(0)
    odo = LidarOdometry(be)
    odo.setParameters(op)
    mapper = Mapper(be, odo)
    mapper.setParameters(mp)
    stage = {"upload": 0.0, "odometry": 0.0, "mapping": 0.0}
    t_all = time.perf_counter()
    for k, raw in enumerate(scans32):
        t0 = time.perf_counter()
        cloud = PointCloud.from_pointcloud2(be, raw)
        t1 = time.perf_counter()
        ok1 = odo.addRangeScan(cloud, 0.1 * k)
        be.synchronize()
        t2 = time.perf_counter()
        ok2 = mapper.addRangeMeasurement(cloud, 0.1 * k)
        be.synchronize()
        t3 = time.perf_counter()
        cloud.release()
        assert ok1 and ok2, (k, ok1, ok2)
        if k > 0:  # frame 0 only initialises
            stage["upload"] += t1 - t0
            stage["odometry"] += t2 - t1
            stage["mapping"] += t3 - t2
    wall = time.perf_counter() - t_all
    n = args.frames - 1
    T_gt = np.linalg.inv(poses[0]) @ poses[args.frames - 1]
    dt, dr = syn.se3_error(mapper.getMapToRangeSensor(), T_gt)
    out = {"workload": f"configs[2]: OS-128-like stream, {len(scans[0])} raw pts/scan, {args.frames} frames, voxel 0.1, knn 20/r 3, map voxel 0.1",
           "gpu_scans_per_sec_mapping_only": n / stage["mapping"], "gpu_scans_per_sec_odometry_plus_mapping": n / (stage["odometry"] + stage["mapping"] + stage["upload"]),
           "gpu_ms_per_scan": {k: 1e3 * v / n for k, v in stage.items()}, "map_points": len(mapper.getActiveSubmap().getMapPointCloud()),
           "final_pose_error_vs_truth": {"dt_m": dt, "dr_rad": dr}, "wall_s": wall}
    if args.cpu_frames > 1:
        import test_pipeline_gpu as tp
        from oracle import pyoracle as po

        po.lib().orc_set_num_threads(min(32, os.cpu_count() or 1))  # best thread count of the CPU sweep on the MI355X host
        ref = tp._OracleLoop(po, mp, op)
        cpu = {"odometry": 0.0, "mapping": 0.0}
        for k in range(args.cpu_frames):
            t0 = time.perf_counter()
            ref.odometry(scans[k], 0.1 * k)
            t1 = time.perf_counter()
            ref.mapping(scans[k], 0.1 * k)
            t2 = time.perf_counter()
            if k > 0:
                cpu["odometry"] += t1 - t0
                cpu["mapping"] += t2 - t1
        m = args.cpu_frames - 1
        out["cpu_scans_per_sec_mapping_only"] = m / cpu["mapping"]
        out["cpu_scans_per_sec_odometry_plus_mapping"] = m / (cpu["odometry"] + cpu["mapping"])
        out["cpu_ms_per_scan"] = {k: 1e3 * v / m for k, v in cpu.items()}
        out["cpu_threads"] = po.lib().orc_num_threads()
        out["cpu_kind"] = "CPU restatement of Open3D v0.15.1 (oracle), same orchestration"
        dt, dr = syn.se3_error(np.linalg.inv(poses[0]) @ poses[args.cpu_frames - 1], ref.T)
        out["cpu_pose_error_vs_truth"] = {"dt_m": dt, "dr_rad": dr}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
