"""open3d_slam's OWN Odometry.cpp / Mapper.cpp / ScanToMapRegistration.cpp / Submap.cpp / SubmapCollection.cpp with
integration/open3d_slam_o3ds.patch applied, compiled (oracle/ref_build, where the checkout is) and RUN here against libo3ds_backend.so:
the patched reference driving the MI355X -- what a maintainer who applies the patch gets, as far as this image allows (Eigen and the
open3d::geometry::PointCloud container are stand-ins, oracle/ref_build/shim; every Open3D ALGORITHM on the path is replaced by the
patch and runs on the device).  Checked against (a) the device-resident Python loop, (b) the CPU oracle loop -- which itself agrees with
the UNPATCHED reference's loop (tests/test_oracle_vs_reference.py) -- and timed: the scans/s of the patched reference's own two workers.
"""
import json
import os

import numpy as np
import pytest

from open3d_slam_amd import synthetic as syn

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _patched_available():
    from oracle import ref

    return os.path.isfile(ref.LIB_PATCHED) or ref.sources_present()


def _scans(frames, stride=1):
    scene = syn.make_scene()
    poses = syn.figure_eight_poses(200, 0.1)
    return [np.asarray(syn.os128_scan(scene, poses[k], frame=k), dtype=np.float32)[::stride] for k in range(frames)], poses


def _device_resident_loop(scans, mp, op):
    from open3d_slam_amd import backend
    from open3d_slam_amd.mapper import Mapper
    from open3d_slam_amd.odometry import LidarOdometry
    from open3d_slam_amd.pointcloud import PointCloud

    be = backend.Backend(0)
    odo = LidarOdometry(be)
    odo.setParameters(op)
    mapper = Mapper(be, odo)
    mapper.setParameters(mp)
    out = []
    for k, raw in enumerate(scans):
        cloud = PointCloud.from_pointcloud2(be, raw)
        assert odo.addRangeScan(cloud, 0.1 * k) and mapper.addRangeMeasurement(cloud, 0.1 * k)
        cloud.release()
        out.append((mapper.getMapToRangeSensor().copy(), odo.getOdomToRangeSensor(0.1 * k).copy()))
    n_map = len(mapper.getActiveSubmap().getMapPointCloud())
    be.close()
    return out, n_map


@pytest.mark.skipif(not _patched_available(), reason="oracle/_ref/libo3dslam_ref_patched.so is neither built nor buildable here")
def test_the_patched_reference_runs_on_the_gpu_and_matches_the_device_loop_and_the_oracle():
    import bench
    from oracle import pyoracle as po
    from oracle import ref
    from oracle.pipeline import OracleLoop

    mp, op = bench.stream_parameters()
    frames = 24  # carving acts at the 2nd, 12th and 22nd insertion
    scans, truth = _scans(frames)
    R = ref.ReferenceSlam(mp, op, patched=True)
    ok, M, O, ms, n_map = R.run_stream(scans)
    R.close()
    assert ok == frames
    dev, dev_map = _device_resident_loop(scans, mp, op)
    worst = max(max(*syn.se3_error(M[k], dev[k][0]), *syn.se3_error(O[k], dev[k][1])) for k in range(frames))
    assert worst <= 1e-9, worst  # the same kernels on the same values, by way of host clouds of doubles at every seam
    assert n_map == dev_map
    loop = OracleLoop(po, mp, op)
    worst_cpu = 0.0
    for k, s in enumerate(scans):
        s64 = np.asarray(s, dtype=np.float64)
        loop.odometry(s64, k)
        loop.mapping(s64, k)
        worst_cpu = max(worst_cpu, *syn.se3_error(M[k], loop.T))
    assert worst_cpu <= 1e-3, worst_cpu  # f32 storage on the device (BASELINE tolerance); measured ~1e-5
    assert abs(n_map - len(loop.map_p)) <= 0.002 * n_map
    rel = np.linalg.inv(truth[0]) @ truth[frames - 1]
    assert np.linalg.norm(M[-1][:3, 3] - rel[:3, 3]) < 0.05
    # and the reference BEFORE the patch (the same sources unpatched, Open3D calls served by the CPU oracle), fed the same scans: what the
    # patch changes is where the work runs, not what comes out
    R0 = ref.ReferenceSlam(mp, op, patched=False)
    ok0, M0, O0, ms0, n_map0 = R0.run_stream(scans)
    R0.close()
    assert ok0 == frames
    worst_ref = max(max(*syn.se3_error(M[k], M0[k]), *syn.se3_error(O[k], O0[k])) for k in range(frames))
    assert worst_ref <= 1e-3, worst_ref  # f32 storage on the device against f64 on the CPU; measured ~1e-5
    assert abs(n_map - n_map0) <= 0.002 * n_map0
    print(f"patched vs unpatched reference over {frames} frames: worst pose difference {worst_ref:.2e}, map {n_map} vs {n_map0} points "
          f"(unpatched, on this box's CPU allowance: {ms0 / frames:.0f} ms per frame)")


@pytest.mark.skipif(not _patched_available(), reason="oracle/_ref/libo3dslam_ref_patched.so is neither built nor buildable here")
def test_scans_per_second_of_the_patched_reference(record_property):
    """200 full-size frames through the patched reference's two workers, serial and on two threads (timed inside the library; the raw
    scans are host PointClouds of doubles, as rosToOpen3d hands them over).  Writes gpurun_out/patched_reference_stream.json."""
    import bench
    from oracle import ref

    mp, op = bench.stream_parameters()
    scans, truth = _scans(200)
    out = {}
    # (the odometry worker may be `lead` scans ahead of the mapper: the reference's buffers between its workers hold one scan each, Parameters.hpp:82,175)
    for name, threads, lead in (("serial", False, None), ("two_threads", True, 2), ("two_threads_lead_1", True, 1), ("two_threads_lead_4", True, 4), ("two_threads_lead_6", True, 6)):
        warm = ref.ReferenceSlam(mp, op, patched=True)
        warm.run_stream(scans[:8], threads=threads, lead=lead)
        warm.close()
        R = ref.ReferenceSlam(mp, op, patched=True)
        ok, M, O, ms, n_map = R.run_stream(scans, threads=threads, lead=lead)
        R.close()
        rel = np.linalg.inv(truth[0]) @ truth[-1]
        err = float(np.linalg.norm(M[-1][:3, 3] - rel[:3, 3]))
        assert ok == 200 and err < 0.05, (ok, err)
        out[name] = {"scans_per_sec": 200e3 / ms, "ms_total": ms, "map_points": n_map, "final_translation_error_m": err,
                     "ms_per_scan_busy": {k: v / 199.0 for k, v in R.ms_workers.items()}}  # each worker's own clock inside its calls
        record_property(f"patched_reference_{name}_scans_per_sec", 200e3 / ms)
    # the same under the shipped Lua's registration type and down-sampling ratio (GeneralizedIcp, 0.3: parameter_structure_definitions.lua:62,76,109):
    # what the patch does with [O3D] RandomDownSample (o3ds::preprocessScan draws on the device) is on this path only
    mp_s, op_s = bench.stream_parameters(shipped=True)
    for name, threads in (("shipped_serial", False), ("shipped_two_threads", True)):
        warm = ref.ReferenceSlam(mp_s, op_s, patched=True)
        warm.run_stream(scans[:8], threads=threads, lead=2)
        warm.close()
        R = ref.ReferenceSlam(mp_s, op_s, patched=True)
        ok, M, O, ms, n_map = R.run_stream(scans, threads=threads, lead=2)
        R.close()
        rel = np.linalg.inv(truth[0]) @ truth[-1]
        err = float(np.linalg.norm(M[-1][:3, 3] - rel[:3, 3]))
        assert ok == 200 and err < 0.05, (ok, err)
        out[name] = {"scans_per_sec": 200e3 / ms, "ms_total": ms, "map_points": n_map, "final_translation_error_m": err,
                     "ms_per_scan_busy": {k: v / 199.0 for k, v in R.ms_workers.items()}}
    out["what"] = ("open3d_slam's own LidarOdometry::addRangeScan + Mapper::addRangeMeasurement (reference sources with "
                   "integration/open3d_slam_o3ds.patch applied, stand-in Eigen / PointCloud container) on libo3ds_backend.so, 200 frames x "
                   "131072 points, carving every 10th insertion as the reference does")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "patched_reference_stream.json"), "w") as f:
        json.dump(out, f, indent=1)
    assert out["serial"]["scans_per_sec"] > 100
