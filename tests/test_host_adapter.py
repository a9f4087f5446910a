"""C++ host side (open3d_slam_amd/host/o3ds_adapter.hpp: CloudRegistration family, croppers, helpers, DeviceSubmap;
o3ds_mapping.hpp: ScanToMapIcp, Submap, VoxelizedPointCloud and the remaining helpers): the reference-named classes over the C-ABI.
CPU: both programs compile with plain g++ (-Wall -Wextra -Werror) and their device-free checks pass.  GPU: the full self-checking programs run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "open3d_slam_amd", "lib")


def _compile(tmp_path_factory, name, extra=()):
    from open3d_slam_amd import build

    build.build_backend()
    exe = str(tmp_path_factory.mktemp(name) / name)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-pthread", *extra, "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-L" + LIBDIR, "-lo3ds_backend", "-Wl,-rpath," + LIBDIR])
    return exe


@pytest.fixture(scope="module")
def adapter_exe(tmp_path_factory):
    return _compile(tmp_path_factory, "test_adapter")


@pytest.fixture(scope="module")
def mapping_exe(tmp_path_factory):
    return _compile(tmp_path_factory, "test_mapping")


def test_adapter_compiles_and_device_free_checks(adapter_exe):
    out = subprocess.run([adapter_exe, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "no-gpu checks ok" in out.stdout


@pytest.mark.gpu
def test_adapter_on_gpu(adapter_exe, tmp_path):
    out = subprocess.run([adapter_exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, O3DS_TEST_TMPDIR=str(tmp_path)))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gpu checks ok" in out.stdout


def test_mapping_compiles_and_device_free_checks(mapping_exe):
    out = subprocess.run([mapping_exe, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "no-gpu checks ok" in out.stdout


@pytest.mark.gpu
def test_mapping_on_gpu(mapping_exe, tmp_path):
    out = subprocess.run([mapping_exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, O3DS_TEST_TMPDIR=str(tmp_path)))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gpu checks ok" in out.stdout


def test_open3d_branch_of_the_host_headers_type_checks(tmp_path_factory):
    """The O3DS_USE_OPEN3D branch -- the one a maintainer compiles inside open3d_slam, with open3d::geometry::PointCloud and
    Eigen::Isometry3d at the seams -- cannot be built against the real libraries here (neither is in the image); it is type-checked
    against stand-ins with their spelling and memory layout (tests/cpp/open3d_shim), every member of every host class touched once."""
    exe = _compile(tmp_path_factory, "open3d_branch_compiles", extra=("-I" + os.path.join(ROOT, "tests", "cpp", "open3d_shim"),))
    assert subprocess.run([exe], timeout=60).returncode == 0  # main() does nothing; touch() is never called


def test_stream_mapping_program_compiles(tmp_path_factory):
    """tests/cpp/stream_mapping.cpp (the config-2 stream through the C++ classes) builds warning-free; it needs a device to run."""
    _compile(tmp_path_factory, "stream_mapping")


@pytest.mark.gpu
def test_stream_mapping_through_the_cpp_classes():
    """tests/cpp/stream_mapping.cpp: eight OS-128-like scans along the figure-eight through ScanToMapIcp + Submap with host clouds at
    the seam; the program checks the fitness gate of every frame and the final pose against the truth (5 cm) itself."""
    out = subprocess.run(["python", os.path.join(ROOT, "scripts", "bench_stream_cpp.py"), "--frames", "8"], capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "scans_per_sec_mapping_only" in out.stdout


# ---- integration/o3ds_open3d_slam.hpp: the header the open3d_slam patch calls into (tests/test_integration_patch.py applies the patch)
@pytest.fixture(scope="module")
def integration_exe(tmp_path_factory):
    return _compile(tmp_path_factory, "test_integration", extra=("-I" + os.path.join(ROOT, "tests", "cpp", "open3d_shim"), "-I" + os.path.join(ROOT, "include")))


def test_integration_header_compiles_and_device_free_checks(integration_exe):
    out = subprocess.run([integration_exe, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "no-gpu checks ok" in out.stdout


@pytest.mark.gpu
def test_integration_header_on_gpu(integration_exe):
    out = subprocess.run([integration_exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gpu checks ok" in out.stdout


# ---- BASELINE configs[2] through the integration header, host clouds at every seam (tests/cpp/stream_integration.cpp) -------------------
def _si():
    import importlib.util

    spec = importlib.util.spec_from_file_location("stream_integration", os.path.join(ROOT, "scripts", "stream_integration.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_stream_integration_program_compiles(tmp_path):
    from open3d_slam_amd import build

    build.build_backend()
    _si().compile_program(str(tmp_path))  # -Wall -Wextra -Werror against the Open3D / Eigen stand-ins


@pytest.mark.gpu
def test_stream_through_the_integration_header_matches_the_device_resident_loop(tmp_path, backend_f32):
    """40 frames of configs[2] (131 072 points each) through integration/o3ds_open3d_slam.hpp -- exactly the calls
    the patched LidarOdometry::addRangeScan, ScanToMapIcp and Submap make, host clouds at every seam, every result downloaded -- against
    the device-resident loop of the Python mirror on the same scans: the same kernels on the same values in the same order, so map pose
    and odometry agree to 1e-9 at every frame (only the host-side 4x4 algebra differs: numpy's inverse vs the program's rigid inverse).
    The two-thread form (odometryWorker / mappingWorker) must give the serial form's poses bit for bit."""
    import numpy as np

    from open3d_slam_amd import parameters as P, synthetic as syn
    from open3d_slam_amd.mapper import Mapper
    from open3d_slam_amd.odometry import LidarOdometry
    from open3d_slam_amd.pointcloud import PointCloud

    si = _si()
    frames = 40
    scene = syn.make_scene()
    poses = syn.figure_eight_poses(200, 0.1)[:frames]
    scans = [syn.os128_scan(scene, poses[k], frame=k).astype(np.float32) for k in range(frames)]
    path = str(tmp_path / "scans.bin")
    si.write_scans(path, scans, poses)
    exe = si.compile_program(str(tmp_path))
    res = si.run(exe, path, "serial", str(tmp_path / "serial.bin"))
    map_c, odo_c = si.read_poses(str(tmp_path / "serial.bin"), frames)
    res_t = si.run(exe, path, "threads", str(tmp_path / "threads.bin"))
    map_t, odo_t = si.read_poses(str(tmp_path / "threads.bin"), frames)
    np.testing.assert_array_equal(map_t, map_c)
    np.testing.assert_array_equal(odo_t, odo_c)
    assert res["min_fitness"] > 0.9 and res["final_translation_error_m"] < 0.02
    # the device-resident loop through the Python mirror, same parameters (stream_integration.cpp: Setup)
    mp = P.lua_default_mapper_parameters()
    mp.scanMatcher_.icp_.maxNumIter_ = 50
    op = P.OdometryParameters()
    op.scanMatcher_.icp_ = P.IcpParameters(maxNumIter_=50, maxCorrespondenceDistance_=1.0, knn_=20, maxDistanceKnn_=3.0)
    op.scanProcessing_.voxelSize_ = 0.1
    op.scanProcessing_.cropper_ = P.ScanCroppingParameters(croppingMinRadius_=2.0, croppingMaxRadius_=30.0, cropperName_="MinMaxRadius")
    be = backend_f32
    odo = LidarOdometry(be)
    odo.setParameters(op)
    mapper = Mapper(be, odo)
    mapper.setParameters(mp)
    worst = 0.0
    for k in range(frames):
        cloud = PointCloud.from_numpy(be, scans[k].astype(np.float64))
        assert odo.addRangeScan(cloud, 0.1 * k) and mapper.addRangeMeasurement(cloud, 0.1 * k)
        cloud.release()
        dt, dr = syn.se3_error(mapper.getMapToRangeSensor(), map_c[k])
        do, dor = syn.se3_error(odo.odomToRangeSensorCumulative_, odo_c[k])
        worst = max(worst, dt, dr, do, dor)
        assert max(dt, dr, do, dor) <= 1e-9, (k, dt, dr, do, dor)
    assert len(mapper.getActiveSubmap().getMapPointCloud()) == res["map_points"]
    print(f"integration header vs device-resident loop over {frames} frames: worst pose difference {worst:.2e}; "
          f"host seam {res['scans_per_sec']:.0f} scans/s serial, {res_t['scans_per_sec']:.0f} on two threads")
