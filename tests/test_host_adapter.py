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
