"""Seams 2/3 inside open3d_slam: integration/open3d_slam_o3ds.patch applies to the reference sources and the patched tree type-checks.

The patch keeps every class, header and signature of open3d_slam and replaces only the bodies of the scan-matching / map-fusion path
with calls into the C ABI (through integration/o3ds_open3d_slam.hpp).  Here it is applied to a temporary copy of the reference sources
and g++ -fsyntax-only is run on the four patched translation units AND on the two untouched callers that hold Submap by value
(Mapper.cpp, SubmapCollection.cpp: std::vector<Submap>, copies) against the reference's OWN headers; Eigen and Open3D, absent from this
image, are stood in for by declaration-only headers (tests/cpp/ref_shim).  Skipped where the reference checkout is not present (the
GPU box)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/open3d_slam/open3d_slam"
PATCH = os.path.join(ROOT, "integration", "open3d_slam_o3ds.patch")
UNITS = ["Submap.cpp", "ScanToMapRegistration.cpp", "CloudRegistration.cpp", "Odometry.cpp", "Mapper.cpp", "SubmapCollection.cpp"]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("patch") is None or shutil.which("g++") is None,
                                reason="needs the reference checkout, patch and g++")


def _syntax_check(tree, unit):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "tests", "cpp", "ref_shim"), "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "integration"), "-I" + os.path.join(tree, "include"), os.path.join(tree, "src", unit)]
    return subprocess.run(cmd, capture_output=True, text=True)


def test_patch_applies_and_the_patched_sources_type_check(tmp_path):
    tree = tmp_path / "open3d_slam"
    tree.mkdir()
    for sub in ("include", "src"):
        shutil.copytree(os.path.join(REF, sub), tree / sub)
    shutil.copy(os.path.join(REF, "CMakeLists.txt"), tree / "CMakeLists.txt")
    # the shims are good enough for the UNPATCHED sources (otherwise a pass below would mean nothing)
    for unit in UNITS:
        r = _syntax_check(str(tree), unit)
        assert r.returncode == 0, (unit, r.stderr[-2000:])
    r = subprocess.run(["patch", "-p1", "--batch", "-i", PATCH], cwd=tree, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    touched = {ln.split("b/", 1)[1].strip() for ln in open(PATCH) if ln.startswith("+++ b/")}
    assert touched == {"CMakeLists.txt", "include/open3d_slam/Submap.hpp", "include/open3d_slam/Odometry.hpp", "src/CloudRegistration.cpp",
                       "src/ScanToMapRegistration.cpp", "src/Odometry.cpp", "src/Submap.cpp",
                       "src/Mapper.cpp"}  # (Mapper.cpp: two lines -- the scan's Time stamp is passed on to the pre-processing memo)
    for unit in UNITS:
        r = _syntax_check(str(tree), unit)
        assert r.returncode == 0, (unit, r.stderr[-3000:])
    # the path no longer calls Open3D's registration / normal estimation / voxel merge on the host
    for unit, gone in (("CloudRegistration.cpp", ("RegistrationICP(", "RegistrationGeneralizedICP(", "EstimateNormals(")),
                       ("ScanToMapRegistration.cpp", ("scanMatcherCropper_->crop(activeSubmapPointCloud)", "mapBuilderCropper_->crop(in)",
                                                      "o3d_slam::voxelize(", "RandomDownSample(", "scanMatcherCropper_->crop(*wideCropped)")),
                       ("Odometry.cpp", ("cropper_->crop(in)", "o3d_slam::voxelize(", "RandomDownSample(", "registerClouds(cloudPrev_,")),
                       ("Submap.cpp", ("mapCloud_ += *transformedCloud", "voxelizeInsideCroppingVolume(*mapBuilderCropper_"))):
        txt = open(tree / "src" / unit).read()
        for g in gone:
            assert g not in txt, (unit, g)
    # and the public interface of Submap is what it was: only additions
    before = open(os.path.join(REF, "include", "open3d_slam", "Submap.hpp")).read().splitlines()
    after = open(tree / "include" / "open3d_slam" / "Submap.hpp").read().splitlines()
    removed = [ln for ln in before if ln.strip() and ln not in after]
    assert removed == ["  PointCloud sparseMapCloud_, mapCloud_;"], removed
    before = open(os.path.join(REF, "include", "open3d_slam", "Odometry.hpp")).read().splitlines()
    after = open(tree / "include" / "open3d_slam" / "Odometry.hpp").read().splitlines()
    assert [ln for ln in before if ln.strip() and ln not in after] == []  # LidarOdometry only gains a member
