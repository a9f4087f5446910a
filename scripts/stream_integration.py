#!/usr/bin/env python
"""BASELINE configs[2] through integration/o3ds_open3d_slam.hpp (tests/cpp/stream_integration.cpp): the functions the open3d_slam
patch calls, host clouds at every seam.  Library use (bench.py, tests): write_scans / compile_program / run.  Command line:
    python scripts/stream_integration.py --frames 200 --mode serial|threads
The program is compiled against stand-ins with the spelling and memory layout of the Open3D / Eigen types (tests/cpp/open3d_shim);
inside open3d_slam the same header is compiled against the real ones (tests/test_integration_patch.py type-checks that)."""
import argparse
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "open3d_slam_amd", "lib")


def write_scans(path, scans32, poses):
    """scans32: list of (n, 3) float32 arrays (sensor frame), poses: the true map <- sensor poses"""
    with open(path, "wb") as f:
        f.write(struct.pack("ii", len(scans32), len(scans32[0])))
        for s, T in zip(scans32, poses):
            assert s.dtype == np.float32 and s.shape == scans32[0].shape
            f.write(np.ascontiguousarray(np.asarray(T).T, dtype=np.float64).tobytes())  # column-major
            f.write(np.ascontiguousarray(s).tobytes())


def compile_program(out_dir, werror=True):
    exe = os.path.join(out_dir, "stream_integration")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra"] + (["-Werror"] if werror else []) + [
        "-pthread", "-I" + os.path.join(ROOT, "tests", "cpp", "open3d_shim"), "-I" + os.path.join(ROOT, "include"), "-o", exe,
        os.path.join(ROOT, "tests", "cpp", "stream_integration.cpp"), "-L" + LIB, "-lo3ds_backend", "-Wl,-rpath," + LIB]
    subprocess.check_call(cmd)
    return exe


def run(exe, scans_path, mode="serial", poses_path=None, env=None, timeout=600):
    cmd = [exe, scans_path, mode] + ([poses_path] if poses_path else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    if out.returncode != 0:
        raise RuntimeError(f"stream_integration rc={out.returncode}: {out.stdout[-500:]} {out.stderr[-1500:]}")
    return json.loads(out.stdout.strip().splitlines()[-1])


def read_poses(path, frames):
    a = np.fromfile(path, dtype=np.float64).reshape(frames, 2, 4, 4)
    return a[:, 0].transpose(0, 2, 1).copy(), a[:, 1].transpose(0, 2, 1).copy()  # column-major on disk


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--mode", default="serial", choices=["serial", "threads"])
    args = ap.parse_args()
    from open3d_slam_amd import build, synthetic as syn

    build.build_backend()
    scene = syn.make_scene()
    poses = syn.figure_eight_poses(200, 0.1)[: args.frames]
    scans = [syn.os128_scan(scene, poses[k], frame=k).astype(np.float32) for k in range(args.frames)]
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "scans.bin")
    write_scans(path, scans, poses)
    print(json.dumps(run(compile_program(tmp), path, args.mode)))
