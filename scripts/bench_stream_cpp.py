#!/usr/bin/env python
"""The config-2 stream through the C++ host classes (tests/cpp/stream_mapping.cpp): writes the synthetic scans, compiles the
program with g++ against libo3ds_backend.so and runs it.  Host clouds cross the seam, so this is the PCIe-inclusive rate."""
import argparse, os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open3d_slam_amd import build, synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=20)
args = ap.parse_args()
build.build_backend()
scene = syn.make_scene()
poses = syn.figure_eight_poses(200, 0.1)
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "scans.bin")
with open(path, "wb") as f:
    first = syn.os128_scan(scene, poses[0], frame=0)
    f.write(struct.pack("ii", args.frames, len(first)))
    for k in range(args.frames):
        s = first if k == 0 else syn.os128_scan(scene, poses[k], frame=k)
        s = s.astype(np.float32).astype(np.float64)  # what a lidar driver delivers
        f.write(np.ascontiguousarray(poses[k].T, dtype=np.float64).tobytes())  # column-major
        f.write(np.ascontiguousarray(s, dtype=np.float64).tobytes())
exe = os.path.join(tmp, "stream_mapping")
lib = os.path.join(ROOT, "open3d_slam_amd", "lib")
subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "cpp", "stream_mapping.cpp"), "-L" + lib,
                       "-lo3ds_backend", "-Wl,-rpath," + lib])
sys.exit(subprocess.call([exe, path]))
