"""Builds libo3ds_backend.so for gfx950 with hipcc (in-tree, so it travels with the repo snapshot)."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIB_DIR = os.path.join(_PKG, "lib")
LIB = os.path.join(LIB_DIR, "libo3ds_backend.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    srcs = glob.glob(os.path.join(CSRC, "*")) + [os.path.join(_PKG, "..", "include", "o3ds_backend.h")]
    return any(os.path.getmtime(s) > t for s in srcs)


def build_backend(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB,
           os.path.join(CSRC, "backend.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_backend(force=True, verbose=True))
