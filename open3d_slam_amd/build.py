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


LIB_AB = os.path.join(LIB_DIR, "libo3ds_backend_ab.so")  # the same source with -DO3DS_AB_SWITCHES: A/B levers and debugging aids read from
                                                          # the environment (tests / scripts only; the shipped library has none)


def needs_build() -> bool:
    srcs = glob.glob(os.path.join(CSRC, "*")) + [os.path.join(_PKG, "..", "include", "o3ds_backend.h")]
    for lib in (LIB, LIB_AB):
        if not os.path.exists(lib):
            return True
        t = os.path.getmtime(lib)
        if any(os.path.getmtime(s) > t for s in srcs):
            return True
    return False


def _cmd(out: str, extra=()) -> list:
    return [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-Rpass-analysis=kernel-resource-usage",
            # leading pointer / scalar kernel arguments arrive in SGPRs with the dispatch instead of through a first scalar load
            # (icp_fused_kernel issues its state loads from them; code for firmware without the feature is emitted alongside)
            "-mllvm", "-amdgpu-kernarg-preload-count=8", *extra, "-o", out, os.path.join(CSRC, "backend.hip")]


def build_backend(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    jobs = [(_cmd(LIB), LIB), (_cmd(LIB_AB, ["-DO3DS_AB_SWITCHES"]), LIB_AB)]
    procs = []
    for cmd, _ in jobs:  # the two compilations side by side
        if verbose:
            print(" ".join(cmd))
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate() for p in procs]
    for (cmd, lib), p, (_, err) in zip(jobs, procs, outs):
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed ({os.path.basename(lib)}):\n" + err[-4000:])
    _write_resources(outs[0][1])
    return LIB


RESOURCES = os.path.join(LIB_DIR, "kernel_resources.txt")  # one line per kernel: name vgpr sgpr scratch_bytes lds_bytes occupancy


def _write_resources(remarks: str) -> None:
    """The compiler's per-kernel resource remarks, condensed.  tests/test_abi.py holds the registration kernels to ZERO scratch: the
    compiler places VGPR spills inside divergent regions (store under a narrow EXEC mask, reload under a wider one), which once fed a
    garbage quantum to the exact record sums of the f64-storage kernel (round 4)."""
    import re

    rows, cur = [], None
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    with open(RESOURCES, "w") as f:
        for r in rows:
            f.write("%s vgpr %d sgpr %d scratch %d lds %d occupancy %d\n" % (r["name"], r.get("vgpr", -1), r.get("sgpr", -1), r.get("scratch", -1),
                                                                          r.get("lds", -1), r.get("occ", -1)))


def kernel_resources() -> dict:
    out = {}
    if os.path.exists(RESOURCES):
        for line in open(RESOURCES):
            t = line.split()
            out[t[0]] = {t[k]: int(t[k + 1]) for k in range(1, len(t) - 1, 2)}
    return out


if __name__ == "__main__":
    print(build_backend(force=True, verbose=True))
