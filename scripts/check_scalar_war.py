"""Finds, in the gfx950 assembly of the backend, a scalar load (s_load_*) of a global word that is still in flight (no s_waitcnt lgkmcnt(0)
since) when the same wavefront issues a vector STORE to the same address.  The hardware does not order the scalar and the vector memory
pipelines against each other, and the compiler only waits for the scalar load where its RESULT is first used: a kernel that reads a counter
and resets it a few lines later can read its own zero (pm_carve_finish_kernel did, once in ten cold starts).  Uniform loads of words the same
thread overwrites must be vector loads (load_then_store in common.hpp: an agent-scope atomic load).

usage: python scripts/check_scalar_war.py [backend.s]     (without an argument: compiles open3d_slam_amd/csrc/backend.hip to assembly first)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assembly(path=None):
    if path:
        return open(path).read()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "backend.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", "-o", out,
                        os.path.join(ROOT, "open3d_slam_amd", "csrc", "backend.hip")], check=True, capture_output=True)
        return open(out).read()


S_LOAD = re.compile(r"^\s*s_load_dword(x(\d+))?\s+s\[?(\d+)(?::(\d+))?\]?,\s*s\[(\d+):(\d+)\],\s*(0x[0-9a-f]+|\d+)")
V_STORE = re.compile(r"^\s*global_store_(dword(x(\d+))?|byte|short)\s+v\d+,\s*v\[?\d+(?::\d+)?\]?,\s*s\[(\d+):(\d+)\](?:\s+offset:(\d+))?")
V_ATOMIC = re.compile(r"^\s*global_atomic_\w+\s+(?:v\d+,\s*)?v\d+,\s*v\[?\d+(?::\d+)?\]?,\s*s\[(\d+):(\d+)\](?:\s+offset:(\d+))?")
SGPR_WRITE = re.compile(r"^\s*s_\w+\s+s\[?(\d+)(?::(\d+))?\]?,")


def check(text):
    findings = []
    kernel, pending, kernarg = None, [], None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, pending = m.group(1), []
            continue
        if kernel is None or line.lstrip().startswith(";"):
            continue
        if "s_endpgm" in line:
            kernel = None
            continue
        if re.search(r"s_waitcnt.*lgkmcnt\(0\)", line) or "s_waitcnt lgkmcnt(0)" in line:
            pending = []
            continue
        m = S_LOAD.match(line)
        if m:
            width = int(m.group(2) or 1) * 4
            base = (int(m.group(5)), int(m.group(6)))
            if base != (0, 1):  # s[0:1] is the kernel-argument segment: never written by the kernel
                pending.append((base, int(m.group(7), 0), width, line.strip()))
            continue
        m = V_STORE.match(line) or V_ATOMIC.match(line)
        if m and pending:
            if m.re is V_STORE:
                kind = m.group(1)
                width = {"byte": 1, "short": 2}.get(kind, int(m.group(3) or 1) * 4)
                base, off = (int(m.group(4)), int(m.group(5))), int(m.group(6) or 0)
            else:
                width, base, off = 8, (int(m.group(1)), int(m.group(2))), int(m.group(3) or 0)
            for pb, po, pw, pl in pending:
                if pb == base and po < off + width and off < po + pw:
                    findings.append((kernel, pl, line.strip()))
        # a scalar instruction that rewrites a base register pair ends what we know about the loads through it
        m = SGPR_WRITE.match(line)
        if m and not line.lstrip().startswith("s_load"):
            lo, hi = int(m.group(1)), int(m.group(2) or m.group(1))
            pending = [p for p in pending if not (lo <= p[0][1] and p[0][0] <= hi)]
    return findings


if __name__ == "__main__":
    f = check(assembly(sys.argv[1] if len(sys.argv) > 1 else None))
    for k, a, b in f:
        print(k, "\n   ", a, "\n   ", b)
    print(len(f), "scalar load(s) in flight across a vector store to the same address")
    sys.exit(1 if f else 0)
