#!/usr/bin/env python
"""Where the wall time of the patched reference's serial stream goes: oracle/ref_build/ref_driver.cpp's stack sampler (REF_SAMPLE_OUT) over
N frames of BASELINE configs[2].  Writes gpurun_out/patched_reference_samples.txt.  Needs a GPU (the patched reference calls libo3ds_backend.so)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
out = os.path.join(ROOT, "gpurun_out", "patched_reference_samples.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
if os.path.exists(out):
    os.unlink(out)
import bench  # noqa: E402
from oracle import ref  # noqa: E402
import test_patched_reference_gpu as T  # noqa: E402

frames = min(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200, 200)  # (the trajectory has 200 poses)
repeats = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
mp, op = bench.stream_parameters(shipped=("shipped" in sys.argv))
scans, truth = T._scans(frames)
warm = ref.ReferenceSlam(mp, op, patched=True)
warm.run_stream(scans[:8])
warm.close()
os.environ["REF_SAMPLE_OUT"] = out
for _ in range(repeats):  # (each run appends its report)
    R = ref.ReferenceSlam(mp, op, patched=True)
    ok, M, O, ms, n_map = R.run_stream(scans)
    R.close()
    print(f"{frames} frames, {ok} accepted, {frames * 1e3 / ms:.0f} scans/s, busy ms/scan {dict((k, v / (frames - 1)) for k, v in R.ms_workers.items())}")
print(open(out).read())
