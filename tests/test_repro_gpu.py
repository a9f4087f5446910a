"""The odometry + mapper stream repeats BIT FOR BIT, call by call: every cloud an ABI call returns or changes and every registration
result hash the same on fresh handles, with the device allocations disturbed in between (which changes the arrival order of the index
build's atomic scatter -- what made round 1's normals, and with them the whole stream, irreproducible)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


def test_stream_is_identical_call_by_call_on_cold_and_disturbed_pools():
    import repro_stream

    for dirty in (False, True):
        for run, first, frame, same_pose in repro_stream.check(frames=10, runs=3, dirty_pool=dirty, verbose=True):
            assert first is None, (dirty, run, first, frame)
            assert same_pose
