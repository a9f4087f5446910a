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


def test_scratch_that_cleans_up_after_itself_stays_clean():
    """The voxel table of VoxelDownSample and the cell counters of the index build are never cleared between calls (the kernels that use
    them leave them as they found them).  A long, irregular sequence on ONE handle -- sizes from 1 to 60 k points, crops that keep
    everything / something / nothing, growing and shrinking capacities, index builds and normal estimations in between -- must give the
    oracle's output bit for bit every time: a slot or a counter left dirty by one call would corrupt the next."""
    import numpy as np

    from open3d_slam_amd import backend, synthetic as syn
    from oracle import pyoracle as po

    scene = syn.make_scene()
    full = syn.os128_scan(scene, np.eye(4), n_az=512)  # 65 536 points
    rng = np.random.default_rng(5)
    be = backend.Backend(0, backend.PRECISION_F64)
    try:
        for it in range(40):
            n = int(rng.choice([1, 2, 17, 300, 5_000, 20_000, 60_000]))
            pts = full[rng.choice(len(full), n, replace=False)]
            voxel = float(rng.choice([0.05, 0.1, 0.4, 2.0]))
            c = be.upload(pts)
            kind = it % 4
            if kind == 3:  # a volume that holds no point: the table is not touched
                crop_d = backend.make_crop(backend.CROP_MAX_RADIUS, center=(500.0, 0.0, 0.0), rmax=1.0)
                out = be.crop_voxel_down_sample(c, crop_d, voxel)
                assert be.size(out)[0] == 0
            elif kind == 2:
                ctr = tuple(float(x) for x in rng.uniform(-5, 5, 3))
                crop_d = backend.make_crop(backend.CROP_MIN_MAX_RADIUS, center=ctr, rmin=1.0, rmax=12.0)
                keep = po.crop_indices(pts, po.make_crop(po.CROP_MIN_MAX_RADIUS, center=ctr, rmin=1.0, rmax=12.0))
                out = be.crop_voxel_down_sample(c, crop_d, voxel)
                if len(keep):
                    np.testing.assert_array_equal(be.download(out)[0], po.voxel_down_sample(pts[keep], voxel), err_msg=f"call {it}")
                else:
                    assert be.size(out)[0] == 0
            else:
                out = be.voxel_down_sample(c, voxel)
                got = be.download(out)[0]
                np.testing.assert_array_equal(got, po.voxel_down_sample(pts, voxel), err_msg=f"call {it}")
                if kind == 1 and len(got) >= 3:  # an index build + the normals kernel on the result (bit-identical to the oracle's)
                    be.estimate_normals(out, 1.5, 10)
                    np.testing.assert_array_equal(be.download(out)[1], po.estimate_normals(got, 1.5, 10), err_msg=f"normals, call {it}")
            be.free(c)
            be.free(out)
    finally:
        be.close()
