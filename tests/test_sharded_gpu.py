"""The N>1 path with the REAL kernels: two processes sharing the one GPU of the test box, gloo process group carrying the
all-reduce of the CUDA record tensor (RCCL needs one GPU per rank, so NCCL itself cannot be exercised on a 1-GPU box; the
driver's 8-GPU run does that).  Checks open3d_slam_amd.sharded.ShardedIcp in both partitionings against single-process runs."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from open3d_slam_amd import backend, sharded, synthetic as syn

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = syn.make_scene()
    src = syn.vlp16_scan(scene, syn.ground_truth_pose(), n_az=512)
    tgt, nrm = syn.sample_map(scene, 100_000, seed=syn.SEED_MAP + (rank if mode == "submap" else 0))
    be = backend.Backend(0, backend.PRECISION_F64)
    s, t = be.upload(src), be.upload(tgt, nrm)
    be.build_index(t, 1.0)
    drv = sharded.ShardedIcp(be, mode=mode)
    res = drv.register(s, t, len(src), 1.0, max_iter=30, check_every=2)
    Ts = [torch.zeros(16, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(Ts, torch.from_numpy(res["transformation"].ravel().copy()))
    if rank == 0:
        np.savez(out_path, T=res["transformation"], fitness=res["fitness"], rmse=res["inlier_rmse"], iterations=res["iterations"],
                 converged=res["converged"], all_T=np.stack([x.numpy() for x in Ts]))
    be.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["source", "submap"])
def test_two_ranks_one_gpu(tmp_path, backend_f64, oracle, mode):
    import torch.multiprocessing as mp

    from open3d_slam_amd import synthetic as syn

    out = str(tmp_path / f"{mode}.npz")
    mp.spawn(_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    r = np.load(out)
    np.testing.assert_array_equal(r["all_T"][0], r["all_T"][1])  # identical pose on every rank, no broadcast
    scene = syn.make_scene()
    src = syn.vlp16_scan(scene, syn.ground_truth_pose(), n_az=512)
    if mode == "source":
        tgt, nrm = syn.sample_map(scene, 100_000)
        one = backend_f64.icp_point_to_plane(src, tgt, nrm, 1.0, max_iter=30)
        np.testing.assert_allclose(r["T"], one["transformation"], atol=1e-10)
        assert int(r["iterations"]) == one["iterations"] and bool(r["converged"]) == one["converged"]
        assert abs(float(r["fitness"]) - one["fitness"]) < 1e-12
    else:
        # joint registration against both submaps == registration against their union when correspondences are taken per submap:
        # compare with the truth instead (the CPU gloo test checks the joint algebra against the oracle)
        dt, dr = syn.se3_error(r["T"], syn.ground_truth_pose())
        assert dt < 5e-3 and dr < 1e-3 and bool(r["converged"])
