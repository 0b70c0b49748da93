"""-m gpu: the N > 1 DEVICE path on a one-GPU box — two processes share GPU 0, each owns its shard of the edges and runs
the real kernels; the per-edge blocks / counts are exchanged through the host-staged callback (gloo) instead of RCCL
(RCCL refuses two ranks on one device).  Poses must be bit-identical to the single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(rank, world, reduce_fn):
    sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
    import mvicp
    from mvicp import synth
    pb = synth.make_problem(6, 5000)
    eng = mvicp.Engine(0, rank=rank, world=world)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    if reduce_fn is not None:
        eng.comm_set_callback(reduce_fn)
    poses = pb["init"].copy()
    hist = []
    for _ in range(5):
        c, w = eng.correspond(poses, pb["fixed"], 0.05)
        poses, sm = eng.optimize(poses, pb["fixed"])
        hist.append((c.copy(), w.copy(), sm["iterations"]))
    eng.close()
    return poses, hist


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def allreduce(a):
        t = torch.from_numpy(a)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    poses, hist = _run(rank, world, allreduce)
    if rank == 0:
        np.save(out, poses)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_match_single_process_bitwise(tmp_path):
    out = str(tmp_path / "poses.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    P2 = np.load(out)
    P1, hist = _run(0, 1, None)
    assert hist[-1][2] >= 1
    assert np.array_equal(P1, P2), np.abs(P1 - P2).max()
