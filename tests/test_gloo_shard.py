"""N > 1 path on CPU (gloo, world_size 2): the edge partition (mvicp_edge_owner), the per-edge-slot all-reduce
(every slot written by exactly one rank, zeros elsewhere -> the sum is exact) and the replicated host LM must give
poses that are BIT-IDENTICAL to the single-process run.  The oracle stands in for the GPU evaluator (tests only);
on the GPU box the same all-reduce is RCCL over xGMI inside libmvicp_hip (csrc/comm.cpp)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    from mvicp import synth
    orc = orclib.load()
    pb = synth.make_problem(5, 400)
    corr, w = [], []
    for s, d in zip(pb["src"], pb["dst"]):
        f, sec, dist_, wt, _, _ = orc.correspond_edge(pb["pts"][s], pb["init"][s], pb["pts"][d], pb["init"][d], 0.05)
        corr.append((f, sec)); w.append(float(wt))
    return orc, pb, corr, w


def _solve(orc, pb, corr, w, owner, rank, world, reduce_fn):
    from mvicp import lib as L
    E = len(pb["src"])

    def evaluator(poses):
        blocks = np.zeros((E, 91))
        mine = [e for e in range(E) if owner[e] == rank]
        if mine:
            sub = orc.edge_blocks(pb["pts"], pb["nor"], pb["src"][mine], pb["dst"][mine], [corr[e] for e in mine], [w[e] for e in mine], poses, 1, 1)
            blocks[mine] = sub
        return reduce_fn(blocks)

    return L.lm_solve_host(len(pb["pts"]), pb["src"], pb["dst"], pb["init"], pb["fixed"], 2, evaluator, 50)


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc, pb, corr, w = _problem()
    from mvicp import lib as L
    owner = L.edge_owner([len(pb["pts"][s]) for s in pb["src"]], world)

    def allreduce(blocks):
        t = torch.from_numpy(blocks)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    P, sm = _solve(orc, pb, corr, w, owner, rank, world, allreduce)
    # every rank must hold the same poses
    t = torch.from_numpy(P.copy())
    ref = t.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(t, ref)
    if rank == 0:
        np.save(out_path, P)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process_bitwise(tmp_path):
    out = str(tmp_path / "poses.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    P2 = np.load(out)
    orc, pb, corr, w = _problem()
    owner = np.zeros(len(pb["src"]), dtype=np.int32)
    P1, sm = _solve(orc, pb, corr, w, owner, 0, 1, lambda b: b)
    assert sm["iterations"] >= 1
    assert np.array_equal(P1, P2), np.abs(P1 - P2).max()
