"""-m gpu, needs >= 2 GPUs (skipped on a one-GPU box): the REAL multi-GPU path — one process per GPU, edges sharded,
mvicp_comm_unique_id on rank 0 -> broadcast -> mvicp_comm_init (RCCL over xGMI), per-edge blocks summed by ncclAllReduce(fp64)
inside libmvicp_hip (csrc/comm.cpp), counts / medians by the small host all-reduce.  Five ICP rounds; poses must be bit-identical
to the single-process run (every per-edge slot is written by exactly one rank, so the sum is exact).  The same exchange with
the speculative first evaluation switched off must give the same poses too (the queued launch carries its own all-reduce)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _icp(eng, pb, rounds=5):
    poses = pb["init"].copy()
    hist = []
    for _ in range(rounds):
        c, w = eng.correspond(poses, pb["fixed"], 0.05)
        poses, sm = eng.optimize(poses, pb["fixed"])
        hist.append((c.copy(), w.copy(), sm["iterations"]))
    return poses, hist


def _worker(rank, world, port, out, spec):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    import mvicp
    from mvicp import synth
    pb = synth.make_problem(6, 20000)
    eng = mvicp.Engine(rank, rank=rank, world=world)
    eng.set_option("spec_eval", spec)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    rccl = rccl if os.path.exists(rccl) else None
    uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(mvicp.Engine.comm_unique_id(rccl)), dtype=torch.uint8))
    dist.broadcast(uid, 0)
    eng.comm_init(bytes(uid.cpu().numpy().tobytes()), rccl)
    poses, hist = _icp(eng, pb)
    eng.close()
    np.save(f"{out}.{rank}.npy", poses)
    np.save(f"{out}.{rank}.counts.npy", np.array([h[0] for h in hist]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.timeout(900)
@pytest.mark.parametrize("spec", [1, 0])
def test_two_gpus_rccl_match_single_process_bitwise(tmp_path, spec):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
    import mvicp
    from mvicp import synth
    out = str(tmp_path / "poses")
    mp.spawn(_worker, args=(2, _free_port(), out, spec), nprocs=2, join=True)
    pb = synth.make_problem(6, 20000)
    eng = mvicp.Engine(0)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    P1, hist = _icp(eng, pb)
    eng.close()
    for r in range(2):
        P2 = np.load(f"{out}.{r}.npy")
        assert np.array_equal(P1, P2), (r, np.abs(P1 - P2).max())
        assert np.array_equal(np.load(f"{out}.{r}.counts.npy"), np.array([h[0] for h in hist]))   # every rank sees the global counts


def _regime_worker(rank, world, port, out, seed, spec):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    import mvicp
    import regime_seq
    from mvicp import synth
    pb = synth.make_problem(5, 3500, cone_deg=100.0 if seed % 2 == 0 else 45.0, pose_seed=800 + seed)
    eng = mvicp.Engine(rank, rank=rank, world=world)
    eng.set_option("spec_eval", spec)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    rccl = rccl if os.path.exists(rccl) else None
    uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(mvicp.Engine.comm_unique_id(rccl)), dtype=torch.uint8))
    dist.broadcast(uid, 0)
    eng.comm_init(bytes(uid.cpu().numpy().tobytes()), rccl)
    log = regime_seq.run(eng, pb, regime_seq.script(seed, 5))
    eng.close()
    regime_seq.save(f"{out}.{rank}.npz", log)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.timeout(900)
@pytest.mark.parametrize("seed,spec", [(0, 1), (2, 0)])
def test_two_gpus_rccl_regime_transitions_match_single_process(tmp_path, seed, spec):
    """The scripted regime transitions of tests/regime_seq.py (cutoff / fixed mask / method / options / resets / repeated poses) through the REAL
    RCCL exchange on two GPUs: the first 2-GPU box exercises the same sequence tests/test_gpu_multirank.py runs through the host-staged transport."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mvicp
    import regime_seq
    from mvicp import synth
    out = str(tmp_path / "regime")
    mp.spawn(_regime_worker, args=(2, _free_port(), out, seed, spec), nprocs=2, join=True)
    pb = synth.make_problem(5, 3500, cone_deg=100.0 if seed % 2 == 0 else 45.0, pose_seed=800 + seed)
    eng = mvicp.Engine(0)
    eng.set_option("spec_eval", spec)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    log = regime_seq.run(eng, pb, regime_seq.script(seed, 5))
    eng.close()
    for r in range(2):
        regime_seq.assert_equal(f"{out}.{r}.npz", log, (seed, spec, r))
