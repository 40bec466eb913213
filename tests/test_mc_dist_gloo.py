"""CPU, world_size 2 over gloo: the MC-sample sharding + packed all-reduce logic of bayesian_torch_amd.mc_dist.
The packed buffers are produced by the oracle's epilogue (the HIP epilogue kernel is GPU-only); what is under
test is the host logic the N > 1 path adds: shard(), finish_pack(), reduce_packed(), unpack()."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bayesian_torch_amd import mc_dist


def test_shard_is_a_partition():
    for S in (1, 2, 7, 32, 128):
        for world in (1, 2, 3, 8):
            if S < world:
                continue
            cover = []
            for r in range(world):
                first, n = mc_dist.shard(S, r, world)
                cover += list(range(first, first + n))
            assert cover == list(range(S))
            sizes = [mc_dist.shard(S, r, world)[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, S, B, C, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import bt_oracle as O
        g = torch.Generator().manual_seed(123)
        logits = torch.randn(S, B, C, generator=g) * 3          # every rank can compute any sample (counter-based draws)
        kl = torch.tensor(55.5)
        first, n = mc_dist.shard(S, rank, world)
        p, e, l = O.mc_epilogue_ref(logits[first:first + n])
        packed = torch.cat([p.reshape(-1), e, l.reshape(-1)])
        buf = mc_dist.finish_pack(packed, kl, world)
        assert buf.numel() == mc_dist.pack_size(B, C)
        mc_dist.reduce_packed(buf)
        res = mc_dist.unpack(buf, B, C, S)
        pa, ea, la = O.mc_epilogue_ref(logits)
        ok = (torch.allclose(res["mean_prob"], pa / S, atol=1e-6) and torch.allclose(res["mean_entropy"], ea / S, atol=1e-6)
              and torch.allclose(res["mean_logits"], la / S, atol=1e-5) and abs(float(res["kl"]) - 55.5) < 1e-4)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_packed_allreduce():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, 5, 10, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, True), (1, True)]


def _worker_small(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        try:
            mc_dist.mc_predict(None, torch.zeros(2, 3), 1)      # S_total = 1 < world = 2: refused before any work, on EVERY rank
            q.put((rank, "no error"))
        except RuntimeError as e:
            q.put((rank, "raised" if "smaller than the world size" in str(e) else repr(e)))
        dist.barrier()                                          # nobody is stuck in a collective
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_fewer_samples_than_ranks_raises_on_every_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_small, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(30)
    assert got == {0: "raised", 1: "raised"}, got
