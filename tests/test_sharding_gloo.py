"""CPU suite, part 3: the N>1 path over gloo, world_size 2 (the same code runs over RCCL on the GPUs).
Episodes are sharded with no data-path collective; the gradient all-reduce and the max-over-ranks
timing reduction are the only exchange steps."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dana_amd import parallel, synthetic as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        inputs = S.episode_inputs(5, 2, 1, 32, 48, seed=7, support_size=32)
        shard = parallel.shard_episode(inputs, rank, world)
        b0, b1 = parallel.shard_bounds(5, rank, world)
        assert all(t.size(0) == b1 - b0 for t in shard)
        assert torch.equal(shard[0], inputs[0][b0:b1]) and torch.equal(shard[4], inputs[4][b0:b1])
        # every episode is processed exactly once across ranks
        seen = torch.zeros(5)
        seen[b0:b1] = 1
        dist.all_reduce(seen)
        assert torch.equal(seen, torch.ones(5))
        # gradient mean all-reduce, bucketed (3 tensors, tiny bucket size forces several buckets)
        grads = [torch.full((7, 3), float(rank + 1)), torch.full((5,), 10.0 * (rank + 1)), torch.full((2, 2), -1.0 * rank)]
        parallel.allreduce_mean_(grads, bucket_bytes=64)
        assert torch.allclose(grads[0], torch.full((7, 3), 1.5)) and torch.allclose(grads[1], torch.full((5,), 15.0))
        assert torch.allclose(grads[2], torch.full((2, 2), -0.5))
        # the trainer's flat gradient buckets: all-reduce leaves bucket by bucket as gradients are marked final
        from dana_amd.trainer import FlatBuckets
        torch.manual_seed(3)
        ps = [("a.weight", torch.nn.Parameter(torch.randn(6, 5))), ("b.weight", torch.nn.Parameter(torch.randn(33))),
              ("c.weight", torch.nn.Parameter(torch.randn(4, 4))), ("d.weight", torch.nn.Parameter(torch.randn(7)))]
        before = [p.detach().clone() for _, p in ps]
        fb = FlatBuckets(ps, bucket_bytes=160)  # -> buckets {a}, {b}, {c, d}
        assert [ns for _, _, ns in fb.buckets] == [["a.weight"], ["b.weight"], ["c.weight", "d.weight"]]
        assert all(torch.equal(p.detach(), b0_) for (_, p), b0_ in zip(ps, before))  # values survive the re-pointing
        assert all(o % 4 == 0 for o, _ in fb.offsets.values())
        fb.zero_grad()
        for i, (_, p) in enumerate(ps):
            p.grad.add_(float((rank + 1) * (i + 1)))
        fb.mark_ready(["a.weight"])
        fb.mark_ready(["c.weight"])
        assert fb.launch_order == [0]
        fb.mark_ready(["d.weight", "frozen.weight", "b.weight"])
        assert fb.launch_order == [0, 2, 1]
        fb.wait_all()
        for i, (_, p) in enumerate(ps):  # SUM over the two ranks; the mean's 1/world is folded into the SGD launch
            assert torch.allclose(p.grad, torch.full_like(p, 3.0 * (i + 1)))
        fb.zero_grad()
        fb.mark_ready(["a.weight"])
        try:
            fb.wait_all()
            raise AssertionError("wait_all must refuse while buckets never left")
        except RuntimeError:
            pass
        # step time = max over ranks
        assert parallel.max_over_ranks(1.0 + rank, torch.device("cpu")) == 2.0
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_allreduce_over_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_shard_bounds_cover_everything():
    from dana_amd import parallel
    for total in (1, 4, 5, 16):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
