"""hipGraph replay (dana_amd.graphs) against the eager path: same kernels, same order, same np.random stream -> the
same numbers. The eager path is what every parity test pins against the oracle / the reference's goldens."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(dev, ba=True, way=2, shot=2, train=True, seed=5):
    import dana_amd
    from dana_amd import synthetic as S
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=ba, way=way, shot=shot, classes=["fg", "bg"])
    m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=seed, profile="test"))
    m.to(dev)
    m.train() if train else m.eval()
    return m


def _same(a, b, tol=0.0):
    for x, y in zip(a, b):
        if torch.is_tensor(x):
            if tol == 0.0:
                assert torch.equal(x, y)
            else:
                assert float((x.float() - y.float()).abs().max()) <= tol * max(1.0, float(y.float().abs().max()))
        else:
            assert x == y


def test_graphed_train_forward_equals_eager_with_the_reference_rng_stream(dev):
    from dana_amd import synthetic as S
    from dana_amd.graphs import GraphedDAnA
    m = _model(dev)
    inputs = [t.to(dev) for t in S.episode_inputs(2, 2, 2, 192, 256, seed=9)]
    other = [t.to(dev) for t in S.episode_inputs(2, 2, 2, 192, 256, seed=10)]
    run = GraphedDAnA(m, *inputs)
    assert run.g2 is not None  # host RNG: two graphs around the one round trip
    for ins, seed in ((inputs, 3), (other, 4), (inputs, 3)):
        np.random.seed(seed)
        with torch.no_grad():
            ref = [t.clone() if torch.is_tensor(t) else t for t in m(*ins)]
        np.random.seed(seed)
        out = run(*ins)
        torch.cuda.synchronize()
        _same(out, ref)
        assert out[0].shape == (2, 128, 5) and bool(torch.isfinite(out[3]))


def test_graphed_forward_does_not_depend_on_what_recycled_memory_holds(dev):
    """replays between allocations that leave garbage in the caching allocator's blocks: every scratch word a captured
    kernel reads must have been written by the graph itself (regression: a strided memset node that did not replay)"""
    from dana_amd import synthetic as S
    from dana_amd.graphs import GraphedDAnA
    m = _model(dev)
    inputs = [t.to(dev) for t in S.episode_inputs(2, 2, 2, 192, 256, seed=9)]

    def garbage(seed):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        ts = [torch.randint(-2 ** 31, 2 ** 31 - 1, (n,), dtype=torch.int32, device=dev, generator=g)
              for n in (1 << 24, 1 << 22, 1 << 20, 1 << 18, 3 << 16, 5 << 14, 1 << 12)]
        del ts

    np.random.seed(3)
    with torch.no_grad():
        ref = [t.clone() if torch.is_tensor(t) else t for t in m(*inputs)]
    run = GraphedDAnA(m, *inputs)
    for it in range(12):
        garbage(it)
        np.random.seed(3)
        with torch.no_grad():
            _same(m(*inputs), ref)  # the eager path too
        garbage(100 + it)
        np.random.seed(3)
        out = run(*inputs)
        torch.cuda.synchronize()
        _same(out, ref)


def test_graphed_eval_forward_equals_eager(dev):
    from dana_amd import synthetic as S
    from dana_amd.graphs import GraphedDAnA
    m = _model(dev, train=False)
    inputs = [t.to(dev) for t in S.episode_inputs(1, 1, 2, 160, 224, seed=2)]
    run = GraphedDAnA(m, *inputs)
    assert run.g2 is None
    with torch.no_grad():
        ref = [t.clone() if torch.is_tensor(t) else t for t in m(*inputs)]
    out = run(*inputs)
    torch.cuda.synchronize()
    _same(out, ref)
    assert out[3:] == (0, 0, 0, 0, None)


def test_graphed_forward_with_device_rng_is_one_graph_and_advances_its_counter(dev):
    from dana_amd import synthetic as S
    from dana_amd.graphs import GraphedDAnA
    m = _model(dev, ba=False)
    m.device_rng = True
    inputs = [t.to(dev) for t in S.episode_inputs(2, 2, 2, 192, 256, seed=9)]
    m._rng_calls = 0
    with torch.no_grad():
        eager = [[t.clone() for t in m(*inputs)] for _ in range(3)]
    run = GraphedDAnA(m, *inputs, warmup=0)
    assert run.g2 is None  # no host round trip
    m._rng_counter(dev).zero_()
    for k in range(3):  # replay k draws what the eager call number k drew
        out = run(*inputs)
        torch.cuda.synchronize()
        _same(out, eager[k])
    assert not torch.equal(eager[0][0], eager[1][0])


def _params(m):
    return torch.cat([p.detach().reshape(-1)[::53] for p in m.parameters() if p.requires_grad]).cpu().numpy()


@pytest.mark.parametrize("rccl", [False, True], ids=["single", "rccl1rank"])
def test_graphed_training_iteration_equals_trainer_step(dev, rccl):
    """three iterations replayed from hipGraphs == three eager Trainer.step calls (SGD with momentum, weights updated
    in place under the graphs). rccl1rank: a 1-rank RCCL group with always_reduce -> the multi-rank graph layout (cut
    before the trunk's backward, bucket all-reduces between the replays, SGD in its own graph)."""
    import torch.distributed as dist
    from dana_amd import synthetic as S
    from dana_amd.graphs import GraphedTrainer
    from dana_amd.trainer import Trainer
    if rccl:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29671"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        inputs = [t.to(dev) for t in S.episode_inputs(1, 2, 2, 160, 224, seed=6)]
        m0 = _model(dev)
        t0 = Trainer(m0, 0.01, bucket_bytes=8 << 20)
        for it in range(5):
            np.random.seed(40 + it)
            ref_out = t0.step(*inputs)
        torch.cuda.synchronize()
        ref = _params(m0)
        ref_losses = [float(x) for x in ref_out[3:7]]
        m1 = _model(dev)
        t1 = Trainer(m1, 0.01, bucket_bytes=8 << 20, always_reduce=rccl)
        for it in range(2):  # the capture's own eager warm-up iterations ARE the first two iterations
            np.random.seed(40 + it)
            if it == 0:
                gt = None
            t1.step(*inputs)
        gt = GraphedTrainer(t1, *inputs, warmup=0)
        assert len(gt.graphs) == (4 if rccl else 2)
        if rccl:
            assert sum(len(b) for _, b in gt.graphs) >= 3  # the buckets leave between the replays
        for it in range(2, 5):
            np.random.seed(40 + it)
            out = gt.step(*inputs)
        torch.cuda.synchronize()
        got = _params(m1)
        d = np.abs(got - ref).max()
        assert d <= 1e-6 + 1e-4 * np.abs(ref).max(), d  # (RoIAlign-backward atomics are unordered)
        for a, b in zip([float(x) for x in out[3:7]], ref_losses):
            assert abs(a - b) <= 1e-4 * max(1.0, abs(b))
        assert t1.steps == 5
        # an EAGER forward after graphed training must see the updated weights (the graphs own their derived copies, the
        # eager path's packed / Winograd / two-segment caches have to be re-derived): eval forward of both twins
        m0.eval(), m1.eval()
        ev = [t.to(dev) for t in S.episode_inputs(1, 1, 2, 160, 224, seed=6)]
        with torch.no_grad():
            e0, e1 = m0(*ev), m1(*ev)
        torch.cuda.synchronize()
        # (the twins' weights differ by the unordered RoIAlign-backward atomics, ~1e-6: a near-tie in the proposal layer may
        # flip, so compare row-wise and ask for the bulk of the rois)
        close = ((e0[0].view(-1, 5) - e1[0].view(-1, 5)).abs().max(1).values < 0.05) & \
                ((e0[2] - e1[2]).abs().max(1).values <= 1e-3) & ((e0[1] - e1[1]).abs().max(1).values <= 1e-3)
        assert float(close.float().mean()) >= 0.9, float(close.float().mean())
        from dana_amd import ops
        live = ops.pack_conv_weight(m1.RCNN_rpn.RPN_Conv.weight)  # (the eager plan's copy is the LIVE weight's, bit for bit)
        assert torch.equal(m1._get_plan()["rpn_conv_w"], live)
        blk = m1.RCNN_base[6][1]
        assert torch.equal(m1._get_plan()["layers"][2][1]["c1"]["w"], ops.pack_conv_weight(blk.conv1.weight))
    finally:
        if rccl:
            dist.destroy_process_group()


def test_replays_keep_the_host_rng_count_and_the_warmup_keeps_np_random(dev):
    """advisor round 3: (a) the device-RNG replays advance the device Philox counter; the host-side call count follows, so
    an eager device-RNG forward AFTER the replays continues the stream instead of repeating the first draws; (b)
    GraphedTrainer's eager warm-up iterations consume np.random draws in host-RNG mode: the state is restored with the
    parameters, so the training trajectory does not depend on the warm-up."""
    from dana_amd import synthetic as S
    from dana_amd.graphs import GraphedDAnA, GraphedTrainer
    from dana_amd.trainer import Trainer
    m = _model(dev, ba=False)
    m.device_rng = True
    inputs = [t.to(dev) for t in S.episode_inputs(2, 2, 2, 192, 256, seed=9)]
    m._rng_calls = 0
    with torch.no_grad():
        eager = [[t.clone() for t in m(*inputs)] for _ in range(4)]  # draws 0..3 of the counter sequence
    m._rng_calls = 0
    run = GraphedDAnA(m, *inputs, warmup=0)
    assert m._rng_calls == 0  # (the capture's own forward draws nothing and leaves the host count where it was)
    assert int(m._rng_counter(dev).item()) == 0
    for k in range(3):
        out = run(*inputs)
        torch.cuda.synchronize()
        _same(out, eager[k])
    assert m._rng_calls == 3
    with torch.no_grad():
        nxt = m(*inputs)  # eager, device RNG: continues with draw 3, not draw 0
    torch.cuda.synchronize()
    _same(nxt, eager[3])
    # (b)
    m2 = _model(dev)
    tr = Trainer(m2, 0.01)
    tin = [t.to(dev) for t in S.episode_inputs(1, 2, 2, 160, 224, seed=6)]
    np.random.seed(77)
    before = np.random.get_state()[1].copy()
    p0 = _params(m2)
    GraphedTrainer(tr, *tin, warmup=2)
    assert np.array_equal(np.random.get_state()[1], before) and np.array_equal(_params(m2), p0) and tr.steps == 0


def test_trainer_leaves_the_models_configured_forward_path_alone(dev):
    """advisor round 3: the Trainer's shared [query | support] buffers are a preference for the forwards that save for ITS
    backward only -- eval / inference / bench forwards of the same model keep merge_trunk / merge_from as configured, and
    give the same bits as before the Trainer existed"""
    from dana_amd import synthetic as S
    from dana_amd.trainer import Trainer
    m = _model(dev)
    inputs = [t.to(dev) for t in S.episode_inputs(1, 2, 2, 160, 224, seed=6)]
    np.random.seed(5)
    with torch.no_grad():
        ref = [t.clone() if torch.is_tensor(t) else t for t in m(*inputs)]
    cfgd = (m.merge_trunk, m.merge_from)
    tr = Trainer(m, 0.0)  # lr 0: the step leaves the weights alone
    assert (m.merge_trunk, m.merge_from) == cfgd and m._train_merge == (True, 3)
    np.random.seed(5)
    out_t = tr.step(*inputs)  # saving forward: shared buffers (bit-identical outputs in every merge mode)
    np.random.seed(5)
    with torch.no_grad():
        out = m(*inputs)
    torch.cuda.synchronize()
    _same(out, ref)
    # (the saving forward keeps the two-output RoIAlign / two query GEMMs its backward reads, the forward-only run folds the
    # positional encoding into one projection: same rois, scores equal to fp32 roundoff)
    assert torch.equal(out_t[0], ref[0])
    assert float((out_t[1] - ref[1]).abs().max()) <= 2e-6 and float((out_t[2] - ref[2]).abs().max()) <= 2e-5
