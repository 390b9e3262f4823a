"""Launch-program replay (dana_amd.program) against the eager path it was recorded from: the same launches with the same
arguments on the same streams behind the same event edges -> the same numbers, at a fraction of the host time. The eager
path is what every parity test pins against the oracle / the reference's goldens (train.py:125-143, dana.py:87-220)."""
import os
import time

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


def _params(m):
    return np.concatenate([p.detach().float().cpu().numpy().ravel() for _, p in sorted(m.named_parameters())])


def test_program_eval_forward_equals_eager_bit_for_bit(dev):
    from dana_amd import synthetic as S
    from dana_amd.program import ProgramDAnA
    m = _model(dev, train=False, way=1)
    a = [t.to(dev) for t in S.episode_inputs(2, 1, 2, 160, 224, seed=6)]
    b = [t.to(dev) for t in S.episode_inputs(2, 1, 2, 160, 224, seed=7)]
    with torch.no_grad():
        ea = [t.clone() if torch.is_tensor(t) else t for t in m(*a)]
        eb = [t.clone() if torch.is_tensor(t) else t for t in m(*b)]
    run = ProgramDAnA(m, *a)
    assert run.p2 is None and run.p1.stats["launches"] > 80 and run.p1.stats["host_callbacks"] == 0
    for inputs, ref in ((b, eb), (a, ea), (b, eb)):  # other data through the same program, back and forth
        out = run(*inputs)
        torch.cuda.synchronize()
        for x, y in zip(out, ref):
            assert torch.equal(x, y) if torch.is_tensor(x) else x == y


def test_program_train_forward_equals_eager_with_the_reference_rng_stream(dev):
    """train-mode forward: two programs around the one host round trip; the np.random draws happen live, between them"""
    from dana_amd import synthetic as S
    from dana_amd.program import ProgramDAnA
    m = _model(dev)
    inputs = [t.to(dev) for t in S.episode_inputs(2, 2, 2, 160, 224, seed=6)]
    refs = []
    for seed in (3, 4):
        np.random.seed(seed)
        with torch.no_grad():
            refs.append([t.clone() if torch.is_tensor(t) else t for t in m(*inputs)])
    run = ProgramDAnA(m, *inputs)
    assert run.p2 is not None
    for seed, ref in ((4, refs[1]), (3, refs[0])):
        np.random.seed(seed)
        out = run(*inputs)
        torch.cuda.synchronize()
        for x, y in zip(out, ref):
            assert torch.equal(x, y) if torch.is_tensor(x) else x == y


def test_program_forward_with_device_rng_is_one_program_and_advances_its_counter(dev):
    """opt-in device RNG (row N2): no host round trip -> the whole train-mode forward is ONE program; the Philox call counter
    is data in device memory, so replay k draws what the eager call number k drew"""
    from dana_amd import synthetic as S
    from dana_amd.program import ProgramDAnA
    m = _model(dev, ba=False)
    m.device_rng = True
    inputs = [t.to(dev) for t in S.episode_inputs(2, 2, 2, 192, 256, seed=9)]
    m._rng_calls = 0
    with torch.no_grad():
        eager = [[t.clone() for t in m(*inputs)] for _ in range(4)]
    m._rng_calls = 0
    run = ProgramDAnA(m, *inputs, warmup=0)  # the recording is forward number 0
    assert run.p2 is None and m._rng_calls == 1
    for k in range(1, 4):
        out = run(*inputs)
        torch.cuda.synchronize()
        for x, y in zip(out, eager[k]):
            assert torch.equal(x, y)
    assert not torch.equal(eager[1][0], eager[2][0]) and m._rng_calls == 4
    with torch.no_grad():  # an eager forward afterwards continues the same stream of draws
        nxt = m(*inputs)
    assert not any(torch.equal(nxt[0], e[0]) for e in eager)


def test_program_refuses_what_it_cannot_replay(dev):
    from dana_amd import program
    p = program.LaunchProgram(dev)
    x = torch.ones(8, device=dev)
    with pytest.raises(RuntimeError, match="no replay rule"):
        with p.recording():
            torch.nonzero(x)  # data-dependent output shape: refused at record time, never silently dropped
    p = program.LaunchProgram(dev)
    with pytest.raises(RuntimeError, match="no replay rule"):
        with p.recording():
            float(x.sum())  # a host read of device data inside a program: the cut between two programs is the place for it
    p = program.LaunchProgram(dev)
    with pytest.raises(RuntimeError, match="no replay rule"):
        with p.recording():
            x.cpu()  # ... also as a plain device-to-host copy
    # ... and the in-place / factory forms it knows replay into the SAME memory
    p = program.LaunchProgram(dev)
    with p.recording():
        z = torch.zeros(8, device=dev)
        z.add_(x)
        c = z.clone()
        s = torch.sin(c)  # a functional op with an out= overload: replayed into the recorded output
    # zeros / add_ / clone became entries of the C loop (dana_fill_zero, dana_axpy_rows, dana_copy_d2d); sin stays a torch op
    assert p.stats["torch_ops"] == 1 and p.stats["launches"] == 3
    z.fill_(7.0)
    c.fill_(9.0)
    s.fill_(0.0)
    p.run()
    torch.cuda.synchronize()
    assert torch.equal(z, x) and torch.equal(c, x) and torch.equal(s, torch.sin(x))


@pytest.mark.parametrize("rccl", [False, True], ids=["single", "rccl1rank"])
def test_program_training_iteration_equals_trainer_step(dev, rccl):
    """three iterations replayed from launch programs == three eager Trainer.step calls (SGD with momentum, weights updated
    in place under the programs). rccl1rank: a 1-rank RCCL group with always_reduce -> every bucket's all-reduce is a host
    callback inside the second program, issued where the eager backward issues it."""
    import torch.distributed as dist
    from dana_amd import synthetic as S
    from dana_amd.program import ProgramTrainer
    from dana_amd.trainer import Trainer
    if rccl:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29673"
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
        for it in range(2):
            np.random.seed(40 + it)
            t1.step(*inputs)
        before = _params(m1)
        np.random.seed(123)
        state = np.random.get_state()[1].copy()
        pt = ProgramTrainer(t1, *inputs, warmup=0)
        torch.cuda.synchronize()
        # recording runs a real iteration and puts everything back: parameters, step count, the host RNG
        assert np.array_equal(_params(m1), before) and t1.steps == 2
        assert np.array_equal(np.random.get_state()[1], state)
        st = pt.p2.stats
        assert pt.p1.stats["launches"] > 100 and st["launches"] > 250 and st["torch_ops"] <= 12 and st["streams"] >= 4
        assert st["host_callbacks"] == (1 + (len(t1.weights.buckets) + len(t1.biases.buckets)) if rccl else 1)
        for it in range(2, 5):
            np.random.seed(40 + it)
            out = pt.step(*inputs)
        torch.cuda.synchronize()
        got = _params(m1)
        d = np.abs(got - ref).max()
        assert d <= 1e-6 + 1e-4 * np.abs(ref).max(), d  # (RoIAlign-backward atomics are unordered)
        for a, b in zip([float(x) for x in out[3:7]], ref_losses):
            assert abs(a - b) <= 1e-4 * max(1.0, abs(b))
        assert t1.steps == 5
        # an EAGER step after replayed ones continues the same trajectory (derived weight copies re-derived from the live
        # weights), and so does a replay after that
        np.random.seed(45)
        t0.step(*inputs)
        np.random.seed(45)
        t1.step(*inputs)
        np.random.seed(46)
        t0.step(*inputs)
        np.random.seed(46)
        pt.step(*inputs)
        torch.cuda.synchronize()
        d = np.abs(_params(m1) - _params(m0)).max()
        assert d <= 1e-6 + 2e-4 * np.abs(ref).max(), d
    finally:
        if rccl:
            dist.destroy_process_group()


def test_program_training_iteration_with_device_rng_is_one_program(dev):
    """device RNG (row N2): no host round trip -> zero_grad .. forward .. backward .. SGD is ONE program; iteration k draws
    what the eager iteration number k draws (the Philox call counter is data in device memory)"""
    from dana_amd import synthetic as S
    from dana_amd.program import ProgramTrainer
    from dana_amd.trainer import Trainer
    inputs = [t.to(dev) for t in S.episode_inputs(1, 2, 2, 160, 224, seed=6)]
    m0, m1 = _model(dev), _model(dev)
    m0.device_rng = m1.device_rng = True
    t0, t1 = Trainer(m0, 0.01), Trainer(m1, 0.01)
    for _ in range(4):
        t0.step(*inputs)
    t1.step(*inputs)
    pt = ProgramTrainer(t1, *inputs, warmup=0)
    assert pt.p2 is None and pt.p1.stats["launches"] > 300 and m1._rng_calls == 1 and t1.steps == 1
    for _ in range(3):
        out = pt.step(*inputs)
    torch.cuda.synchronize()
    assert m1._rng_calls == 4 and t1.steps == 4
    a, b = _params(m0), _params(m1)
    d = np.abs(a - b).max()
    assert d <= 1e-6 + 1e-4 * np.abs(a).max(), d  # (RoIAlign-backward atomics are unordered)


def test_program_replay_needs_a_fraction_of_the_eager_host_time(dev):
    """the reason the programs exist: host time per training iteration (enqueue only, the host round trip's wait excluded)"""
    from dana_amd import ops, synthetic as S
    from dana_amd.program import ProgramTrainer
    from dana_amd.trainer import Trainer
    inputs = [t.to(dev) for t in S.episode_inputs(2, 2, 3, 320, 480, seed=6)]
    m = _model(dev, shot=3)
    tr = Trainer(m, 1e-3)
    pt = ProgramTrainer(tr, *inputs, warmup=2)

    def host_ms(step, n=6):
        best = 1e9
        for _ in range(n):
            torch.cuda.synchronize()
            ops.HOST_WAIT[0] = 0.0
            t0 = time.perf_counter()
            step(*inputs)
            dt = time.perf_counter() - t0 - ops.HOST_WAIT[0]
            best = min(best, dt * 1e3)
        torch.cuda.synchronize()
        return best

    eager, prog = host_ms(tr.step), host_ms(pt.step)
    print("host enqueue per training iteration: eager %.2f ms, launch programs %.2f ms" % (eager, prog))
    assert prog < 0.5 * eager
