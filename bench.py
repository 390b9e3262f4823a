#!/usr/bin/env python
"""bench.py -- DAnA forward hot path on MI355X (BASELINE.json metric: query-images/sec, res50,
way=2 shot=3 bs=4). One process per GPU; a "step" is ONE pass of the hot path -- the train-mode
DAnARCNN forward (trunk -> BA/CISA attention -> RPN -> proposals/NMS -> RoIAlign -> attention RCNN
head -> losses) -- over one batch of 4 synthetic episodes (600x1000 query + way*shot 320x320
supports; 320 because the reference hard-codes the 20x20 support map) that is
already resident in HBM. Weak scaling: every rank runs its own 4 episodes, no data-path collective
(episodes are independent).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel family = the fp32-MFMA implicit
GEMM, timed live with HIP events on its launch stream) and, at N=1, `cpu_baseline` (the oracle
port timed on the host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2516.6  # v_mfma_f32_32x32x16_bf16 dense: 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz
# fp32 contractions on the bf16 matrix cores cost six bf16 products per fp32 multiply (exact 3-way operand split)
PEAK_SPLIT_TFLOPS = round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="episodes per GPU per step")
    ap.add_argument("--way", type=int, default=2)
    ap.add_argument("--shot", type=int, default=3)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=1000)
    ap.add_argument("--model", default="DAnA", choices=["DAnA", "frcnn", "fsod", "meta", "fgn"],
                    help="DAnA: the hot path (default); frcnn / fsod / meta / fgn: the sibling detectors of utils.py:109-116 (row N4) "
                         "on the same operators -- forward modes (frcnn, meta: the training iteration too)")
    ap.add_argument("--trunk", type=int, default=50, choices=[50, 101],
                    help="101: the resnet101 trunk of resnet.py:199 ([3,4,23,3] blocks; BASELINE configs[3]). The reference "
                         "never builds it (dana.py:337), so there is no reference run -- oracle-checked only "
                         "(tests/test_gpu_model.py::test_res101_trunk_opt_in_vs_oracle)")
    ap.add_argument("--support-size", type=int, default=320,
                    help="support image side. 320 = the only size the reference can run (it hard-codes the 20x20 map); "
                         "224 (BASELINE.json's wording) runs the opt-in generalised pooling: NO oracle, no parity claim")
    ap.add_argument("--ba", dest="ba", action="store_true", default=True,
                    help="full BA+CISA attention = BASELINE configs[2], the reference's own default (utils.py:108, "
                         "train.py:72 never passes use_BA_block): the headline")
    ap.add_argument("--no-ba", dest="ba", action="store_false", help="CISA only (configs[1])")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the nested rocprofv3 --pmc passes (HBM traffic / MFMA busy of the dominant kernel); "
                         "roofline.traffic then comes from profiles/ and says so")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configs[1] (CISA only) measurement")
    ap.add_argument("--mode", default="train", choices=["train", "eval", "infer", "step"],
                    help="train: train-mode forward (variant F, the headline); eval: inference forward; infer: inference "
                         "forward + per-image detection post-processing (the loop of inference.py:96-142); step: the full "
                         "training iteration fwd+bwd+gradient all-reduce+SGD (variant S) as the headline")
    ap.add_argument("--no-train-step", action="store_true", help="skip the secondary variant-S measurement")
    ap.add_argument("--device-rng", action="store_true",
                    help="sample the training targets with the device Philox RNG (no host sync; NOT the reference's "
                         "np.random stream, which is the default and what the parity tests pin)")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "eager", "program"],
                    help="graph: replay captured hipGraphs (graphs.py: ~4x less host time per step); eager: one C-ABI call "
                         "per kernel from Python; auto (default): a short trial of both, the faster one is timed and both "
                         "trial figures are reported")
    ap.add_argument("--eager", action="store_true", help="= --launch eager")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--single-stream", action="store_true",
                    help="collapse the model's streams to one (kernels run alone: for rocprof per-kernel averages)")
    ap.add_argument("--dump-launches", default="", help="write the per-launch igemm table to this file")
    return ap.parse_args()


def config_label(args):
    """which BASELINE.json configuration the arguments are (per-GPU shape), or that they are none of them"""
    if args.model != "DAnA" or args.support_size != 320 or args.way != 2:
        return "no BASELINE configuration"
    if args.trunk == 101:
        return ("BASELINE configs[3] as far as it is defined (res101 trunk, per-GPU shape; way 2: the reference's positive / "
                "negative support split admits no third class, dana.py:103-104) -- no reference run exists, oracle-checked")
    if (args.height, args.width, args.shot) == (600, 1000, 3) and args.batch == 4:
        return "BASELINE configs[2]" if args.ba else "BASELINE configs[1]"
    if (args.height, args.width, args.shot) == (800, 1333, 10) and args.ba:
        return "BASELINE configs[4] (per-GPU shape: %d of its 16 episodes%s)" % (
            args.batch, "" if args.batch == 2 else "; the configuration puts 2 on each of 8 GPUs")
    return "no BASELINE configuration (configs[2]'s model at another shape)"


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(args, sd):
    """The oracle port (oracle/model_ref.py) on the host cores: ONE episode of the same workload, timed with all cores
    (capped at 64 threads) and, SURVEY 8d, with 8 threads."""
    from dana_amd import synthetic as S
    from oracle import model_ref as O
    way = args.way if args.mode == "train" else 1
    inputs = S.episode_inputs(1, way, args.shot, args.height, args.width, seed=1996)

    def one():
        with torch.no_grad():
            O.forward(sd, *inputs, args.mode == "train", way, args.shot, args.ba, nms_inclusive=False)

    def sample(threads, budget_s, max_reps):
        torch.set_num_threads(threads)
        np.random.seed(3)
        one()  # warm-up (thread pools, oneDNN primitive caches)
        reps, t0 = 0, time.time()
        while reps < max_reps and (time.time() - t0) < budget_s:  # bounded sample
            one()
            reps += 1
        return (time.time() - t0) / reps, reps

    cores = min(os.cpu_count() or 1, 64)  # torch-CPU convs stop scaling (and oversubscribe) beyond ~64 threads
    runs = []
    for threads, budget, reps_max in ((cores, 10.0, 12),) + (((8, 8.0, 8),) if cores > 8 else ()):
        dt, reps = sample(threads, budget, reps_max)  # together ~15-20 s of CPU work
        runs.append({"value": round(1.0 / dt, 4), "unit": "query-images/sec", "cores": threads,
                     "sample": "%d x 1 episode, %d threads, %.2f s/episode" % (reps, threads, dt)})
    torch.set_num_threads(cores)
    best = max(runs, key=lambda r: r["value"])  # (on a many-core host the 8-thread run can beat the all-cores one)
    return {"value": best["value"], "unit": "query-images/sec", "cores": best["cores"], "kind": "port",
            "cpu": cpu_model_name(), "host_logical_cpus": os.cpu_count(), "torch": torch.__version__,
            "sample": "1 episode (1 query %dx%d + %d supports 320x320) per repetition, %s-mode forward, oracle/model_ref.py on "
                      "torch-CPU fp32; value = the faster of the thread counts sampled" % (
                          args.height, args.width, way * args.shot, args.mode),
            "thread_counts": runs}


def pmc_passes(args, kernel):
    """Nested `rocprofv3 --pmc <group> --kernel-trace` runs of this script (few steps, single stream, no roofline / CPU
    legs), one counter group per pass -> per-launch HBM bytes and matrix-core busy fraction of `kernel`.
    Returns None when rocprofv3 is unavailable or a pass fails (the caller then falls back to profiles/)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None
    base = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline",
            "--no-train-step", "--single-stream", "--no-pmc", "--no-secondary", "--batch", str(args.batch),
            "--way", str(args.way), "--shot", str(args.shot), "--height", str(args.height), "--width", str(args.width),
            "--mode", args.mode] + ([] if args.ba else ["--no-ba"])
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    short = kernel.replace(" ", "").rstrip(">")  # (prefix match: the BPRE template argument follows BM, BN, STEM)
    sums = {}
    for group in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]):
        out = tempfile.mkdtemp(prefix="dana_pmc_", dir="/tmp")
        cmd = [rp, "--pmc"] + group + ["--kernel-trace", "--output-format", "csv", "-d", out, "-o", "pmc", "--"] + base
        try:
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                  start_new_session=True)
            try:
                pr.wait(timeout=240)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)
                return None
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if pr.returncode != 0 or not files:
                return None
            with open(files[0]) as fh:
                for r in csv.DictReader(fh):
                    if r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace(" ", "").startswith(short):
                        a = sums.setdefault(r["Counter_Name"], [0.0, 0])
                        a[0] += float(r["Counter_Value"])
                        a[1] += 1
        finally:
            shutil.rmtree(out, ignore_errors=True)
    if "FETCH_SIZE" not in sums or "WRITE_SIZE" not in sums:
        return None
    kb = 2.0 * sums["FETCH_SIZE"][0] / sums["FETCH_SIZE"][1] + sums["WRITE_SIZE"][0] / sums["WRITE_SIZE"][1]
    res = {"traffic": round(kb * 1024.0),
           "traffic_unit": "HBM bytes per %s launch: (2 x FETCH_SIZE + WRITE_SIZE) KB, rocprofv3 --pmc, this run" % kernel,
           "traffic_launches_sampled": sums["FETCH_SIZE"][1]}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in sums and "GRBM_GUI_ACTIVE" in sums and sums["GRBM_GUI_ACTIVE"][0] > 0:
        # SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMD matrix pipes (256 CUs x 4); rocprofv3 sums
        # GRBM_GUI_ACTIVE over the 8 XCDs, so /8 is the launch's duration in shader cycles -> fraction of the chip's
        # matrix-pipe cycles spent executing MFMAs (clock-independent)
        res["mfma_busy"] = round(sums["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (sums["GRBM_GUI_ACTIVE"][0] / 8.0 * 1024.0), 4)
        res["mfma_busy_is"] = "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) over the %s launches" % kernel
    return res


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: re-execute this very command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` (one rank per GPU, rendezvous on 127.0.0.1, a free
    port), hand its output through and return its exit status. The launch contract of DESIGN.md 6: `--gpus` IS the
    number of ranks; under an external launcher WORLD_SIZE must agree with it."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(args.gpus, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # (the driver's N > 1 command passes both; a mismatch would print a line whose n_gpus is not what was asked for)
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE line, the JSON: everything else this process (or a C library inside it: RCCL's version banner)
    # writes to descriptor 1 goes to stderr from here on, the JSON line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch.distributed as dist
    # test hook (1-GPU boxes): DANA_BENCH_BACKEND=gloo runs every rank on cuda:0 and exchanges over gloo, to exercise
    # the multi-rank control flow where RCCL (one rank per device) cannot; the driver's runs use nccl = RCCL
    backend = os.environ.get("DANA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    _t_start = time.perf_counter()

    def phase(name):
        if os.environ.get("DANA_BENCH_TIMING") and rank == 0:
            print("[bench %7.1f s] %s" % (time.perf_counter() - _t_start, name), file=sys.stderr, flush=True)

    def dist_barrier():
        if backend == "nccl":
            dist.barrier(device_ids=[local])
        else:
            dist.barrier()

    def flush_c_stdout():
        """RCCL prints a version banner through C stdio when its first communicator comes up; on a pipe that buffer is
        flushed at process exit, i.e. BEHIND rank 0's JSON line. Push it out now: the JSON line stays the last line."""
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass

    rccl_ranks_seen = None
    if world > 1:  # one tiny all_reduce before anything else: how many ranks does the collective backend really span?
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        rccl_ranks_seen = int(one.item())
        flush_c_stdout()
        if rccl_ranks_seen != world:
            raise SystemExit("bench.py: the %s process group spans %d ranks, %d were launched" % (backend, rccl_ranks_seen, world))

    import dana_amd
    from dana_amd import ops, synthetic as S
    training = args.mode in ("train", "step")
    way = args.way if training else 1
    if args.model != "DAnA" and args.mode not in ("train", "eval", "step"):
        raise SystemExit("--model %s supports --mode train / eval / step" % args.model)
    def build_model(name, ba, way_, shot_, trunk=50):
        if name == "DAnA" and trunk == 101:
            from dana_amd.dana import DAnARCNN
            m_ = DAnARCNN(["fg", "bg"], "concat", 256, 256, pretrained=False, semantic_enhance=ba, num_way=way_, num_shot=shot_)
            m_.trunk_layers = (3, 4, 23, 3)
            m_.create_architecture()
            return m_
        return dana_amd.get_model(name, pretrained=False, use_BA_block=ba, way=way_, shot=shot_, classes=["fg", "bg"])

    model = build_model(args.model, args.ba, args.way, args.shot, args.trunk)
    sd = S.fill_state_dict(model.state_dict(), seed=11, profile="test")  # random init, O(1) activations
    if args.trunk == 101 and args.model == "DAnA":
        sd = S.tame_res101_weights(sd)
    model.load_state_dict(sd)
    model.to(dev)
    model.train() if training else model.eval()
    model._single_stream = bool(args.single_stream)
    model.device_rng = bool(args.device_rng)
    model.generalised_support = args.support_size != 320
    # every rank gets its own episodes (weak scaling), already resident in HBM before the timed region
    inputs = [t.to(dev) for t in S.episode_inputs(args.batch, way, args.shot, args.height, args.width,
                                                  seed=1996 + rank, support_size=args.support_size)]

    trainer = [None]

    def train_step():
        # train.py:125-143: zero_grad, forward, summed loss, backward (HIP kernels, bucketed RCCL all-reduce of the
        # gradients overlapped with it), fused SGD
        if trainer[0] is None:
            from dana_amd.trainer import Trainer
            trainer[0] = Trainer(model, lr=1e-5)
        return trainer[0].step(*inputs)

    if args.model == "frcnn":
        inputs = inputs[:4]  # faster_rcnn.py:35: (im_data, im_info, gt_boxes, num_boxes)
    elif args.model == "meta":
        inputs = inputs + [inputs[2].clone()]  # meta.py:39,48: all_cls_gt_boxes (one class in the synthetic episodes)
    if args.model == "fsod":  # same test-profile taming as the goldens: its correlations sum 49 products per channel
        model.load_state_dict(S.tame_fsod_weights(sd))
    elif args.model == "fgn":
        model.load_state_dict(S.tame_fgn_weights(sd))

    def fwd_step():
        with torch.no_grad():
            return model(*inputs)

    def infer_step():
        # inference.py:96-142: forward, then per image de-normalise / decode / clip / threshold / sort / NMS (one C call)
        from dana_amd.postprocess import detections
        with torch.no_grad():
            rois, cls_prob, bbox_pred = model(*inputs)[:3]
            return [detections(rois[i:i + 1], cls_prob[i * rois.size(1):(i + 1) * rois.size(1)],
                               bbox_pred[i * rois.size(1):(i + 1) * rois.size(1)], inputs[1][i:i + 1])
                    for i in range(rois.size(0))]

    eager_step = {"step": train_step, "infer": infer_step}.get(args.mode, fwd_step)
    step = eager_step
    if args.eager:
        args.launch = "eager"
    replayable = args.model == "DAnA" and not args.single_stream
    use_graphs = args.launch in ("auto", "graph") and replayable
    use_programs = args.launch in ("auto", "program") and replayable and not args.device_rng
    np.random.seed(1996 + rank)
    graphed = {}
    cands = {"eager": eager_step}

    def with_postprocess(run):
        def step_():
            from dana_amd.postprocess import detections
            rois, cls_prob, bbox_pred = run(*run.inputs)[:3]
            return [detections(rois[i:i + 1], cls_prob[i * rois.size(1):(i + 1) * rois.size(1)],
                               bbox_pred[i * rois.size(1):(i + 1) * rois.size(1)], inputs[1][i:i + 1])
                    for i in range(rois.size(0))]
        return step_

    def all_agree(flag):
        """1.0 on every rank -> True (a launch mode is used by all ranks or by none)"""
        if world > 1:
            t_ = torch.tensor([flag], device=dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MIN)
            flag = float(t_.item())
        return flag > 0.5

    if use_graphs:
        # hipGraph replay (graphs.py): the forward = two captured graphs around the one host round trip (the reference's
        # np.random draws need the fg / bg counts), the training iteration likewise (+ backward + SGD; multi-rank: cut
        # once more so that the RCCL all-reduce overlaps the trunk's backward)
        from dana_amd.graphs import GraphedDAnA, GraphedTrainer
        if args.mode == "step":
            train_step()  # builds the trainer
            gtr = graphed["trainer"] = GraphedTrainer(trainer[0], *inputs)
            cands["graph"] = lambda: gtr.step(*gtr.inputs)  # noqa: E731
        else:
            run = graphed["forward"] = GraphedDAnA(model, *inputs)
            cands["graph"] = with_postprocess(run) if args.mode == "infer" else (lambda: run(*run.inputs))
    if use_programs:
        # launch-program replay (program.py): the EAGER step's own launches on the eager step's streams, recorded once and
        # re-issued from one Python loop -- two programs around the host round trip; multi-rank: the bucket all-reduces
        # are host callbacks inside the second one
        from dana_amd.program import ProgramDAnA, ProgramTrainer
        ok_, why_ = 1.0, None
        try:
            if args.mode == "step":
                train_step()
                ptr = graphed["program_trainer"] = ProgramTrainer(trainer[0], *inputs)
                cands["program"] = lambda: ptr.step(*ptr.inputs)  # noqa: E731
            else:
                prun = graphed["program_forward"] = ProgramDAnA(model, *inputs)
                cands["program"] = with_postprocess(prun) if args.mode == "infer" else (lambda: prun(*prun.inputs))
        except Exception as e_:  # noqa: BLE001  (a replay mode that cannot be recorded here must not cost the line)
            ok_, why_ = 0.0, "%s: %s" % (type(e_).__name__, str(e_)[:160])
        ok_ = all_agree(ok_)
        if not ok_:
            cands.pop("program", None)
            use_programs = False
            graphed["program_error"] = why_ or "another rank could not record its programs"
    graph_step = cands.get("graph")

    def median_interval(evs):
        """median GPU-side interval between consecutive iteration-end events (ms): robust against the host stalls of a
        shared box, reported NEXT to the wall-clock figure the contract asks for"""
        iv = sorted(a.elapsed_time(b) for a, b in zip(evs, evs[1:]))
        return round(iv[len(iv) // 2], 3) if iv else None

    def trial(fn, k):
        """ms per step of k steps: the MEDIAN GPU-side interval between consecutive steps (one slow step -- a collector
        pause, the first eager steps after a capture re-deriving their plan -- must not decide the launch mode), falling
        back to the wall-clock mean where no median is available"""
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True)]
        marks[0].record()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
        torch.cuda.synchronize()
        mean = (time.perf_counter() - t0) / k * 1e3
        return median_interval(marks) or mean

    def pick_launch(c, k, forced=None):
        """time every candidate launch mode of one step (k steps each, two interleaved rounds, the better round counts) and
        pick the fastest; every rank takes the same one (times are max-reduced over the ranks first: N ranks share a host).
        -> (name, {name_ms_per_step: ...})"""
        names = sorted(c)
        if forced in c:
            return forced, None
        if len(names) == 1:
            return names[0], None
        best = {n: 1e30 for n in names}
        for _ in range(2 if world == 1 else 1):  # (N ranks: every trial step is a collective step; one round)
            for n in names:
                best[n] = min(best[n], trial(c[n], k))
        t = torch.tensor([best[n] for n in names], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        vals = [float(x) for x in t.tolist()]
        tr_ = {"%s_ms_per_step" % n: round(v, 3) for n, v in zip(names, vals)}
        tr_["steps_each"] = k
        return names[vals.index(min(vals))], tr_

    phase("launch modes built (graph capture / program recording)")
    chosen, launch_trial = pick_launch(cands, max(5, min(20, args.steps)), None if args.launch == "auto" else args.launch)
    step = cands[chosen]
    timed_graphs = chosen == "graph"
    for _ in range(args.warmup):
        step()

    def host_enqueue_ms(fn, k=20):
        """host time per step spent ISSUING work (Python + runtime calls), excluding the time blocked on the GPU in the
        training forward's one D2H read"""
        torch.cuda.synchronize()
        ops.HOST_WAIT[0] = 0.0
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        dt_ = time.perf_counter() - t0 - ops.HOST_WAIT[0]
        torch.cuda.synchronize()
        return round(1e3 * dt_ / k, 3)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist_barrier()
        torch.cuda.synchronize()

    phase("launch trial done: %s" % chosen)
    barrier()
    marks_f = [torch.cuda.Event(enable_timing=True)]
    marks_f[0].record()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        marks_f.append(torch.cuda.Event(enable_timing=True))
        marks_f[-1].record()
    barrier()
    dt = time.perf_counter() - t0
    per_rank_ms = None
    if world > 1:
        every = torch.zeros(world, device=dev, dtype=torch.float64)
        every[rank] = dt
        dist.all_reduce(every)  # (a straggler must be visible: every rank's own time next to the max the metric uses)
        per_rank_ms = [round(1000.0 * float(x) / args.steps, 3) for x in every.tolist()]
        dt = float(every.max().item())

    host_ms = {n: host_enqueue_ms(f, 10 if n == "eager" else 20) for n, f in cands.items()}
    what = {"step": "training step fwd+bwd+allreduce+SGD",
            "infer": "eval-mode forward + detection post-processing per image"}.get(args.mode, "%s-mode forward" % args.mode)
    result = {
        "metric": "query-images/sec (res50, way=%d, shot=%d, bs=%d per GPU, %s)" % (
            args.way, args.shot, args.batch, what),
        "value": round(world * args.batch * args.steps / dt, 3),
        "unit": "query-images/sec",
        "n_gpus": world,
        "rccl_ranks_seen": rccl_ranks_seen,  # sum of ones over the process group (None at N = 1: no group)
        "collective_backend": (("RCCL (torch 'nccl')" if backend == "nccl" else backend) if world > 1 else None),
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1000.0 * dt / args.steps, 3),
        "ms_per_episode": round(1000.0 * dt / args.steps / args.batch, 3),
        "ms_per_step_median": median_interval(marks_f),
        "ms_per_step_per_rank": ({"min": min(per_rank_ms), "max": max(per_rank_ms), "all": per_rank_ms}
                                 if per_rank_ms else None),
        "launch": ("hipGraph replay, %d graph launches per step (graphs.py)" % (
            (len(graphed["trainer"].graphs) + (graphed["trainer"].g_anchor is not None)) if args.mode == "step" else
            (1 + (graphed["forward"].g0 is not None) + (graphed["forward"].g2 is not None))))
        if timed_graphs else ("launch-program replay: the eager step's launches on its own streams, re-issued from one loop "
                              "(program.py)" if chosen == "program" else "eager (one C-ABI call per kernel from Python)"),
        "launch_trial": launch_trial,
        "launch_program_error": graphed.get("program_error"),
        "host_enqueue_ms_per_step": host_ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 (operands, accumulation, results; multiplies as an exact bf16x3 split, six bf16 MFMA products each)"
                 if ops.get_mfma_mode() else "f32",
        "data": "synthetic (seeded N(0,64^2) query/support pixels, 3 gt boxes/image, random-init weights)",
        "config": {"workload": ("" if args.model == "DAnA" else "[sibling detector '%s'] " % args.model) +
                               "%s: res50 way=%d shot=%d bs=%d, %dx%d queries + %d %dx%d "
                               "supports/episode%s, %s, %s" % (
                                   config_label(args), args.way, args.shot, args.batch, args.height, args.width,
                                   way * args.shot, args.support_size, args.support_size,
                                   "" if args.support_size == 320 else " (generalised support pooling: NOT a reference "
                                   "configuration, no oracle)", "BA+CISA" if args.ba else "CISA only", what),
                   "global_batch": world * args.batch, "parallelism": "episodes sharded, %d rank(s)" % world,
                   "target_sampling": "device Philox RNG" if args.device_rng else "host np.random (reference stream)",
                   "configs_not_run": "configs[3] as written (res101, way 5) cannot run on the reference (dana.py:337 builds "
                                      "resnet50() whatever num_layers says; way > 2 breaks dana.py:103-108): its res101 trunk is "
                                      "an opt-in here (`--trunk 101`, line `configs_3`, oracle-checked), way stays 2; configs[4] "
                                      "(800x1333, shot 10) is covered by tests/test_gpu_model.py::test_config4_* and line `configs_4`",
                   "contractions": ("fp32 operands and accumulation; multiplies as an exact 3-way bf16 split, six products on "
                                    "v_mfma_f32_32x32x16_bf16 (error vs fp64 at the f32-MFMA kernel's level; "
                                    "f32_mfma_only = the same step with v_mfma_f32_32x32x2_f32)")
                                   if ops.get_mfma_mode() else "v_mfma_f32_32x32x2_f32"},
    }

    phase("timed region + host enqueue times done")
    if args.mode == "train" and not args.no_train_step:
        # secondary measurement, every rank: variant S, the full training iteration with the
        # gradient all-reduce over RCCL as its one exchange step. Same episodes, same timing protocol.
        ks, kw = max(3, min(args.steps, 20)), 5  # the first iterations grow the allocator's pools
        for _ in range(kw):
            train_step()
        eager_train_step = train_step
        # (gloo moves the 148 MB through the host, ~0.4 s per iteration: the test-only backend gets shorter side measurements)
        slow_exchange = world > 1 and backend != "nccl"
        ts_cands = {"eager": eager_train_step}
        if use_graphs:
            from dana_amd.graphs import GraphedTrainer
            gtr = graphed["trainer"] = GraphedTrainer(trainer[0], *inputs)
            ts_cands["graph"] = lambda: gtr.step(*gtr.inputs)  # noqa: E731
        if use_programs:
            from dana_amd.program import ProgramTrainer
            ok_ = 1.0
            try:
                ptr = graphed["program_trainer"] = ProgramTrainer(trainer[0], *inputs)
                ts_cands["program"] = lambda: ptr.step(*ptr.inputs)  # noqa: E731
            except Exception as e_:  # noqa: BLE001
                ok_ = 0.0
                graphed["program_error"] = "%s: %s" % (type(e_).__name__, str(e_)[:160])
            if not all_agree(ok_):
                ts_cands.pop("program", None)
        ts_chosen, ts_trial = pick_launch(ts_cands, max(3, min(12, args.steps)), None if args.launch == "auto" else args.launch)
        train_step = ts_cands[ts_chosen]
        for _ in range(3):
            train_step()
        barrier()
        marks_s = [torch.cuda.Event(enable_timing=True)]
        marks_s[0].record()
        t0 = time.perf_counter()
        for _ in range(ks):
            train_step()
            marks_s.append(torch.cuda.Event(enable_timing=True))
            marks_s[-1].record()
        barrier()
        dts = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dts], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
        tr = trainer[0]
        result["train_step"] = {
            "what": "train.py:125-143 iteration: forward + backward (HIP kernels) + bucketed gradient all-reduce "
                    "(%s) + fused SGD" % ("%s, %d ranks" % ("RCCL" if backend == "nccl" else backend, world) if world > 1
                                       else "single rank: no exchange"),
            "value": round(world * args.batch * ks / dts, 3), "unit": "query-images/sec", "steps": ks, "warmup": kw,
            "ms_per_step": round(1000.0 * dts / ks, 3), "ms_per_step_median": median_interval(marks_s),
            "gradient_mbytes": round(4e-6 * sum(fb.numel for fb, _, _ in tr.groups), 1),
            "buckets": sum(len(fb.buckets) for fb, _, _ in tr.groups),
            "launch": {"graph": "hipGraph replay", "program": "launch-program replay", "eager": "eager"}[ts_chosen],
            "launch_trial": ts_trial,
            "host_enqueue_ms_per_step": {n: host_enqueue_ms(f, 3 if slow_exchange else (5 if n == "eager" else 10))
                                         for n, f in ts_cands.items()},
        }

        # The exchange, isolated (north_star's "RCCL all-reduce on the loss gradients only"; train.py:104-105,138-139): (a) the
        # bucketed all-reduce ALONE -- every bucket issued from the exchange stream with nothing else on the GPU, bracketed
        # by events on the caller's stream; (b) what of it the iteration does not hide behind the backward: the eager
        # iteration with the collectives against the same iteration without them, interleaved. Every N, every rank; a
        # single-GPU run drives a 1-rank RCCL group (a sum over one rank: same launches, same bytes through the library, no
        # wire) -- its train_step figures above stay the no-exchange iteration.
        try:
            fbs = [fb for fb, _, _ in tr.groups]
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                dist.init_process_group("nccl" if backend == "nccl" else backend, rank=0, world_size=1,
                                        **({"device_id": dev} if backend == "nccl" else {}))
            coll_prev = [fb.collective for fb in fbs]

            def set_coll(on):
                for fb in fbs:
                    fb.collective = on

            set_coll(True)
            for _ in range(2):
                eager_train_step()  # (communicator warm-up: the first collectives build RCCL's channels)
            torch.cuda.synchronize()
            flush_c_stdout()
            alone = []
            for _ in range(3 if slow_exchange else 7):
                barrier()
                for fb in fbs:
                    fb.zero_grad_bookkeeping()
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                for fb in fbs:
                    for i in range(len(fb.buckets)):
                        fb._works.append(fb.reduce_bucket(i))
                    fb._pending = [0] * len(fb.buckets)
                for fb in fbs:
                    fb.wait_all()
                ev1.record()
                torch.cuda.synchronize()
                alone.append(ev0.elapsed_time(ev1))
            t_on, t_off = [], []
            for _ in range(1 if slow_exchange else 2):
                set_coll(True)
                barrier()
                t_on.append(trial(eager_train_step, 3 if slow_exchange else 6))
                set_coll(False)
                barrier()
                t_off.append(trial(eager_train_step, 3 if slow_exchange else 6))
            for fb, c in zip(fbs, coll_prev):
                fb.collective = c
            ex = torch.tensor([sorted(alone)[len(alone) // 2], min(t_on), min(t_off)], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(ex, op=dist.ReduceOp.MAX)
            nbytes = 4 * sum(fb.numel for fb in fbs)
            result["train_step"]["exchange"] = {
                "ranks": world, "backend": "RCCL" if backend == "nccl" else backend,
                "allreduce_ms_alone": round(float(ex[0]), 3), "bytes": nbytes, "buckets": sum(len(fb.buckets) for fb in fbs),
                "algbw_GBps_alone": round(nbytes / (float(ex[0]) * 1e-3) / 1e9, 1),
                "iteration_ms_with_exchange": round(float(ex[1]), 3), "iteration_ms_without_exchange": round(float(ex[2]), 3),
                "exposed_ms": round(float(ex[1]) - float(ex[2]), 3),
                "timing": "eager issue, median GPU-side interval of 6 iterations, best of 2 interleaved rounds; max over ranks"}
        except Exception as e:  # noqa: BLE001 (a box without a usable RCCL must still print the line)
            result["train_step"]["exchange"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}

    tprof, tprof_steps = None, 0
    if "train_step" in result and not args.no_roofline:
        # variant S's roofline: per-launch brackets over the whole training iteration (forward + data gradients + weight
        # gradients), one stream. EVERY rank runs these iterations (they carry the gradient all-reduce: a rank-0-only
        # pass would wait for its peers forever); rank 0 reports its own launches.
        ops.PROFILE = []
        model._single_stream = True
        tprof_steps = max(2, min(5, args.steps))
        for _ in range(2):
            eager_train_step()
        torch.cuda.synchronize()
        ops.PROFILE = []
        for _ in range(tprof_steps):
            eager_train_step()
        torch.cuda.synchronize()
        model._single_stream = bool(args.single_stream)
        tprof, ops.PROFILE = ops.PROFILE, None
    phase("train_step (incl. exchange) done")
    if rank == 0 and not args.no_roofline and args.mode != "step":
        # dominant kernel family: the implicit-GEMM contraction (every conv / Linear / bmm). Same K steps, each launch
        # bracketed by HIP events recorded on the stream the kernel is launched on (torch's current stream).
        # The product overlaps independent branches on several streams; for a per-kernel duration the
        # timing pass runs them on ONE stream so that every launch is measured alone on the chip.
        def contraction_pass():
            ops.PROFILE = []
            model._single_stream = True
            torch.cuda.synchronize()
            bounds = [0]
            for _ in range(args.steps):
                eager_step()
                bounds.append(len(ops.PROFILE))
            torch.cuda.synchronize()
            model._single_stream = bool(args.single_stream)
            prof, ops.PROFILE = ops.PROFILE, None
            contraction_pass.bounds = bounds
            return prof

        def launch_table(rows, bounds):
            """(index in the step, tag) -> [flops, mean ms, executed flops] over the steps with the usual launch count (a step
            that rebuilds a cached operand has one launch more and would shift every index behind it)"""
            counts = [b - a for a, b in zip(bounds, bounds[1:])]
            usual = max(set(counts), key=counts.count)
            steps_ = [(a, b) for a, b in zip(bounds, bounds[1:]) if b - a == usual]
            per = {}
            for a, b in steps_:
                for i, row in enumerate(rows[a:b]):
                    e = per.setdefault((i, row[0]), [row[1], 0.0, row[5]])
                    e[1] += row[2].elapsed_time(row[3]) / len(steps_)
            return per

        def family(rows):
            f = sum(r[1] for r in rows)
            x = sum(r[5] for r in rows)
            t = sum(r[2].elapsed_time(r[3]) for r in rows) * 1e-3
            return f, x, t

        split = ops.get_mfma_mode() != 0
        prof = contraction_pass()
        prof_bounds = contraction_pass.bounds
        flops, executed, secs = family(prof)
        ms = secs * 1e3
        launches = len(prof) // args.steps
        achieved = flops / secs / 1e12
        by = {}
        for row in prof:
            a = by.setdefault(row[0].split(" ")[0], [0.0, 0.0, 0])
            a[0] += row[1]
            a[1] += row[2].elapsed_time(row[3])
            a[2] += 1
        mfma = [r for r in prof if not r[0].startswith("conv7x7") and " N=2 " not in r[0] and " N=4 " not in r[0]]
        peak = PEAK_SPLIT_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
        wino = [r for r in prof if r[0].startswith("wino")]
        direct = [r for r in prof if not r[0].startswith("wino")]
        fw, xw, tw = family(wino) if wino else (0.0, 0.0, 1e-9)
        fd, xd, td = family(direct)
        traffic_kernel = "igemm_split_kernel_128<0" if split else "igemm_f32_kernel<64, 64, 0>"  # (the 128 x 128 tile; prefix match: the BPRE argument follows)
        result["roofline"] = {
            "bound": "mfma",
            "kernel": ("igemm_split_kernel (fp32 operands split exactly into 3 bf16 each, 6 x v_mfma_f32_32x32x16_bf16 per "
                       "K=16, fp32 accumulation)" if split else
                       "igemm_f32_kernel (v_mfma_f32_32x32x2_f32 implicit GEMM)"),
            "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None,
            "algorithmic_bytes_per_launch": round(sum(r[4] for r in mfma) / max(len(mfma), 1)),
            "launches_per_step": launches,
            "algorithmic_gflop_per_step": round(flops / args.steps / 1e9, 1),
            "executed_gflop_per_step": round(executed / args.steps / 1e9, 1),
            "kernel_ms_per_step": round(ms / args.steps, 3),
            "kernel_ms_per_step_is": "SERIAL sum of the launches' durations, each bracketed ALONE on one stream; the timed "
                                     "step overlaps the query and support trunks on two streams, so ms_per_step can be smaller "
                                     "-- frac is the conservative per-launch figure, whole_step_tflops the overlapped one",
            # the two sub-families, separately: launches that run the DIRECT contraction (algorithmic == executed
            # multiply-adds) and the Winograd F(4x4,3x3) launches (input transform + 36 plane GEMMs + output transform
            # in one timed bracket: 4x fewer multiply-adds than the algorithmic figure they are priced at)
            "families": {
                "direct": {"algorithmic_tflops": round(fd / td / 1e12, 2), "frac_of_peak": round(fd / td / 1e12 / peak, 4),
                           "ms_per_step": round(td * 1e3 / args.steps, 3), "gflop_per_step": round(fd / args.steps / 1e9, 1)},
                "winograd": {"algorithmic_tflops": round(fw / tw / 1e12, 2), "executed_tflops": round(xw / tw / 1e12, 2),
                             "executed_frac_of_peak": round(xw / tw / 1e12 / peak, 4),
                             "ms_per_step": round(tw * 1e3 / args.steps, 3),
                             "algorithmic_gflop_per_step": round(fw / args.steps / 1e9, 1),
                             "executed_gflop_per_step": round(xw / args.steps / 1e9, 1)} if wino else None,
            },
            "by_kind_tflops": {k: round(v[0] / (v[1] * 1e-3) / 1e12, 2) for k, v in by.items()},
            "whole_step_tflops": round(flops / args.steps / (dt / args.steps) / 1e12, 2),
        }
        if split:
            # `achieved` counts ALGORITHMIC fp32 FLOPs (2 per multiply-add of the direct convolution); the matrix cores
            # issue six bf16 products for each EXECUTED multiply-add, so the peak is the bf16 dense peak / 6.
            result["roofline"]["peak_is"] = "%.1f TFLOP/s bf16 dense MFMA / 6 products per fp32 multiply" % PEAK_BF16_MFMA_TFLOPS
            result["roofline"]["mfma_issued_tflops"] = round(6.0 * executed / secs / 1e12, 1)  # what the matrix cores ran
            result["roofline"]["mfma_issued_frac_of_bf16_peak"] = round(6.0 * executed / secs / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)
            result["roofline"]["vs_f32_mfma_peak"] = round(achieved / PEAK_FP32_MFMA_TFLOPS, 4)
            # The denominators, named. `peak` is the arithmetic ceiling of the six-product kernel at the
            # 2.4 GHz the data sheet quotes; the chip does not hold that clock on this kernel with real data (the shader
            # clock follows the number of busy matrix pipes: profiles/r6_overlap.md, r2_gemm_power.md). `attainable` is what
            # the SAME kernel reaches on one chip-filling GEMM (16384 x 4096 x 4096, N(0,1) operands, weights pre-split)
            # measured in THIS run on THIS box -- no tails, no launch gaps, no fixed phases: the distance of the step from
            # what this kernel can do at all.
            try:
                gm, gn, gk = 16384, 4096, 4096
                ga = torch.randn(gm, gk, device=dev)
                gb = ops.split_weight(torch.randn(gn, gk, device=dev) * 0.05, gn, gk)
                gc_ = torch.empty(gm, gn, device=dev)
                for _ in range(3):
                    ops.gemm_nt(ga, gb, gm, gn, gk, out=gc_, ldc=gn)
                torch.cuda.synchronize()
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
                for _ in range(12):
                    ops.gemm_nt(ga, gb, gm, gn, gk, out=gc_, ldc=gn)
                g1.record()
                torch.cuda.synchronize()
                att = 2.0 * gm * gn * gk * 12 / (g0.elapsed_time(g1) * 1e-3) / 1e12
                del ga, gb, gc_
                result["roofline"]["attainable"] = round(att, 1)
                result["roofline"]["attainable_is"] = ("algorithmic TFLOP/s of the same six-product kernel on ONE 16384x4096x4096 "
                                                       "GEMM of N(0,1) data, this run, this box (12 launches back to back)")
                result["roofline"]["frac_of_attainable"] = round(achieved / att, 4)
                result["roofline"]["executed_frac_of_attainable"] = round(executed / secs / 1e12 / att, 4)
                # ... and the timed step itself (two trunk streams overlapped, every non-contraction kernel and gap included)
                # against the same figure: how much of the step's wall clock is NOT already the kernel's attainable rate
                result["roofline"]["whole_step_frac_of_attainable"] = round(result["roofline"]["whole_step_tflops"] / att, 4)
                # (the Winograd launches are PRICED at the direct conv's FLOPs and execute 4x fewer: the same ratio in
                #  EXECUTED multiply-adds, which is what a plain GEMM's `attainable` counts)
                result["roofline"]["whole_step_executed_frac_of_attainable"] = round(
                    executed / args.steps / (dt / args.steps) / 1e12 / att, 4)
                result["roofline"]["frac_of_bf16_peak"] = result["roofline"]["mfma_issued_frac_of_bf16_peak"]
            except Exception as e:  # noqa: BLE001  (never lose the line over a side measurement)
                result["roofline"]["attainable_error"] = str(e)[:120]
        if args.dump_launches:
            per = launch_table(prof, prof_bounds)
            with open(args.dump_launches, "w") as fh:
                for (i, tag), (f, t, x) in sorted(per.items()):
                    fh.write("%3d %-40s %9.2f GF %9.1f us %7.2f TF/s (executed %7.2f)\n" % (
                        i, tag, f / 1e9, t * 1e3, f / t / 1e9, x / t / 1e9))
        # HBM traffic and matrix-core busy cycles of the dominant kernel from PMC counters, measured NOW: nested
        # rocprofv3 --pmc passes over this very command (short, single-stream), one counter group per pass, corrected as
        # MI355X_MICROARCH.md prescribes (gfx950 FETCH_SIZE counts 64 B per 128-B request of wide reads: doubled).
        pmc = None if (args.no_pmc or world > 1) else pmc_passes(args, traffic_kernel)  # (one process per box only)
        if pmc is not None:
            result["roofline"].update(pmc)
        else:
            try:
                with open(os.path.join(ROOT, "profiles", "r6_pmc_traffic.json")) as fh:
                    j = json.load(fh)
                pre = traffic_kernel.replace(" ", "").rstrip(">")
                ms_ = [v for k, v in j.items() if k.replace(" ", "").startswith(pre)]
                result["roofline"]["traffic"] = round(sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in ms_) /
                                                      sum(v["launches"] for v in ms_))
                result["roofline"]["traffic_source"] = "profiles/r6_pmc_traffic.json (committed PMC run, not this run)"
            except (OSError, KeyError, ValueError, ZeroDivisionError):
                pass
        if tprof is not None:
            kts = tprof_steps
            def kind_of(tag):
                t0_ = tag.split(" ")[0]
                return "wgrad" if t0_.startswith("wgrad") else ("dgrad" if t0_.startswith("dgrad") else "forward_and_linear_adjoints")

            fam = {}
            for row in tprof:
                a = fam.setdefault(kind_of(row[0]), [0.0, 0.0, 0.0, 0])
                a[0] += row[1]
                a[1] += row[5]
                a[2] += row[2].elapsed_time(row[3]) * 1e-3
                a[3] += 1
            ft, xt, tt = sum(a[0] for a in fam.values()), sum(a[1] for a in fam.values()), sum(a[2] for a in fam.values())
            if args.dump_launches:  # the iteration's per-launch table beside the forward's
                per, nl = {}, len(tprof) // kts
                for i, row in enumerate(tprof):
                    a = per.setdefault((i % nl, row[0]), [row[1], 0.0, row[5]])
                    a[1] += row[2].elapsed_time(row[3]) / kts
                with open(os.path.splitext(args.dump_launches)[0] + "_train_step.txt", "w") as fh:
                    for (i, tag), (f, t, x) in sorted(per.items()):
                        fh.write("%3d %-44s %9.2f GF %9.1f us %7.2f TF/s (executed %7.2f)\n" % (
                            i, tag, f / 1e9, t * 1e3, f / t / 1e9, x / t / 1e9))
            result["train_step"]["roofline"] = {
                "bound": "mfma", "achieved": round(ft / tt / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ft / tt / 1e12 / peak, 4),
                "algorithmic_gflop_per_step": round(ft / kts / 1e9, 1), "executed_gflop_per_step": round(xt / kts / 1e9, 1),
                "kernel_ms_per_step": round(tt * 1e3 / kts, 3), "launches_per_step": len(tprof) // kts,
                "families": {k: {"algorithmic_tflops": round(a[0] / a[2] / 1e12, 2), "frac_of_peak": round(a[0] / a[2] / 1e12 / peak, 4),
                                 "ms_per_step": round(a[2] * 1e3 / kts, 3), "gflop_per_step": round(a[0] / kts / 1e9, 1),
                                 "launches_per_step": a[3] // kts} for k, a in sorted(fam.items())},
                "what": "every contraction launch of the iteration (igemm_split_kernel, wgrad_split_128_kernel, the "
                        "Winograd-domain forward / data-gradient / weight-gradient launches with their transforms), each "
                        "bracketed alone on one stream; forward_and_linear_adjoints = the forward's launches plus the "
                        "GEMMs of the Linear / attention adjoints",
            }
        if rank == 0 and args.model == "DAnA" and training and not args.no_secondary:
            # the same step with the query and the support batch sharing ONE launch per trunk conv (model.merge_trunk,
            # merge_from 0): fewer, fuller launches -- a better per-launch roofline -- but a single stream of them, and the
            # two-stream default overlaps each launch's tail with the other batch's kernels. Both are reported.
            keep_merge = (model.merge_trunk, model.merge_from)
            model.merge_trunk, model.merge_from = True, 0
            for _ in range(3):
                eager_step()
            torch.cuda.synchronize()
            km = max(5, args.steps // 4)
            t0 = time.perf_counter()
            for _ in range(km):
                eager_step()
            torch.cuda.synchronize()
            dtm = time.perf_counter() - t0
            keep_steps, args.steps = args.steps, km
            profm = contraction_pass()
            args.steps = keep_steps
            model.merge_trunk, model.merge_from = keep_merge
            fm, xm, tm_ = family(profm)
            dm = [r for r in profm if not r[0].startswith("wino")]
            fdm, _xdm, tdm = family(dm)
            result["merged_trunk"] = {
                "what": "query + support batch in one launch per trunk conv (dual-geometry contraction, one batched Winograd "
                        "plane GEMM), eager",
                "value": round(args.batch * km / dtm, 3), "unit": result["unit"], "ms_per_step": round(dtm / km * 1e3, 3),
                "roofline": {"achieved": round(fm / tm_ / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                             "frac": round(fm / tm_ / 1e12 / peak, 4), "launches_per_step": len(profm) // km,
                             "kernel_ms_per_step": round(tm_ * 1e3 / km, 3),
                             "direct_frac_of_peak": round(fdm / tdm / 1e12 / peak, 4)},
            }
        if split and not args.no_secondary:
            # For reference: the same step and the same contraction pass with every contraction on the f32 MFMA.
            ops.set_mfma_mode(0)
            for _ in range(3):
                eager_step()
            torch.cuda.synchronize()
            k0 = max(5, args.steps // 4)
            t0 = time.perf_counter()
            for _ in range(k0):
                eager_step()
            torch.cuda.synchronize()
            dt0 = time.perf_counter() - t0
            keep_steps, args.steps = args.steps, k0
            prof0 = contraction_pass()
            args.steps = keep_steps
            f0, _x0, t0s = family(prof0)
            ops.set_mfma_mode(1)
            result["f32_mfma_only"] = {
                "value": round(args.batch * k0 / dt0, 3), "unit": result["unit"],
                "ms_per_step": round(dt0 / k0 * 1e3, 3),
                "roofline": {"bound": "mfma", "kernel": "igemm_f32_kernel (v_mfma_f32_32x32x2_f32 implicit GEMM)",
                             "achieved": round(f0 / t0s / 1e12, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(f0 / t0s / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                             "kernel_ms_per_step": round(t0s * 1e3 / k0, 3)},
            }
    phase("roofline passes done")
    if (rank == 0 and world == 1 and args.mode == "train" and not args.no_secondary and args.model == "DAnA"
            and not args.device_rng and not args.single_stream):
        # secondary object: the same forward with the training targets sampled by the device Philox RNG -- no host round
        # trip, so the whole forward is ONE hipGraph. Same distributions as the reference's np.random draws, another random
        # stream (which is why it is not the headline: the parity tests pin the host-RNG path). The one-graph replay has no
        # host-issue gaps and no graph-to-graph hand-over (0.6 ms in the three-graph host-RNG replay: tools/sync_gap.py).
        from dana_amd.graphs import GraphedDAnA
        model.device_rng = True
        try:
            rung = GraphedDAnA(model, *inputs)
            for _ in range(5):
                rung(*rung.inputs)
            torch.cuda.synchronize()
            kd = max(10, args.steps // 2)
            t0 = time.perf_counter()
            for _ in range(kd):
                rung(*rung.inputs)
            torch.cuda.synchronize()
            dtd = time.perf_counter() - t0
            result["device_rng_one_graph"] = {
                "what": "train-mode forward, targets sampled by the device Philox RNG (opt-in: DAnARCNN.device_rng), replayed "
                        "as one hipGraph", "value": round(args.batch * kd / dtd, 3), "unit": "query-images/sec",
                "ms_per_step": round(1e3 * dtd / kd, 3), "steps": kd}
            del rung
            if use_programs or args.launch in ("auto", "program"):
                # ... and as ONE launch program: no host round trip AND the eager step's GPU schedule -- what the sync-free
                # design of row N2 exists for (the one-graph replay loses 6-8 % on the GPU to hipGraphLaunch's scheduling)
                from dana_amd.program import ProgramDAnA
                runp = ProgramDAnA(model, *inputs)
                for _ in range(5):
                    runp(*runp.inputs)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(kd):
                    runp(*runp.inputs)
                torch.cuda.synchronize()
                dtp = time.perf_counter() - t0
                result["device_rng_one_program"] = {
                    "what": "the same device-RNG forward replayed as one launch program (program.py): no host round trip, the "
                            "eager step's own launches on its own streams", "value": round(args.batch * kd / dtp, 3),
                    "unit": "query-images/sec", "ms_per_step": round(1e3 * dtp / kd, 3), "steps": kd,
                    "host_enqueue_ms_per_step": host_enqueue_ms(lambda: runp(*runp.inputs), 20)}
                del runp
        finally:
            model.device_rng = False
    if rank == 0 and world == 1 and args.mode == "train" and args.ba and not args.no_secondary and args.model == "DAnA":
        # secondary object: BASELINE configs[1] (CISA only, use_BA_block=False) on the same episodes
        m1 = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=False, way=args.way, shot=args.shot, classes=["fg", "bg"])
        m1.load_state_dict(S.fill_state_dict(m1.state_dict(), seed=11, profile="test"))
        m1.to(dev).train()
        m1.device_rng = bool(args.device_rng)
        k1 = max(10, args.steps // 2)

        def wall(fn):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k1):
                fn()
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        with torch.no_grad():
            # eager issue and hipGraph replay both timed, the faster reported (like the headline: which of the two wins depends
            # on how the runtime maps the graph's branches onto its hardware queues)
            dt1, launch1 = wall(lambda: m1(*inputs)), "eager"
            if use_graphs:
                from dana_amd.graphs import GraphedDAnA
                run1 = GraphedDAnA(m1, *inputs)
                dtg = wall(lambda: run1(*run1.inputs))
                if dtg < dt1:
                    dt1, launch1 = dtg, "hipGraph replay"
                del run1
            if use_programs and not m1.device_rng:
                from dana_amd.program import ProgramDAnA
                run1 = ProgramDAnA(m1, *inputs)
                dtg = wall(lambda: run1(*run1.inputs))
                if dtg < dt1:
                    dt1, launch1 = dtg, "launch-program replay"
                del run1
        result["configs_1_cisa_only"] = {"value": round(args.batch * k1 / dt1, 3), "unit": "query-images/sec",
                                         "ms_per_step": round(1e3 * dt1 / k1, 3), "steps": k1, "launch": launch1}
        del m1

    def secondary_workload(label, mode, batch, height, width, shot, k, trunk=50):
        """another BASELINE configuration in the same line: its own model / episodes, eager and hipGraph replay timed
        (the faster is the value), and its own per-launch contraction roofline (single-stream pass, HIP events)."""
        train_ = mode == "train"
        way_ = args.way if train_ else 1
        m2 = build_model("DAnA", True, args.way, shot, trunk)
        sd2 = S.fill_state_dict(m2.state_dict(), seed=11, profile="test")
        m2.load_state_dict(S.tame_res101_weights(sd2) if trunk == 101 else sd2)
        m2.to(dev)
        m2.train() if train_ else m2.eval()
        in2 = [t.to(dev) for t in S.episode_inputs(batch, way_, shot, height, width, seed=1996)]
        np.random.seed(1996)
        with torch.no_grad():
            eager2 = lambda: m2(*in2)  # noqa: E731
            t_eager = trial(eager2, k)
            t_graph = t_prog = None
            if use_graphs:
                from dana_amd.graphs import GraphedDAnA
                run2 = GraphedDAnA(m2, *in2)
                t_graph = trial(lambda: run2(*run2.inputs), k)
                del run2
            if use_programs:
                from dana_amd.program import ProgramDAnA
                run2 = ProgramDAnA(m2, *in2)
                t_prog = trial(lambda: run2(*run2.inputs), k)
                del run2
            ops.PROFILE = []
            m2._single_stream = True
            torch.cuda.synchronize()
            for _ in range(k):
                eager2()
            torch.cuda.synchronize()
            m2._single_stream = False
            prof2, ops.PROFILE = ops.PROFILE, None
        f2 = sum(r[1] for r in prof2)
        x2 = sum(r[5] for r in prof2)
        t2 = sum(r[2].elapsed_time(r[3]) for r in prof2) * 1e-3
        modes = {"eager": t_eager, "hipGraph replay": t_graph, "launch-program replay": t_prog}
        launch2, best = min(((n, v) for n, v in modes.items() if v is not None), key=lambda nv: nv[1])
        pk = PEAK_SPLIT_TFLOPS if ops.get_mfma_mode() else PEAK_FP32_MFMA_TFLOPS
        del m2, in2
        torch.cuda.empty_cache()
        return {"workload": label, "value": round(batch / best * 1e3, 3), "unit": "query-images/sec",
                "ms_per_step": round(best, 3), "steps": k, "batch": batch,
                "timing": "median GPU-side interval between consecutive steps (trial()), not the wall-clock mean of the headline",
                "launch": launch2,
                "eager_ms_per_step": round(t_eager, 3), "graph_ms_per_step": None if t_graph is None else round(t_graph, 3),
                "program_ms_per_step": None if t_prog is None else round(t_prog, 3),
                "roofline": {"bound": "mfma", "achieved": round(f2 / t2 / 1e12, 2), "peak": pk, "unit": "TFLOP/s",
                             "frac": round(f2 / t2 / 1e12 / pk, 4), "launches_per_step": len(prof2) // k,
                             "algorithmic_gflop_per_step": round(f2 / k / 1e9, 1), "executed_gflop_per_step": round(x2 / k / 1e9, 1),
                             "kernel_ms_per_step": round(t2 * 1e3 / k, 3)}}

    default_shape = (args.height, args.width, args.shot, args.batch, args.support_size) == (600, 1000, 3, 4, 320)
    if (rank == 0 and world == 1 and args.mode == "train" and args.ba and not args.no_secondary and args.model == "DAnA"
            and default_shape and not args.single_stream):
        # BASELINE configs[0] on the GPU path (eval-mode forward, ONE 600x1000 query + 3 supports: the inference.py shape and
        # SURVEY 8d's parity-checked inference path) and configs[4]'s per-GPU shape (800x1333, shot 10, 2 of its 16 episodes)
        result["eval_b1"] = secondary_workload("BASELINE configs[0] on the HIP path: eval-mode forward, 1 query 600x1000 + 3 "
                                               "supports 320x320, BA+CISA", "eval", 1, 600, 1000, 3, 10)
        result["configs_4"] = secondary_workload("BASELINE configs[4] per-GPU shape: train-mode forward, 2 episodes of 800x1333 "
                                                 "queries + 20 supports 320x320 (way 2, shot 10), BA+CISA", "train", 2, 800, 1333, 10, 10)
        # configs[3] as far as it is defined: the resnet101 trunk (resnet.py:199), shot 5, ONE episode (its bs 8 puts one on
        # each of 8 GPUs); way stays 2 (dana.py:103-104 has no third class). No reference run exists (dana.py:337): throughput
        # + roofline only, the trunk oracle-checked in tests/test_gpu_model.py::test_res101_trunk_opt_in_vs_oracle
        result["configs_3"] = secondary_workload("BASELINE configs[3] as far as it is defined (NO reference run: dana.py:337 never "
                                                 "builds res101; oracle-checked trunk): train-mode forward, res101 trunk, 1 episode "
                                                 "of a 600x1000 query + 10 supports 320x320 (way 2, shot 5), BA+CISA",
                                                 "train", 1, 600, 1000, 5, 10, trunk=101)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.support_size == 320 and args.model == "DAnA":
        result["cpu_baseline"] = cpu_baseline(args, sd)  # (the oracle, like the reference, only runs 320x320 supports)
    if rank == 0:
        # compact summary as the LAST key: the driver's record keeps the top-level keys and the last 2 000 characters of
        # the line, and the scaling-relevant numbers sit in the middle of it
        def pick(key, *path):
            v = result.get(key)
            for k_ in path:
                v = v.get(k_) if isinstance(v, dict) else None
            return v

        ts, ex = result.get("train_step") or {}, (result.get("train_step") or {}).get("exchange") or {}
        summary = {
            "img_s": result["value"], "ms": result["ms_per_step"], "frac": pick("roofline", "frac"),
            "frac_of_bf16_peak": pick("roofline", "frac_of_bf16_peak"), "attainable_tflops": pick("roofline", "attainable"),
            "frac_of_attainable": pick("roofline", "frac_of_attainable"),
            "whole_step_frac_of_attainable": pick("roofline", "whole_step_frac_of_attainable"),
            "whole_step_executed_frac_of_attainable": pick("roofline", "whole_step_executed_frac_of_attainable"), "launch": result["launch"].split(":")[0].split(",")[0],
            "host_enqueue_ms": result.get("host_enqueue_ms_per_step"),
            "train_step_host_enqueue_ms": (result.get("train_step") or {}).get("host_enqueue_ms_per_step"),
            "mfma_busy": pick("roofline", "mfma_busy"), "direct_frac": pick("roofline", "families", "direct", "frac_of_peak"),
            "launches": pick("roofline", "launches_per_step"),
            "train_step_ms": ts.get("ms_per_step"), "train_step_ms_median": ts.get("ms_per_step_median"),
            "train_step_launch": ts.get("launch"), "train_step_frac": pick("train_step", "roofline", "frac"),
            "train_step_families": {k_: v_.get("frac_of_peak") for k_, v_ in (pick("train_step", "roofline", "families") or {}).items()
                                    if isinstance(v_, dict)} or None,
            "exchange": {k_: ex.get(k_) for k_ in ("ranks", "allreduce_ms_alone", "exposed_ms", "buckets", "bytes", "error") if k_ in ex} or None,
            "f32_mfma_only": {"img_s": pick("f32_mfma_only", "value"), "frac": pick("f32_mfma_only", "roofline", "frac")},
            "merged_trunk": {"img_s": pick("merged_trunk", "value"), "frac": pick("merged_trunk", "roofline", "frac")},
            "eval_b1": {"img_s": pick("eval_b1", "value"), "ms": pick("eval_b1", "ms_per_step"), "frac": pick("eval_b1", "roofline", "frac")},
            "configs_4": {"img_s": pick("configs_4", "value"), "ms": pick("configs_4", "ms_per_step"), "frac": pick("configs_4", "roofline", "frac")},
            "configs_3": {"img_s": pick("configs_3", "value"), "ms": pick("configs_3", "ms_per_step"), "frac": pick("configs_3", "roofline", "frac")},
            "configs_1": pick("configs_1_cisa_only", "value"), "device_rng_one_graph": pick("device_rng_one_graph", "value"),
            "device_rng_one_program": pick("device_rng_one_program", "value"),
            "cpu_img_s": pick("cpu_baseline", "value"),
        }
        cb = result.pop("cpu_baseline", None)
        if cb is not None:
            result["cpu_baseline"] = cb
        result["summary"] = summary
        flush_c_stdout()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if world > 1:
        dist_barrier()  # the other ranks wait for rank 0's roofline pass before tearing down
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
