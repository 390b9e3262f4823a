"""CPU suite, part 2: the C-ABI library loads and exports every symbol include/dana_hip.h declares
(no compute without a GPU), argument errors surface as DanaError with a message, and the host-side
logic of the product package (module API, state_dict contract, config, target assignment, synthetic
generator) behaves like the reference / the oracle."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

import dana_amd
from dana_amd import _lib, ops, synthetic as S, targets as T
from dana_amd.config import cfg, cfg_from_list
from oracle import model_ref as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 35
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), "libdana_hip.so does not export %s" % name
    # and nothing torch-typed leaks into the boundary: only C scalars and raw pointers
    allowed = {"int", "long", "float", "double", "size_t", "dana_stream_t", "const float*", "float*", "const int*",
               "int*", "const unsigned char*", "unsigned long long", "void*", "const unsigned long long*",
               "unsigned long long*", "long long*", "const long long*", "void**", "const char*", "const void*"}
    for name, (ret, args) in protos.items():
        assert ret in ("int", "size_t", "const char*"), (name, ret)
        for ty, _ in args:
            assert ty in allowed, "%s: parameter type %r is not a plain C scalar / raw pointer" % (name, ty)
    assert _lib.lib().query("dana_abi_version") == 1
    # the debug / tuning switches live in their own header (include/dana_hip_debug.h), are exported too, and none of them
    # is declared by the drop-in ABI
    dbg = _lib.parse_header(_lib.DEBUG_HEADER)
    assert {"dana_set_epilogue_mode", "dana_set_sort_mode", "dana_set_igemm_trace", "dana_debug_force_tile",
            "dana_debug_stream_create_cumask"} <= set(dbg)
    assert not set(dbg) & set(protos)
    for name in dbg:
        assert hasattr(cdll, name), "libdana_hip.so does not export %s" % name


def test_argument_errors_are_reported_without_a_gpu():
    L = _lib.lib()
    with pytest.raises(_lib.DanaError, match="bad shape"):
        L.call("dana_roi_align_forward", None, None, None, 1, 0, 8, 8, 1, 1.0, 7, 7, 0, 0, 0, 0, None, None, 0, None)
    with pytest.raises(_lib.DanaError, match="multiples of 4"):
        L.call("dana_gemm_nt", 16, 16, 16, None, None, None, 8, 8, 6, 6, 6, 8, 0, 1, 0, 0, 0, 1.0, 0, None)
    with pytest.raises(_lib.DanaError, match="workspace"):
        L.call("dana_nms", 16, 100, 1, 0.7, 0, 0, 16, 100, 16, None, 0, None)
    # mask words + kept-row / folded-column words + state + the transposed diagonal / first three super-diagonal words
    assert L.query("dana_nms_workspace_bytes", 12000, 4) == 4 * 12000 * 188 * 8 + 4 * (2 * 188 * 8 + 16) + 4 * 4 * 12000 * 8
    # empty inputs are a no-op, like the reference (nms.h:17-18, ROIAlign_cuda.cu:278-281)
    L.call("dana_roi_align_forward", None, None, None, 1, 8, 8, 8, 0, 1.0, 7, 7, 0, 0, 0, 0, None, None, 0, None)


def test_launch_program_executor_reissues_recorded_calls_without_a_gpu():
    """include/dana_hip.h "launch programs" (csrc/program.hip): recorded C-ABI calls are re-issued by one C loop through
    libffi -- exercised here on entry points that need no GPU: a configuration call, the empty-input no-op of
    dana_roi_align_forward (19 arguments, a float in the middle, five null pointers) and its argument-error path"""
    import struct
    from dana_amd.program import LaunchProgram
    L = _lib.lib()
    h = ctypes.c_void_p()
    L.call("dana_program_create", ctypes.cast(ctypes.byref(h), ctypes.c_void_p))

    def add(name, *args):
        fn = L.fn[name]
        sig = "".join(LaunchProgram._SIG[t] for t in fn.argtypes)
        assert len(sig) == len(args)
        words = (ctypes.c_ulonglong * len(args))()
        for k, (c, a) in enumerate(zip(sig, args)):
            if c == "f":
                words[k] = struct.unpack("<I", struct.pack("<f", a))[0]
            elif c == "d":
                words[k] = struct.unpack("<Q", struct.pack("<d", a))[0]
            else:
                words[k] = (0 if a is None else int(a)) & 0xFFFFFFFFFFFFFFFF
        L.call("dana_program_add_call", h, ctypes.cast(fn, ctypes.c_void_p), sig.encode(), ctypes.cast(words, ctypes.c_void_p),
               len(args))

    prev = ops.get_mfma_mode()
    add("dana_set_mfma_mode", 0)
    add("dana_roi_align_forward", None, None, None, 1, 8, 8, 8, 0, 1.0, 7, 7, 0, 0, 0, 0, None, None, 0, None)  # R = 0: no-op
    add("dana_set_mfma_mode", 1)
    add("dana_roi_align_forward", None, None, None, 1, 0, 8, 8, 1, 1.0, 7, 7, 0, 0, 0, 0, None, None, 0, None)  # C = 0: refused
    add("dana_set_mfma_mode", 0)  # (never reached)
    assert L.query("dana_program_size", h) == 5
    ops.set_mfma_mode(1)
    assert L.query("dana_program_run", h, 0, 2) == 0 and ops.get_mfma_mode() == 0
    assert L.query("dana_program_run", h, 2, 5) == -1 and ops.get_mfma_mode() == 1  # stopped at the failing entry
    assert b"bad shape" in L.cdll.dana_last_error()
    with pytest.raises(_lib.DanaError, match="bad range"):
        L.call("dana_program_run", h, 3, 9)
    with pytest.raises(_lib.DanaError, match="signature character"):
        L.call("dana_program_add_call", h, ctypes.cast(L.fn["dana_set_mfma_mode"], ctypes.c_void_p), b"q",
               ctypes.cast((ctypes.c_ulonglong * 1)(0), ctypes.c_void_p), 1)
    L.call("dana_program_destroy", h)
    ops.set_mfma_mode(prev)


def test_mfma_mode_switch_roundtrip_and_errors():
    """dana_set_mfma_mode / dana_get_mfma_mode (host-side state only: no GPU needed)"""
    L = _lib.lib()
    prev = ops.get_mfma_mode()
    assert prev in (0, 1)
    assert ops.set_mfma_mode(0) == prev and ops.get_mfma_mode() == 0
    assert ops.set_mfma_mode(1) == 0 and ops.get_mfma_mode() == 1
    with pytest.raises(_lib.DanaError, match="mode must be"):
        L.call("dana_set_mfma_mode", 7)
    with pytest.raises(_lib.DanaError, match="mode must be"):
        L.call("dana_set_mfma_mode", 2)  # forced tiles are a debug switch now (dana_debug_force_tile)
    ops.force_tile(3)
    assert ops.get_mfma_mode() == 1
    ops.force_tile(0)
    # a CU mask that leaves an XCD without CUs is refused before any HIP call (bit i = CU i // 8 of XCD i % 8)
    mask, out = (ctypes.c_uint * 8)(*([0x0f0f0f0f] * 8)), ctypes.c_void_p()
    with pytest.raises(_lib.DanaError, match="every XCD needs"):
        L.call("dana_debug_stream_create_cumask", ctypes.cast(mask, ctypes.c_void_p), 8, 8,
               ctypes.cast(ctypes.byref(out), ctypes.c_void_p))
    assert ops.get_mfma_mode() == 1
    ops.set_mfma_mode(prev)


def test_ops_refuse_cpu_tensors():
    with pytest.raises(RuntimeError, match="no CPU"):
        ops.roi_align_forward(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5), 1.0, 7, 7, 0)
    with pytest.raises(RuntimeError):
        dana_amd._C.nms(torch.zeros(3, 4), torch.zeros(3), 0.5)


def test_module_api_and_state_dict_contract():
    """346 entries (344 without the BA layer), 70 trainable tensors / 37 113 489 params."""
    m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
    sd = m.state_dict()
    assert len(sd) == 346
    assert sum(k.startswith("RCNN_base.") for k in sd) == 258 and sum(k.startswith("RCNN_top.") for k in sd) == 60
    for k in ("rpn_unary_layer.weight", "rcnn_adapt_q_layer.bias", "rpn_channel_k_layer.weight",
              "RCNN_rpn.RPN_Conv.weight", "RCNN_rpn.RPN_cls_score.bias", "rcnn_transform_layer.weight",
              "output_score_layer.linear1.weight", "RCNN_bbox_pred.bias", "RCNN_base.6.5.bn3.running_var",
              "RCNN_top.0.0.downsample.0.weight"):
        assert k in sd, k
    assert tuple(sd["RCNN_rpn.RPN_Conv.weight"].shape) == (512, 2048, 3, 3)
    assert tuple(sd["RCNN_rpn.RPN_cls_score.weight"].shape) == (24, 512, 1, 1)
    assert tuple(sd["output_score_layer.linear1.weight"].shape) == (1024, 3136)
    trainable = [p for p in m.parameters() if p.requires_grad]
    assert len(trainable) == 70 and sum(p.numel() for p in trainable) == 37113489
    assert sum(p.numel() for p in m.parameters()) == 37389009
    assert "bias" in "".join(n for n, _ in m.named_parameters())  # train.py:79-85 keys off 'bias' in name
    m.train()
    assert not m.RCNN_base[4].training and m.RCNN_base[5].training  # dana.py:370-385
    assert all(not mod.training for mod in m.RCNN_base.modules() if isinstance(mod, torch.nn.BatchNorm2d))
    assert len(dana_amd.get_model("DAnA", pretrained=False, use_BA_block=False).state_dict()) == 344
    with pytest.raises(Exception):
        dana_amd.get_model("cisa")  # undefined in the reference too (utils.py:117-118)
    with pytest.raises(RuntimeError, match="HIP device"):
        m.eval()(*S.episode_inputs(1, 1, 3, 64, 64))


def test_config_overrides():
    assert cfg.ANCHOR_SCALES == [4, 8, 16, 32] and cfg.MAX_NUM_GT_BOXES == 50 and cfg.TRAIN.BATCH_SIZE == 128
    cfg_from_list(["TRAIN.RPN_POST_NMS_TOP_N", "1000"])
    assert cfg.TRAIN.RPN_POST_NMS_TOP_N == 1000
    cfg_from_list(["TRAIN.RPN_POST_NMS_TOP_N", "2000"])
    with pytest.raises(AssertionError):
        cfg_from_list(["NOT_A_KEY", "1"])


def test_synthetic_generator_is_portable_and_seeded():
    a, b = S.normal("x", (4, 5), 2.0, 1.0, seed=3), S.normal("x", (4, 5), 2.0, 1.0, seed=3)
    assert torch.equal(a, b) and not torch.equal(a, S.normal("y", (4, 5), 2.0, 1.0, seed=3))
    big = S.normal("z", (200000,), 1.0, 0.0, seed=1)
    assert abs(float(big.mean())) < 0.01 and abs(float(big.std()) - 1.0) < 0.01
    im, info, gt, nb, sup = S.episode_inputs(2, 2, 3, 64, 96)
    assert sup.shape == (2, 6, 3, 320, 320) and gt.shape == (2, 50, 5) and (gt[:, :3, 4] == 1).all()


def test_anchor_table_matches_the_oracle():
    assert np.array_equal(T.generate_anchors(scales=np.array([4, 8, 16, 32])), O.generate_anchors(scales=[4, 8, 16, 32]))


def test_torch_ops_registration_of_the_five_C_operators():
    """the reference's pybind `model._C` (vision.cpp:7-13) as PyTorch custom operators: csrc/torch_ops.cpp
    registers them with TORCH_LIBRARY over the C ABI; dana_amd._C binds to them when the shim library is built"""
    assert dana_amd._C.BINDING == "torch.ops.dana", dana_amd._C.BINDING
    for name in ("nms", "roi_align_forward", "roi_align_backward", "roi_pool_forward", "roi_pool_backward", "roi_align",
                 "roi_pool"):
        assert hasattr(torch.ops.dana, name)
    schema = str(torch.ops.dana.roi_align_forward.default._schema)
    assert "Tensor input, Tensor rois, float spatial_scale, int pooled_height, int pooled_width, int sampling_ratio" in schema
    # CPU tensors are refused with the reference's wording; empty input -> empty result without a launch (nms.h:17-18)
    with pytest.raises(RuntimeError, match="Not compiled with CPU support"):
        torch.ops.dana.roi_align_forward(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5), 1.0, 7, 7, 0)
    with pytest.raises(RuntimeError, match="Not compiled with CPU support"):
        dana_amd._C.roi_pool_forward(torch.zeros(1, 4, 8, 8), torch.zeros(1, 5), 1.0, 7, 7)
    assert dana_amd._C.nms(torch.zeros(0, 4), torch.zeros(0), 0.5).shape == (0,)


def test_C_module_installs_as_the_reference_model_C():
    """`sys.modules['model._C'] = dana_amd._C` is the whole binding a reference caller needs -- the
    roi_layers wrappers of lib/model/roi_layers/*.py import `from model import _C` and call these five names"""
    import sys
    import types
    stub = types.ModuleType("model")
    stub._C = dana_amd._C
    saved = {k: sys.modules.get(k) for k in ("model", "model._C")}
    sys.modules["model"], sys.modules["model._C"] = stub, dana_amd._C
    try:
        ns = {}
        exec("from model import _C\nnames = [_C.nms, _C.roi_align_forward, _C.roi_align_backward, _C.roi_pool_forward, "
             "_C.roi_pool_backward]", ns)
        assert all(callable(f) for f in ns["names"])
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_bench_rejects_a_launcher_whose_world_size_is_not_gpus():
    """bench.py's launch contract: under an external launcher WORLD_SIZE must equal --gpus; a mismatch
    exits non-zero before any device work (it would otherwise print a line whose n_gpus is not what was asked for)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert pr.returncode != 0 and b"WORLD_SIZE=2" in pr.stderr and not pr.stdout.strip()
