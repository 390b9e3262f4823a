"""Import the reference (/root/reference) in THIS build container to pin the oracle and to emit golden
vectors. Nothing here runs on the GPU box and nothing of the reference is copied into the repo:
the stubs and the scratch build of the reference's CPU `_C` live under /tmp.

Recipe: stub easydict / torchvision / cv2 (imported, unused on the path);
build lib/model/csrc/{vision.cpp,cpu/*.cpp} from a scratch copy with the two-token
`.type()` -> `.scalar_type()` edit torch>=1.11 needs (cpu/ROIAlign_cpu.cpp:242, cpu/nms_cpu.cpp:71).
"""
import os
import shutil
import sys
import types

REF = "/root/reference"
SCRATCH = "/tmp/dana_ref_scratch"


def available():
    return os.path.isdir(os.path.join(REF, "lib", "model"))


def _write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text)


def _stubs():
    d = os.path.join(SCRATCH, "stubs")
    _write(os.path.join(d, "easydict", "__init__.py"), '''
class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v
    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        super().__setitem__(k, v)
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)
    def __setattr__(self, k, v):
        self[k] = v
''')
    _write(os.path.join(d, "torchvision", "__init__.py"), "")
    _write(os.path.join(d, "torchvision", "models.py"), "")
    _write(os.path.join(d, "torchvision", "utils.py"), "")
    _write(os.path.join(d, "cv2", "__init__.py"), "")
    return d


def _build_ref_C():
    import torch.utils.cpp_extension as ext
    src = os.path.join(SCRATCH, "csrc")
    if not os.path.isdir(src):
        shutil.copytree(os.path.join(REF, "lib", "model", "csrc"), src)
        for rel in ("cpu/ROIAlign_cpu.cpp", "cpu/nms_cpu.cpp"):
            p = os.path.join(src, rel)
            t = open(p).read()
            t = t.replace("AT_DISPATCH_FLOATING_TYPES(input.type(),", "AT_DISPATCH_FLOATING_TYPES(input.scalar_type(),")
            t = t.replace("AT_DISPATCH_FLOATING_TYPES(dets.type(),", "AT_DISPATCH_FLOATING_TYPES(dets.scalar_type(),")
            open(p, "w").write(t)
    bdir = os.path.join(SCRATCH, "build")
    os.makedirs(bdir, exist_ok=True)
    return ext.load("dana_ref_C", [os.path.join(src, "vision.cpp"), os.path.join(src, "cpu", "nms_cpu.cpp"),
                                   os.path.join(src, "cpu", "ROIAlign_cpu.cpp")],
                    extra_include_paths=[src], build_directory=bdir, with_cuda=False, verbose=False)


_state = {}


def load():
    """-> dict(cfg=..., C=<reference CPU _C>, dana=<module lib/model/framework/dana.py>, ...)"""
    if _state:
        return _state
    sys.dont_write_bytecode = True
    sys.path[:0] = [_stubs(), os.path.join(REF, "lib")]
    C = _build_ref_C()
    import model  # the reference's lib/model package
    sys.modules["model._C"] = C
    model._C = C
    from model.utils.config import cfg, cfg_from_file, cfg_from_list
    cfg_from_file(os.path.join(REF, "cfgs", "res50.yml"))
    cfg_from_list(["ANCHOR_SCALES", "[4, 8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "50"])
    from model.framework import dana
    from model.rpn import generate_anchors, bbox_transform, proposal_layer
    _state.update(cfg=cfg, C=C, dana=dana, generate_anchors=generate_anchors, bbox_transform=bbox_transform,
                  proposal_layer=proposal_layer)
    return _state


def build_model(use_ba, way, shot, attention_type="concat"):
    r = load()
    m = r["dana"].DAnARCNN(["fg", "bg"], attention_type, 256, 256, pretrained=False, semantic_enhance=use_ba,
                           num_way=way, num_shot=shot)
    m.create_architecture()
    return m


def build_frcnn():
    """the reference's plain Faster R-CNN sibling (lib/model/framework/faster_rcnn.py:122-203), class-agnostic, 2 classes"""
    load()
    from model.framework import faster_rcnn
    m = faster_rcnn.FasterRCNN(["fg", "bg"], pretrained=False)
    m.create_architecture()
    return m


def build_meta(way, shot):
    """the reference's Meta R-CNN sibling (lib/model/framework/meta.py:173-251)"""
    load()
    from model.framework import meta
    m = meta.METARCNN(["fg", "bg"], pretrained=False, num_way=way, num_shot=shot)
    m.create_architecture()
    return m


def build_fsod(way, shot):
    """the reference's FSOD sibling (lib/model/framework/fsod.py:252-327): attention RPN + multi-relation head"""
    load()
    from model.framework import fsod
    m = fsod.FSOD(["fg", "bg"], pretrained=False, num_way=way, num_shot=shot)
    m.create_architecture()
    return m


def build_fgn(way, shot):
    """the reference's FGN sibling (lib/model/framework/fgn.py:190-259)"""
    load()
    from model.framework import fgn
    m = fgn.FGN(["fg", "bg"], pretrained=False, num_way=way, num_shot=shot)
    m.create_architecture()
    return m
