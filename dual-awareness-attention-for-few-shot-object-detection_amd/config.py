"""Global mutable configuration, mirroring lib/model/utils/config.py (`cfg`, cfg_from_file,
cfg_from_list) with the defaults the DAnA path reads (config.py:19-303) already merged with
cfgs/res50.yml and the CLI override of utils.py:70-71 (ANCHOR_SCALES [4,8,16,32], 50 gt boxes).
Read at call time, like the reference (e.g. proposal_layer.py:72-75), so drivers may mutate it."""
import ast


class AttrDict(dict):
    """easydict-style attribute dictionary (nested dicts become AttrDicts)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


cfg = AttrDict(
    TRAIN=dict(
        LEARNING_RATE=0.001, MOMENTUM=0.9, WEIGHT_DECAY=0.0001, GAMMA=0.1, DOUBLE_BIAS=False, TRUNCATED=False,
        BIAS_DECAY=False, BATCH_SIZE=128, FG_FRACTION=0.25, FG_THRESH=0.5, BG_THRESH_HI=0.5, BG_THRESH_LO=0.0,
        BBOX_NORMALIZE_TARGETS_PRECOMPUTED=True, BBOX_NORMALIZE_MEANS=(0.0, 0.0, 0.0, 0.0),
        BBOX_NORMALIZE_STDS=(0.1, 0.1, 0.2, 0.2), BBOX_INSIDE_WEIGHTS=(1.0, 1.0, 1.0, 1.0),
        RPN_POSITIVE_OVERLAP=0.7, RPN_NEGATIVE_OVERLAP=0.3, RPN_CLOBBER_POSITIVES=False, RPN_FG_FRACTION=0.5,
        RPN_BATCHSIZE=256, RPN_NMS_THRESH=0.7, RPN_PRE_NMS_TOP_N=12000, RPN_POST_NMS_TOP_N=2000, RPN_MIN_SIZE=8,
        RPN_BBOX_INSIDE_WEIGHTS=(1.0, 1.0, 1.0, 1.0), RPN_POSITIVE_WEIGHT=-1.0, HAS_RPN=True,
    ),
    TEST=dict(NMS=0.3, RPN_NMS_THRESH=0.7, RPN_PRE_NMS_TOP_N=6000, RPN_POST_NMS_TOP_N=300, RPN_MIN_SIZE=16,
              HAS_RPN=True, BBOX_REG=True),
    RESNET=dict(MAX_POOL=False, FIXED_BLOCKS=1),
    POOLING_MODE="align",
    POOLING_SIZE=7,
    MAX_NUM_GT_BOXES=50,
    ANCHOR_SCALES=[4, 8, 16, 32],
    ANCHOR_RATIOS=[0.5, 1, 2],
    FEAT_STRIDE=[16],
    CUDA=False,
    CROP_RESIZE_WITH_MAX_POOL=False,
    EXP_DIR="res50",
)


def _merge(a, b):
    for k, v in a.items():
        if k not in b:
            raise KeyError("{} is not a valid config key".format(k))
        if isinstance(v, dict):
            _merge(v, b[k])
        else:
            b[k] = v


def cfg_from_file(filename):
    """config.py:371: merge a YAML file into cfg."""
    import yaml
    with open(filename, "r") as f:
        _merge(AttrDict(yaml.safe_load(f)), cfg)


def cfg_from_list(cfg_list):
    """config.py:380: ['A.B', 'value', ...] overrides."""
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        keys = k.split(".")
        d = cfg
        for sub in keys[:-1]:
            assert sub in d
            d = d[sub]
        assert keys[-1] in d
        try:
            value = ast.literal_eval(v)
        except Exception:
            value = v
        d[keys[-1]] = value
