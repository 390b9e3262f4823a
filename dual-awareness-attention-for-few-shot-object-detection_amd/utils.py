"""The model factory of the reference's top-level utils.py:108-127 (`get_model`)."""
from .dana import DAnARCNN
from .frcnn import FasterRCNN, MetaRCNN
from .fgn import FGN
from .fsod import FSOD


def get_model(name, pretrained=True, use_BA_block=True, way=2, shot=3, classes=[]):
    """utils.py:108-127. 'DAnA' is the hot path of this build; 'frcnn', 'fsod', 'meta' and 'fgn' are the
    siblings on the same operators (row N4); 'cisa' is undefined in the reference itself
    (utils.py:117-118 names a class that does not exist)."""
    if name == "DAnA":
        model = DAnARCNN(classes, "concat", 256, 256, pretrained=pretrained, semantic_enhance=use_BA_block,
                         num_way=way, num_shot=shot)
    elif name == "frcnn":
        model = FasterRCNN(classes, pretrained=pretrained)
    elif name == "fsod":
        model = FSOD(classes, pretrained=pretrained, num_way=way, num_shot=shot)
    elif name == "fgn":
        model = FGN(classes, pretrained=pretrained, num_way=way, num_shot=shot)
    elif name == "meta":
        model = MetaRCNN(classes, pretrained=pretrained, num_way=way, num_shot=shot)
    else:
        raise Exception("network '%s' is not defined in this build (DAnA, frcnn, fsod, meta, fgn)" % name)
    model.create_architecture()
    return model
