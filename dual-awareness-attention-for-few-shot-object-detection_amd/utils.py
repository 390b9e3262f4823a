"""The model factory of the reference's top-level utils.py:108-127 (`get_model`)."""
from .dana import DAnARCNN


def get_model(name, pretrained=True, use_BA_block=True, way=2, shot=3, classes=[]):
    """utils.py:108-127. Only the DAnA entry is in this build's scope; the sibling
    baselines (frcnn/fsod/meta/fgn) are listed as 'next' row N4."""
    if name == "DAnA":
        model = DAnARCNN(classes, "concat", 256, 256, pretrained=pretrained, semantic_enhance=use_BA_block,
                         num_way=way, num_shot=shot)
    else:
        raise Exception("network '%s' is not defined in this build (DAnA only)" % name)
    model.create_architecture()
    return model
