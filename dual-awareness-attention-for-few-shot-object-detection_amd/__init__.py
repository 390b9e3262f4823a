"""MI355X-native DAnA forward path (hand-written gfx950 kernels behind the reference's module API).

Importable as ``dana_amd`` (the directory name carries hyphens; ``dana_amd/__init__.py`` at the
repository root points its ``__path__`` here)."""
from .config import cfg, cfg_from_file, cfg_from_list  # noqa: F401
from .utils import get_model  # noqa: F401
from .dana import DAnARCNN  # noqa: F401
from . import ops, roi_layers, _C, postprocess, graphs  # noqa: F401
