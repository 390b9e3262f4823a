"""Import alias: ``import dana_amd`` -> the package in
``dual-awareness-attention-for-few-shot-object-detection_amd/`` (a directory name Python cannot import)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "dual-awareness-attention-for-few-shot-object-detection_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
