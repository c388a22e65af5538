"""Importable alias of the package directory
``audiovidsum-a-multi-modal-approach-to-video-summarization_amd`` (whose name is
not a Python identifier).  ``import avsum_amd.ops`` resolves inside that directory.
"""
import os as _os

_REAL = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
    "audiovidsum-a-multi-modal-approach-to-video-summarization_amd",
)
__path__ = [_REAL]
exec(compile(open(_os.path.join(_REAL, "__init__.py")).read(), _os.path.join(_REAL, "__init__.py"), "exec"))
