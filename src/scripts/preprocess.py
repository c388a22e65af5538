"""Alias of avsum_amd.scripts.preprocess (reference import path `src.scripts.preprocess`); nothing runs on import."""
from avsum_amd.scripts.preprocess import *  # noqa: F401,F403
from avsum_amd.scripts import preprocess as _real

globals().update({k: v for k, v in vars(_real).items() if not k.startswith('__')})
