"""Alias of avsum_amd.features.fusion (reference import path `src.features.fusion`)."""
from avsum_amd.features.fusion import *  # noqa: F401,F403
from avsum_amd.features import fusion as _real

globals().update({k: v for k, v in vars(_real).items() if not k.startswith('__')})
