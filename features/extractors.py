"""Alias of avsum_amd.features.extractors (reference import path `features.extractors`)."""
from avsum_amd.features.extractors import *  # noqa: F401,F403
from avsum_amd.features import extractors as _real

globals().update({k: v for k, v in vars(_real).items() if not k.startswith('__')})
