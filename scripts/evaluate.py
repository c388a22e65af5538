"""Alias of avsum_amd.scripts.evaluate (reference import path `scripts.evaluate`)."""
from avsum_amd.scripts.evaluate import *  # noqa: F401,F403
from avsum_amd.scripts import evaluate as _real

globals().update({k: v for k, v in vars(_real).items() if not k.startswith('__')})
