"""Alias of avsum_amd.evaluation.metrics (reference import path `src.evaluation.metrics`)."""
from avsum_amd.evaluation.metrics import *  # noqa: F401,F403
from avsum_amd.evaluation import metrics as _real

globals().update({k: v for k, v in vars(_real).items() if not k.startswith('__')})
