"""Alias of avsum_amd.utils.shot_metrics (reference import path `utils.shot_metrics`)."""
from avsum_amd.utils.shot_metrics import *  # noqa: F401,F403
from avsum_amd.utils import shot_metrics as _real

globals().update({k: v for k, v in vars(_real).items() if not k.startswith('__')})
