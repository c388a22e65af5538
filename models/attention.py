"""Alias of avsum_amd.models.attention (reference import path `models.attention`)."""
from avsum_amd.models.attention import *  # noqa: F401,F403
from avsum_amd.models import attention as _real

globals().update({k: v for k, v in vars(_real).items() if not k.startswith('__')})
