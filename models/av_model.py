"""Alias of avsum_amd.models.av_model (reference import path `models.av_model`)."""
from avsum_amd.models.av_model import *  # noqa: F401,F403
from avsum_amd.models import av_model as _real

globals().update({k: v for k, v in vars(_real).items() if not k.startswith('__')})
