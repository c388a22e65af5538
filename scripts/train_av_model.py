"""Alias of avsum_amd.scripts.train_av_model (reference import path `scripts.train_av_model`)."""
from avsum_amd.scripts.train_av_model import *  # noqa: F401,F403
from avsum_amd.scripts import train_av_model as _real

globals().update({k: v for k, v in vars(_real).items() if not k.startswith('__')})
