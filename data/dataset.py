"""Alias of avsum_amd.data.dataset (reference import path `data.dataset`)."""
from avsum_amd.data.dataset import *  # noqa: F401,F403
from avsum_amd.data import dataset as _real

globals().update({k: v for k, v in vars(_real).items() if not k.startswith('__')})
