"""Alias of avsum_amd.utils.alignments (reference import path `utils.alignments`)."""
from avsum_amd.utils.alignments import *  # noqa: F401,F403
from avsum_amd.utils import alignments as _real

globals().update({k: v for k, v in vars(_real).items() if not k.startswith('__')})
