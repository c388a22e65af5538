"""MI355X-native audio-visual frame-scoring hot path (features.extractors,
features.fusion, models.attention, models.av_model of the AudioVidSum
reference) behind the C-ABI of libavsum_hip.so.  Import as ``avsum_amd``."""
__version__ = "0.1.0"
