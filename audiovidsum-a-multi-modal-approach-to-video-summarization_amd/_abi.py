"""ctypes binding of libavsum_hip.so (the C-ABI declared in include/avsum_hip.h).

The product path has no CPU fallback: if the library is missing, or a call
returns a non-zero status, this module raises.  Build with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C csrc``.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint, c_void_p

# torch first: its HIP runtime (torch/lib/libamdhip64) must be the one this process initialises.  Loading
# libavsum_hip.so before torch pulls in /opt/rocm's copy of the same SONAME instead, and the second runtime to come
# up then reports "no ROCm-capable device".
import torch  # noqa: F401  (memory, streams and the HIP runtime come from here)

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# AVS_STUDY_LIB=1 loads the kernel-study build (`make study`: the same library with the ablation switches of tools/
# compiled in, plus avs_debug_flags); the product never sets it
# AVS_STUDY_LIB=fp16emu loads the accuracy-study flavour (`make fp16emu`: AVS_F16X2 with the lo halves forced to zero =
# the arithmetic of a plain fp16-storage mode); tools/fp16_storage_study.py only
STUDY = os.environ.get("AVS_STUDY_LIB") == "1"
_FLAVOUR = os.environ.get("AVS_STUDY_LIB")
LIB_PATH = os.path.join(_PKG_DIR, "lib", "libavsum_hip_study.so" if STUDY else
                        "libavsum_hip_fp16emu.so" if _FLAVOUR == "fp16emu" else "libavsum_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG_DIR), "include", "avsum_hip.h")

AVS_F32, AVS_BF16, AVS_F32_ACC64, AVS_F32_SPLIT, AVS_F16X2 = 0, 1, 2, 3, 4
AVS_W_ROWS, AVS_W_KSTEP32 = 0, 1
TILE_AUTO, TILE_128, TILE_256, TILE_224, STAGING_GENERIC = 0, 1, 2, 3, 4      # avs_conv_desc.variant
X_F16P8, Y_F16P8, RES_F16P8 = 1, 2, 4                                         # avs_conv_desc.formats
LSTM_AUTO, LSTM_STREAM, LSTM_RESIDENT_20_8, LSTM_RESIDENT_16_8 = 0, 1, 2, 3
LSTM_SPLIT4 = 4   # (host-side choice: the avs_lstm*_split_f32 entry points - one recurrence over four CUs)
ACT_NONE, ACT_RELU = 0, 1
BIAS_NONE, BIAS_COL, BIAS_ROW = 0, 1, 2
E_UNSUPPORTED = -6


class AvsError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("dtype", c_int),
        ("n", c_int), ("h", c_int), ("w", c_int),
        ("cin", c_int), ("kh", c_int), ("kw", c_int),
        ("sh", c_int), ("sw", c_int), ("ph", c_int), ("pw", c_int),
        ("ho", c_int), ("wo", c_int), ("cout", c_int),
        ("x_img_stride", c_int64), ("x_row_stride", c_int64), ("x_px_stride", c_int64),
        ("w_row_stride", c_int64), ("y_px_stride", c_int64),
        ("act", c_int), ("alpha", c_float), ("w_layout", c_int), ("variant", c_int), ("formats", c_int),
    ]


P = c_void_p  # device pointer
_SIGNATURES = {
    "avs_abi_version": (c_int, []),
    "avs_last_error": (c_char_p, []),
    "avs_device_info": (c_int, [c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int64), c_char_p, c_int]),
    "avs_f16x2_pack_f32": (c_int, [P, P, c_int64, P]),
    "avs_f16x2_unpack_f32": (c_int, [P, P, c_int64, P]),
    "avs_f16p8_pack_f32": (c_int, [P, P, c_int64, P]),
    "avs_f16p8_unpack_f32": (c_int, [P, P, c_int64, P]),
    "avs_conv2d_nhwc": (c_int, [POINTER(ConvDesc), P, P, P, P, P]),
    "avs_conv2d_bnstats_workspace_bytes": (c_int64, [POINTER(ConvDesc), c_int64]),
    "avs_conv2d_nhwc_bnstats": (c_int, [POINTER(ConvDesc), P, P, P, c_int64, P, P, c_float, P, P, P, c_int64, P]),
    "avs_conv1x1_bn_bf16": (c_int, [P, c_int64, c_int, P, c_int64, c_int, c_int64, c_int, P, P, c_float, P, c_int64,
                                    c_int, P, c_int64, P]),
    "avs_conv1x1_bn_in_bf16": (c_int, [P, c_int64, c_int, P, P, P, c_int64, c_int, c_int64, c_int, P, P, c_float, P,
                                       c_int64, c_int, P, c_int64, P]),
    "avs_stem_workspace_bytes": (c_int64, [c_int]),
    "avs_stem_conv_bn_pool_bf16": (c_int, [P, c_int, c_float, POINTER(c_float), POINTER(c_float), P, c_int64, c_int, P, P,
                                           c_float, c_int, c_int, P, P, P, P, c_int64, P]),
    "avs_stem_f16x2_workspace_bytes": (c_int64, [c_int]),
    "avs_stem_conv_pool_f16x2": (c_int, [P, c_int, P, c_int64, c_int, P, P, c_float, P, P, P, P, c_int64, P]),
    "avs_bn_gram_affine_bf16": (c_int, [P, c_int64, c_int, P, P, P, c_int64, c_int, c_int64, c_int, P, P, c_float, P, P,
                                        P, c_int64, P]),
    "avs_conv1x1_affine_bf16": (c_int, [P, c_int64, c_int, P, P, P, c_int64, c_int, c_int64, c_int, P, P, P, c_int64,
                                        P, P, c_int, P, c_int64, P]),
    "avs_bn_gram_affine_f16x2": (c_int, [P, c_int64, c_int, P, P, P, c_int64, c_int, c_int64, c_int, P, P, c_float, P, P,
                                         P, c_int64, P]),
    "avs_conv2d_nhwc_affine": (c_int, [POINTER(ConvDesc), P, P, P, c_int64, P, P, P, c_int64, P, P, P]),
    "avs_conv2d_nhwc_split": (c_int, [POINTER(ConvDesc), P, P, P, P, c_int, P, c_int64, c_int, P]),
    "avs_conv2d_bnlocal_tile_rows": (c_int, [POINTER(ConvDesc), c_int64]),
    "avs_conv2d_nhwc_bnlocal": (c_int, [POINTER(ConvDesc), P, P, P, c_int64, P, P, c_float, P, c_int64, P]),
    "avs_conv2d_bncluster_workspace_bytes": (c_int64, [POINTER(ConvDesc), c_int64, c_int]),
    "avs_conv2d_nhwc_bncluster": (c_int, [POINTER(ConvDesc), P, P, P, c_int64, c_int, P, P, c_float, P, c_int64, P, c_int64,
                                          c_uint, P]),
    "avs_gemm_nt": (c_int, [c_int, c_int, c_int, c_int, P, c_int64, c_int64, P, c_int64, c_int64, P, c_int64,
                            c_int64, P, c_int, c_int64, c_float, c_int, c_int, P]),
    "avs_pull_copy_u8": (c_int, [P, P, c_int64, c_int, P]),
    "avs_frames_normalize_u8": (c_int, [c_int, P, c_int, c_int, c_int, c_float, POINTER(c_float), POINTER(c_float),
                                        POINTER(c_float), P, c_int, c_int, c_int, c_int, P]),
    "avs_resize_bilinear_u8": (c_int, [P, c_int, c_int, c_int, P, c_int, c_int, P]),
    "avs_bn_batch_stats": (c_int, [c_int, P, c_int64, c_int, c_int64, P, c_int, P, P, c_float, P, P, P]),
    "avs_bn_apply": (c_int, [c_int, P, c_int64, c_int, c_int64, P, c_int, c_int64, P, P, P, c_int64, c_int, P,
                             c_int64, P]),
    "avs_pool2d_nhwc": (c_int, [c_int, c_int, P, c_int, c_int, c_int, c_int, c_int64, c_int, c_int, c_int, P, c_int, P,
                                c_int, c_int, c_int64, P]),
    "avs_bn_maxpool_nhwc": (c_int, [c_int, P, c_int, c_int, c_int, c_int, c_int64, P, c_int, P, P, c_int, c_int, c_int,
                                    c_int, P, c_int, c_int, c_int64, P]),
    "avs_global_avgpool_nhwc": (c_int, [c_int, P, c_int, c_int, c_int, P, c_int64, P]),
    "avs_segment_mean_f32": (c_int, [P, c_int64, c_int, P, c_int, P, c_int64, P]),
    "avs_hsv_frame_diff_u8": (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    "avs_reflect_pad_f32": (c_int, [P, c_int64, c_int, P, c_int64, P]),
    "avs_stft_f64": (c_int, [P, c_int64, c_int64, c_int, c_int, P, c_int, c_int, P, P]),
    "avs_stft_mel_fused_f32": (c_int, [P, c_int64, P, P, P, P, P, P, c_int, P, P, P, P, P]),
    "avs_stft_mel_segmean_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "avs_stft_mel_segmean_f32": (c_int, [P, c_int64, P, P, P, P, P, P, c_int, P, c_int, P, P, c_int, P, c_int, c_float, P,
                                         c_int64, P, c_int64, P, c_int64, P]),
    "avs_stft_mel_segmean_batch_f32": (c_int, [P, P, P, c_int, P, P, P, P, P, P, c_int, P, c_int, P, P, c_int, P, c_float, P,
                                               c_int64, P, c_int64, P, c_int64, P]),
    "avs_power_mel_f32": (c_int, [P, c_int64, c_int, P, P, P, c_int, c_int, P, P, P]),
    "avs_clamp_topdb_f32": (c_int, [P, c_int64, P, c_float, P]),
    "avs_fill_f32": (c_int, [P, c_int64, c_float, P]),
    "avs_quantize_f32": (c_int, [P, c_int64, c_float, c_float, c_float, P, P]),
    "avs_resample_f32": (c_int, [P, c_int64, c_int, P, c_int, c_int, c_int, c_int, P, c_int64, P]),
    "avs_lstm_f32": (c_int, [P, P, c_int, c_int, c_uint, P, c_int, P, c_int64, c_int, c_int, P]),
    "avs_mha_batchaxis_f32": (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    "avs_score_head_f32": (c_int, [P, c_int64, c_int, c_int64, P, P, P, P]),
    "avs_mhsa_flash_f32": (c_int, [P, P, P, c_int64, c_int, c_int, c_int, c_int, P, c_int64, P]),
    "avs_mhsa_flash_f16x2": (c_int, [P, P, P, c_int64, c_int, c_int, c_int, c_int, P, c_int64, P]),
    "avs_softmax_rows_f32": (c_int, [P, c_int64, c_int, c_int64, P]),
    "avs_softmax_bwd_rows_f32": (c_int, [P, P, c_int64, c_int, c_int64, c_float, P]),
    "avs_transpose_f32": (c_int, [P, c_int, c_int, c_int64, P, c_int64, P]),
    "avs_colsum_f32": (c_int, [P, c_int64, c_int, c_int64, P, P, P]),
    "avs_relu_dropout_bwd_f32": (c_int, [P, P, P, c_int64, P, P]),
    "avs_mul_f32": (c_int, [P, P, c_int64, P, P]),
    "avs_score_head_bwd_f32": (c_int, [P, P, P, c_int64, c_int, c_int64, P, P, P, P]),
    "avs_lstm_train_fwd_f32": (c_int, [P, P, c_int, c_int, c_uint, P, c_int, P, c_int64, c_int, P, P, c_int, P]),
    "avs_lstm_bwd_f32": (c_int, [P, c_int64, c_int, P, P, P, c_int, c_int, c_uint, P, c_int, P, c_int, P]),
    "avs_lstm_split_workspace_bytes": (c_size_t, [c_int, c_int]),
    "avs_lstm_split_f32": (c_int, [P, P, c_int, c_int, c_uint, P, c_int, P, c_int64, c_int, P, P, P, c_size_t, c_uint, P]),
    "avs_lstm_bwd_split_f32": (c_int, [P, c_int64, c_int, P, P, P, c_int, c_int, c_uint, P, c_int, P, P, c_size_t, c_uint, P]),
    "avs_cdist_f64": (c_int, [P, c_int, P, c_int, c_int, P, P]),
    "avs_dtw_workspace_bytes": (c_int64, [c_int, c_int]),
    "avs_dtw_path_f64": (c_int, [P, c_int, c_int, P, c_int64, P, P, P, P]),
    "avs_gather_scale_f32": (c_int, [P, c_int64, c_int, P, P, c_int, P, P]),
}

_lib = None


def declared_symbols():
    """Names of every function include/avsum_hip.h declares (parsed from the header)."""
    import re
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(avs_[a-z0-9_]+)\s*\(", text)))


def lib():
    """Load libavsum_hip.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AvsError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run __graft_entry__.build() or `make -C <package>/csrc`). There is no CPU fallback."
        )
    handle = ctypes.CDLL(LIB_PATH)
    if STUDY:   # the kernel-study build's ablation switches and rule setters (tools/ only)
        handle.avs_debug_flags.restype = None
        handle.avs_debug_flags.argtypes = [c_int]
        for name, args in (("avs_tune_short_reduction_bytes", [c_int]), ("avs_tune_tall_rule", [c_int, c_int]),
                           ("avs_tune_pipeline", [c_int]), ("avs_tune_bnlocal", [c_int]),
                           ("avs_tune_convbn_narrow", [c_int])):
            getattr(handle, name).restype = None
            getattr(handle, name).argtypes = args
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    if handle.avs_abi_version() != 5:
        raise AvsError(f"ABI version mismatch: library reports {handle.avs_abi_version()}, binding expects 5")
    if STUDY and os.environ.get("AVS_TUNE_CONVBN_NARROW") is not None:
        handle.avs_tune_convbn_narrow(int(os.environ["AVS_TUNE_CONVBN_NARROW"]))
    if STUDY and os.environ.get("AVS_TUNE_PIPELINE") is not None:  # kernel-study override of the library default
        handle.avs_tune_pipeline(int(os.environ["AVS_TUNE_PIPELINE"]))
    _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().avs_last_error()
        raise AvsError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")


def device_info(dev=0):
    cu, clk, mem = c_int(), c_int(), c_int64()
    arch = ctypes.create_string_buffer(64)
    check(lib().avs_device_info(dev, ctypes.byref(cu), ctypes.byref(clk), ctypes.byref(mem), arch, 64),
          "avs_device_info")
    return {"cu_count": cu.value, "clock_khz": clk.value, "hbm_bytes": mem.value, "arch": arch.value.decode()}
