/*
 * avsum_hip.h — C-ABI of libavsum_hip.so, the MI355X (gfx950) hot path of the
 * audio-visual frame-scoring pipeline.
 *
 * The reference (AudioVidSum) has no FFI layer: its boundary is the Python
 * class/function API of features/extractors.py, features/fusion.py,
 * models/attention.py and models/av_model.py.  This header is the thin C-ABI
 * the Python host mirror of those classes calls (ctypes).  Every entry point
 * names the reference lines it replaces.
 *
 * Conventions (all entry points):
 *   - `extern "C"`, plain pointers and sizes; no torch / C++ types.
 *   - returns int status: AVS_OK (0) or a negative AVS_E_* code; never throws.
 *     avs_last_error() returns a thread-local message for the last failure.
 *   - every pointer named d_* is DEVICE memory owned by the caller (the host
 *     side passes torch tensor data_ptr()s); nothing is allocated or freed.
 *   - asynchronous on the caller's `stream` (a hipStream_t cast to void*;
 *     NULL = the null stream); no host synchronisation inside.
 *   - tensors are dense row-major unless a stride is given; "NHWC" activations
 *     are [image][row][pixel][channel].
 *   - dtype: AVS_F32 = IEEE fp32 operands, fp32 accumulate on the f32 MFMA
 *     (exact fmaf chain; the parity mode).  AVS_BF16 = bf16 operands, fp32
 *     accumulate on the bf16 MFMA (the throughput mode).  AVS_F32_ACC64
 *     (avs_gemm_nt / avs_conv2d_nhwc only) = fp32 operands and fp32 MFMA over
 *     each 16-element slice of the reduction, slices summed in fp64: used for
 *     the STFT, where a long fp32 running sum would lose the quiet bins.
 *     AVS_F32_SPLIT (avs_gemm_nt / avs_conv2d_nhwc[_bnstats] only) = fp32 operands
 *     in memory, products on the bf16 MFMA: every operand element is split into
 *     hi = bf16(x), lo = bf16(x - hi) in registers and a*b is taken as
 *     ah*bh + ah*bl + al*bh (fp32 accumulate): ~2^-15 relative per product instead
 *     of exact, 5.3x the matrix rate of the f32 MFMA.
 */
#ifndef AVSUM_HIP_H
#define AVSUM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVS_ABI_VERSION 5

enum {
  AVS_OK = 0,
  AVS_E_ARG = -1,       /* null pointer / bad enum / negative size            */
  AVS_E_SHAPE = -2,     /* sizes inconsistent with what the kernel assumes    */
  AVS_E_ALIGN = -3,     /* pointer or stride not 16-byte aligned              */
  AVS_E_HIP = -4,       /* the HIP runtime reported an error at launch        */
  AVS_E_WORKSPACE = -5, /* workspace too small                                */
  AVS_E_UNSUPPORTED = -6 /* this entry point does not take the shape; use the documented alternative */
};

enum { AVS_F32 = 0, AVS_BF16 = 1, AVS_F32_ACC64 = 2, AVS_F32_SPLIT = 3, AVS_F16X2 = 4 };
/* AVS_F16X2 ("split-half" storage, the fast parity-grade arithmetic of the CNN trunk): a value is TWO fp16 numbers,
 * x ~ hi + lo with hi = fp16(x), lo = fp16(x - hi) - 22 significant bits, an absolute floor of 2^-25 (fp16
 * denormals, which the matrix cores keep), magnitudes saturating at 65504.  A tensor has the strides and the byte
 * size of the fp32 tensor of the same shape ("slots" of 4 bytes, all strides below in slots); every aligned run of
 * 8 slots (32 bytes) holds the 8 hi halves, then the 8 lo halves, of 8 consecutive elements of the innermost
 * (channel / reduction) axis - so channel counts, strides and offsets are multiples of 8 slots and pointers 32-byte
 * aligned.  The contraction kernel feeds the hi and lo runs to v_mfma_f32_32x32x16_f16 as they are: a product is
 * hi*hi + hi*lo + lo*hi (error ~2^-21 relative, fp32 accumulation) with NO arithmetic on the operands in the loop;
 * each output is split once, by the kernel that produces it.  Taken by avs_conv2d_nhwc, avs_conv2d_nhwc_bnstats,
 * avs_conv2d_nhwc_bnlocal (no bias / alpha), avs_bn_batch_stats, avs_bn_apply, avs_bn_maxpool_nhwc, avs_pool2d_nhwc,
 * avs_global_avgpool_nhwc (fp32 out), avs_frames_normalize_u8; avs_f16x2_pack_f32 / _unpack_f32 convert.         */
enum { AVS_ACT_NONE = 0, AVS_ACT_RELU = 1 };
enum { AVS_BIAS_NONE = 0, AVS_BIAS_COL = 1, AVS_BIAS_ROW = 2 };

typedef void* avs_stream_t;

/* ---- library ------------------------------------------------------------ */
int avs_abi_version(void);
const char* avs_last_error(void);
/* CU count, max clock (kHz), total HBM bytes and gcnArchName of device `dev`. */
int avs_device_info(int dev, int* cu_count, int* clock_khz, int64_t* hbm_bytes,
                    char* arch, int arch_len);

/* fp32 [n] <-> AVS_F16X2 [n] (n a multiple of 8; the fp32 side 16-byte, the f16x2 side 32-byte aligned).  Weights
 * are converted once per parameter version; the trunk's activations never leave the format.                      */
int avs_f16x2_pack_f32(const float* d_src, void* d_dst, int64_t n, avs_stream_t stream);
int avs_f16x2_unpack_f32(const void* d_src, float* d_dst, int64_t n, avs_stream_t stream);

/* AVS_F16P8: a value as fp16 hi + an 8-BIT remainder, 3 bytes: x ~ hi + (u - 128) * step(hi), hi = fp16(x),
 * step(hi) = ulp(max(|hi|, 2^-6)) / 256 (2^-24, the fp16 grid, below 2^-6), u = round((x - hi) / step) + 128 clamped to
 * 1..255 (128 when hi = 0): 19-20 significant bits, the absolute floor of AVS_F16X2; every remainder is itself an fp16
 * number - the lo half AVS_F16X2 holds for the same value.  Every run of 16 values is 48 bytes: the hi halves of
 * values 0-7, of values 8-15, the 16 remainder bytes.  The storage format of the inner block outputs of ResNet layers
 * 1-2 in the AVS_F16X2 trunk (features/extractors.py:29,65): written by avs_conv2d_nhwc_affine (AVS_Y_F16P8), read as
 * its residual (AVS_RES_F16P8) and as the input of avs_conv2d_nhwc_bnstats (AVS_X_F16P8), where the fp16 lo halves the
 * matrix cores take are rebuilt in registers: the same VALUES as the AVS_F16X2 input path on the same tensor, and the same
 * operand bits for |x| >= 2^-6 (below that the hi | lo split of a value is not unique - the 2^-24 floor - so outputs agree to
 * the format's resolution there, not bit for bit: tests/test_gpu_f16p8.py).
 * Both formats SATURATE at +-65504 and store a NaN as -65504 (the clamp is an fmin / fmax pair): the trunk's un-divided
 * inputs (up to 1131) with He-scale weights give raw stem outputs of O(10^3), far from the limit.
 * fp32 [n] <-> AVS_F16P8 [n], n a multiple of 16, both sides 16-byte aligned (tests, tools).                      */
int avs_f16p8_pack_f32(const float* d_src, void* d_dst, int64_t n, avs_stream_t stream);
int avs_f16p8_unpack_f32(const void* d_src, float* d_dst, int64_t n, avs_stream_t stream);

/* ---- dense contraction (K3, K6, K16-K20 of SURVEY §2.3) ------------------ */

/* Implicit-GEMM 2-D convolution on NHWC activations, weights [cout][kh][kw][cin]:
 *   y[n,ho,wo,co] = act( alpha * sum_{kh,kw,ci} x[n, ho*sh-ph+kh, wo*sw-pw+kw, ci]
 *                                              * w[co,kh,kw,ci] + bias[co] )
 * Replaces torch conv2d inside torchvision resnet50 / inception_v3 as called at
 * features/extractors.py:65,83.  Element strides allow padded / sliced views
 * (channel-concat outputs, pre-padded images).  cin must be a multiple of
 * 16 bytes / element size; out-of-image taps read as zero.                  */
typedef struct {
  int dtype;                /* AVS_F32 | AVS_BF16 (x, w, y); bias is fp32     */
  int n, h, w;              /* images, input rows, input pixels per row       */
  int cin, kh, kw;          /* reduction extent                               */
  int sh, sw, ph, pw;       /* stride, zero padding                           */
  int ho, wo, cout;         /* output extent                                  */
  int64_t x_img_stride;     /* elements between images                        */
  int64_t x_row_stride;     /* elements between input rows                    */
  int64_t x_px_stride;      /* elements between input pixels                  */
  int64_t w_row_stride;     /* elements between output channels in w (>= kh*kw*cin) */
  int64_t y_px_stride;      /* elements between output pixels (>= cout)       */
  int act;                  /* AVS_ACT_*                                      */
  float alpha;
  int w_layout;             /* AVS_W_ROWS: w[cout][w_row_stride], a row = (kh, kw, cin);
                             * AVS_W_KSTEP32 (kh*kw*cin a multiple of S = 32 bf16 / 16 fp32 elements, i.e. of a
                             * 64-byte step; not AVS_F32_ACC64): the same matrix stored
                             * reduction-step major, w[k / S][cout][S] - the 64 bytes of a filter that one
                             * reduction step reads sit next to the neighbouring filters', so the weight tile of
                             * a step is contiguous whole cache lines (+2..6 % on the ResNet layers);
                             * w_row_stride is ignored.  Weights are re-laid out once, offline.              */
  int variant;              /* 0 = the library's own choice.  Per-call overrides of the tile / staging variant (what the
                             * tests and the kernel-study tools force; there is NO process-global tuning state):
                             * AVS_TILE_128 / AVS_TILE_256 (bits 0-1): 128-row or, wherever the variant exists, 256-row
                             * output tiles; AVS_TILE_224 (avs_conv2d_nhwc_bnlocal, AVS_F16X2 only): the 224-row tile
                             * whose waves split the columns, for groups of 193..224 rows - what AVS_TILE_AUTO picks
                             * for such groups, AVS_TILE_256 keeps them on the 256-row tile (and keeps the AVS_F16X2
                             * stride-1 'same' layers - 3x3 under the BatchNorm epilogues, any KH x KW under bias + ReLU -
                             * on the tap-major walk instead of the shifted-row form);
                             * AVS_STAGING_GENERIC (bit 2): the general per-lane gather staging even where
                             * the scalar tap walk applies.  Results do not depend on it beyond fp32 summation order. */
  int formats;              /* 0, or AVS_F16P8 operands of the AVS_F16X2 1x1 forms (bits): AVS_X_F16P8 - the INPUT of
                             * avs_conv2d_nhwc_bnstats (1x1 / stride 1 on dense rows, cin a multiple of 32, >= 64);
                             * AVS_Y_F16P8 / AVS_RES_F16P8 - the OUTPUT / the RESIDUAL of avs_conv2d_nhwc_affine (cout and
                             * the row strides multiples of 16).  Strides stay in elements; a row of c elements is 3 c
                             * bytes.  Other combinations: AVS_E_UNSUPPORTED.                                           */
} avs_conv_desc;
enum { AVS_W_ROWS = 0, AVS_W_KSTEP32 = 1 };
enum { AVS_X_F16P8 = 1, AVS_Y_F16P8 = 2, AVS_RES_F16P8 = 4 };
enum { AVS_TILE_AUTO = 0, AVS_TILE_128 = 1, AVS_TILE_256 = 2, AVS_TILE_224 = 3, AVS_STAGING_GENERIC = 4 };

int avs_conv2d_nhwc(const avs_conv_desc* desc, const void* d_x, const void* d_w,
                    const float* d_bias, void* d_y, avs_stream_t stream);
/* The same with TWO destinations, AVS_F16X2: output columns [0, n_split) go to d_y (row stride desc->y_px_stride), columns
 * [n_split, cout) to d_y2 (row stride y2_px_stride, column c at c - n_split).  Convolutions that read the same input - an
 * Inception block's 1x1 heads (features/extractors.py:26,73-90) - run as ONE contraction over their stacked filters: the
 * input is fetched once, and the head that belongs to the block's concatenated output still lands in its channel slice.
 * n_split, both strides in multiples of 8 slots; each row stride need only cover its OWN columns (y_px_stride >= n_split,
 * y2_px_stride >= cout - n_split).  relu_cols > 0 (a multiple of 8): with desc->act = ReLU only the columns
 * below it are rectified - a stacked head whose bias + ReLU follow a pooling of its output (Inception's branch_pool, run as
 * convolution -> average pooling -> bias -> ReLU) keeps its raw values.                                                  */
int avs_conv2d_nhwc_split(const avs_conv_desc* desc, const void* d_x, const void* d_w, const float* d_bias, void* d_y,
                          int n_split, void* d_y2, int64_t y2_px_stride, int relu_cols, avs_stream_t stream);

/* The same convolution (no bias, no activation) that also produces the BatchNorm batch statistics of its output
 * in its epilogue and folds them into the affine of avs_bn_batch_stats: for every group g = output_row /
 * rows_per_group (equal-sized micro-batch groups; a shorter last group is allowed) and channel c,
 *   d_scale[g,c] = gamma[c] / sqrt(var + eps),  d_shift[g,c] = beta[c] - mean * d_scale[g,c],
 * mean / biased variance of the group's rows (fp32 accumulators, before the output is rounded).  DETERMINISTIC:
 * every row tile stores its column sums into its own slots of the workspace (no atomics) and a second small
 * kernel adds a group's slots in tile order.  Saves the separate read pass of avs_bn_batch_stats
 * (features/extractors.py:29,65: the ResNet trunk's train-mode BatchNorm).
 * avs_conv2d_bnstats_workspace_bytes: bytes of d_ws (uninitialised, 16-byte aligned) for this shape, or
 * AVS_E_UNSUPPORTED when rows_per_group < 64 (use avs_bn_batch_stats then).                                   */
int64_t avs_conv2d_bnstats_workspace_bytes(const avs_conv_desc* desc, int64_t rows_per_group);
int avs_conv2d_nhwc_bnstats(const avs_conv_desc* desc, const void* d_x, const void* d_w, void* d_y,
                            int64_t rows_per_group, const float* d_gamma, const float* d_beta, float eps,
                            float* d_scale, float* d_shift, void* d_ws, int64_t ws_bytes, avs_stream_t stream);

/* Convolution + the WHOLE batch-statistics BatchNorm (+ residual, + ReLU) in one launch, bf16, for equal-sized
 * groups that fit a 256-row output tile whole:
 *   y[m,:] = act( bn_g(conv(x)[m,:]) + residual[m,:] ),  g = m / rows_per_group,  act = desc->act.
 * A tile holds floor(256 / rows_per_group) whole groups, so a group's statistics are sums over that tile's own
 * accumulators: no traffic between workgroups, no atomics, deterministic; one read of x, one write of y, nothing
 * raw in HBM.  Replaces avs_conv2d_nhwc_bnstats + avs_bn_apply for the 14x14 / 7x7 layers of the train-mode
 * ResNet trunk (features/extractors.py:29,65).
 * avs_conv2d_bnlocal_tile_rows returns the rows of a tile that are used (> 0) when the shape takes this form, or
 * AVS_E_UNSUPPORTED: not bf16; cout not a multiple of the column tile; groups of fewer than 43 or more than 256
 * rows or filling less than 3/4 of a tile; a reduction of at most 128 bytes; unaligned output rows.          */
int avs_conv2d_bnlocal_tile_rows(const avs_conv_desc* desc, int64_t rows_per_group);
int avs_conv2d_nhwc_bnlocal(const avs_conv_desc* desc, const void* d_x, const void* d_w, void* d_y,
                            int64_t rows_per_group, const float* d_gamma, const float* d_beta, float eps,
                            const void* d_residual, int64_t ldr, avs_stream_t stream);

/* The same for groups LARGER than a tile, AVS_F16X2: a group = `cluster` consecutive tiles of rows_per_group / cluster
 * rows each (193..224: one 14x14 map per tile - the reference's micro-batches of 4 frames at ResNet-50's layer 3,
 * features/extractors.py:48).  Every tile computes its own centred statistics (mean, sum of squares about it) from its
 * accumulators, publishes them per column as 8-byte {value, epoch} granules (one agent-scope store each: the tag is the
 * flag), reads its partners' granules, and merges them by Chan's update in tile order - every tile of the group obtains
 * the same bits; then normalises its accumulators (+ residual, + desc->act) and writes y once.  Nothing raw in HBM, no
 * statistics / apply passes, deterministic.  The tiles of a group take consecutive block ids (they are dispatched together);
 * the wait for a partner is bounded: the first word of d_xchg counts the waves whose wait ran out (0 after a healthy
 * launch; check it where results are validated).
 * ASSUMES that this launch's workgroups reach the CUs in block order from ONE queue (what a process that owns its GPU gets):
 * complete groups then always finish and free the CUs the next groups need.  When several queues feed the chip at once with
 * kernels of this kind - two streams, or processes time-sharing one GPU - each can hold partial groups that keep the other's
 * partners out; the bounded waits then end the launch and the counter says so (observed with 4 processes on one GPU).
 * d_xchg: avs_conv2d_bncluster_workspace_bytes(...) bytes, 64-byte aligned, ZEROED ONCE by the caller when allocated and
 * then left alone; epoch: a value that is new for this buffer on every call (1, 2, 3, ...; never 0) - granules of earlier
 * calls are then recognisably stale, nothing needs clearing between calls.
 * AVS_E_UNSUPPORTED: not AVS_F16X2, tiles of fewer than 193 or more than 224 rows, cout not a multiple of 128.          */
int64_t avs_conv2d_bncluster_workspace_bytes(const avs_conv_desc* desc, int64_t rows_per_group, int cluster);
int avs_conv2d_nhwc_bncluster(const avs_conv_desc* desc, const void* d_x, const void* d_w, void* d_y,
                              int64_t rows_per_group, int cluster, const float* d_gamma, const float* d_beta, float eps,
                              const void* d_residual, int64_t ldr, void* d_xchg, int64_t xchg_bytes, uint32_t epoch,
                              avs_stream_t stream);
/* AVS_F16X2, 1x1 convolutions whose BatchNorm groups are too large for a tile (the expanding 1x1 layers of ResNet
 * layers 1-2: conv3 and the downsample branch, features/extractors.py:29,65): the batch statistics of the OUTPUT are
 * taken from the second moments of the narrow INPUT (avs_bn_gram_affine_f16x2: mean_y = W mean(a),
 * var_y[n] = w_n^T C w_n, C = a^T a / R - mean mean^T per group), then the convolution is ONE streaming pass with the
 * folded affine in its epilogue (avs_conv2d_nhwc_affine) - the wide output is written once and never re-read.
 *
 * avs_bn_gram_affine_f16x2: x [groups * rows_per_group, k] (k = 64 | 128, row stride lin_stride slots), w [n, k]
 * (n a multiple of 32), both AVS_F16X2.  With d_in_scale / d_in_shift fp32 [groups, k] x is the RAW output of the
 * convolution before, a = relu(x * in_scale + in_shift) (that layer's BatchNorm + ReLU: this pass is its apply
 * pass) and, with d_a_out (may be d_x itself), the finished activation is stored.  Out: the folded affine of the
 * output BatchNorm, d_scale / d_shift fp32 [groups, n] = gamma / sqrt(var_y + eps), beta - mean_y * scale.
 * Deterministic; ~1e-6 relative on var_y for post-ReLU inputs (E[a a^T] - m m^T is formed in fp32).
 *
 * avs_conv2d_nhwc_affine: y = act((conv(x) * scale[g] + shift[g]) + residual * res_scale[g] + res_shift[g]),
 * g = output_row / rows_per_group (>= 32), desc->dtype AVS_F16X2, 1x1 kernel without padding, more than 32 input
 * channels; scale / shift (and the optional residual affine: the residual is then itself a raw convolution output)
 * fp32 [groups, cout]; residual AVS_F16X2 [rows, cout] with row stride ldr or NULL; act = desc->act.              */
int avs_bn_gram_affine_f16x2(const void* d_x, int64_t lin_stride, int k, const float* d_in_scale,
                             const float* d_in_shift, const void* d_w, int64_t ldb, int n, int64_t rows_per_group,
                             int groups, const float* d_gamma, const float* d_beta, float eps, float* d_scale,
                             float* d_shift, void* d_a_out, int64_t lda, avs_stream_t stream);
int avs_conv2d_nhwc_affine(const avs_conv_desc* desc, const void* d_x, const void* d_w, void* d_y,
                           int64_t rows_per_group, const float* d_scale, const float* d_shift,
                           const void* d_residual, int64_t ldr, const float* d_res_scale, const float* d_res_shift,
                           avs_stream_t stream);

/* The whole ResNet-50 stem of the bf16 throughput path: uint8 frames [n,224,224,3] -> (x / denom - mean) / std ->
 * conv1 7x7/2 (64 channels) -> bn1 in batch-statistics mode over groups of frames_per_group frames -> ReLU ->
 * maxpool 3x3/2 pad 1 -> d_y bf16 [n,56,56,64]  (features/extractors.py:126-140 + children()[0:4] of the trunk, :29,65).
 * Because y -> relu(scale * y + shift) is monotone, the 3x3/2 pooling is done on the RAW convolution output on chip
 * before the statistics exist - as a max for the channels with gamma >= 0 and a min for the others (sign(scale) =
 * sign(gamma)): bit-identical to avs_frames_normalize_u8 + avs_conv2d_nhwc_bnstats + avs_bn_maxpool_nhwc on the
 * same convolution values, with 1.35 instead of 4.6 MB of HBM traffic per frame (the normalised image and the
 * 112x112x64 map never reach HBM).
 * apply = 1: d_y receives the finished activations (relu applied when relu != 0).
 * apply = 0: d_y receives the pooled RAW map; the consumer applies relu(scale * y + shift) per group and channel
 *            while staging it (the d_in_scale / d_in_shift operands of avs_conv1x1_affine_bf16 and
 *            avs_bn_gram_affine_bf16): one pass over the map less.
 * d_w: bf16 [64, ldw >= 224], a row = 7 kernel rows x 8 pixels x 4 channels (zero where kx = 7 or channel = 3).
 * d_scale / d_shift [n / frames_per_group, 64] receive bn1's folded affine.  Deterministic (per-tile partial sums
 * added in a fixed order).  d_ws: avs_stem_workspace_bytes(n) bytes, uninitialised.
 * n must be a multiple of frames_per_group (AVS_E_UNSUPPORTED otherwise: use the unfused sequence).                */
int64_t avs_stem_workspace_bytes(int n);
int avs_stem_conv_bn_pool_bf16(const uint8_t* d_frames, int n, float denom, const float* mean3, const float* std3,
                               const void* d_w, int64_t ldw, int frames_per_group, const float* d_gamma,
                               const float* d_beta, float eps, int apply, int relu, void* d_y, float* d_scale,
                               float* d_shift, void* d_ws, int64_t ws_bytes, avs_stream_t stream);

/* The ResNet-50 stem of the AVS_F16X2 (parity-grade) path in one kernel + a fold: uint8 frames [n,224,224,3] ->
 * conv1 7x7/2 of (x / denom - mean) / std -> the batch statistics of bn1 over groups of frames_per_group frames, and the
 * 3x3/2 pad-1 pooling of the RAW convolution output -> d_y AVS_F16X2 [n,56,56,64]: per channel the window's maximum
 * (gamma >= 0) or minimum (gamma < 0), i.e. maxpool(relu(bn1(.))) = relu(scale * d_y + shift) exactly (monotone map,
 * sign(scale) = sign(gamma)); the consumer applies that affine + ReLU (avs_bn_gram_affine_f16x2's d_in_scale / d_in_shift).
 * Replaces avs_frames_normalize_u8 + avs_conv2d_nhwc_bnstats + avs_bn_maxpool_nhwc of features/extractors.py:126-140 +
 * children()[0:4] (:29,65); the normalised image and the 112x112x64 map never reach HBM.
 * The normalisation is folded into the operands (x is affine in the byte value v, and bytes are exact fp16 numbers):
 *   d_w    AVS_F16X2 [64, ldw >= 224 slots], a row = 7 kernel rows x 8 pixels x 4 channels (zero where kx = 7):
 *          channels 0-2 = w[o,c,ky,kx] / (denom * std_c); channel 3 = -sum_c w[o,c,ky,kx] * mean_c / std_c, which the
 *          kernel multiplies by 1 for a pixel inside the image and 0 outside (the convolution's zero padding is of
 *          the NORMALISED input: only the taps inside contribute their constant).
 * Two fp16 MFMAs per product (the image has no lo half).  Statistics: per frame sums about a pivot (the frame's own
 * output at an interior pixel), frames merged by Chan's update in frame order: centred, deterministic.
 * d_scale / d_shift [n / frames_per_group, 64]: bn1's folded affine.  d_ws: avs_stem_f16x2_workspace_bytes(n) bytes.
 * n must be a multiple of frames_per_group (AVS_E_UNSUPPORTED otherwise: use the unfused sequence).                  */
int64_t avs_stem_f16x2_workspace_bytes(int n);
int avs_stem_conv_pool_f16x2(const uint8_t* d_frames, int n, const void* d_w, int64_t ldw, int frames_per_group, const float* d_gamma, const float* d_beta, float eps, void* d_y,
                             float* d_scale, float* d_shift, void* d_ws, int64_t ws_bytes, avs_stream_t stream);

/* 1x1 convolution + batch-statistics BatchNorm (+ residual, + ReLU) in one kernel, bf16, for equal-sized
 * groups of rows_per_group consecutive rows (a micro-batch of frames at one resolution):
 *   y[m,:] = act( bn_g(x[m,:] . w^T) + residual[m,:] ),  statistics of group g = m / rows_per_group.
 * One workgroup owns a whole group for a slab of channels and walks it twice (statistics, then the
 * normalised output), so the raw convolution never goes to HBM: replaces avs_conv2d_nhwc_bnstats +
 * avs_bn_apply for the 1x1 layers of the train-mode ResNet trunk
 * (features/extractors.py:65).  x rows at stride lin_stride; k, n and strides multiples of 8 elements.    */
int avs_conv1x1_bn_bf16(const void* d_x, int64_t lin_stride, int k, const void* d_w, int64_t ldb, int n,
                        int64_t rows_per_group, int groups, const float* d_gamma, const float* d_beta,
                        float eps, const void* d_residual, int64_t ldr, int relu, void* d_y, int64_t ldc,
                        avs_stream_t stream);

/* The same kernel reading a RAW convolution output as its input: the previous layer's BatchNorm + ReLU is applied
 * on the way in,  a[m,k] = bf16( relu( x[m,k] * d_in_scale[g,k] + d_in_shift[g,k] ) )  (avs_bn_apply's arithmetic:
 * bit-identical to running avs_bn_apply on x first), so the previous layer needs no apply pass at all: the
 * conv2 -> bn2 -> relu -> conv3 -> bn3 -> +identity -> relu tail of a ResNet bottleneck in two launches.
 * d_in_scale / d_in_shift are [groups, k] fp32 (avs_conv2d_nhwc_bnstats output); k <= 512 (else AVS_E_UNSUPPORTED).                      */
int avs_conv1x1_bn_in_bf16(const void* d_x, int64_t lin_stride, int k, const float* d_in_scale,
                           const float* d_in_shift, const void* d_w, int64_t ldb, int n,
                           int64_t rows_per_group, int groups, const float* d_gamma, const float* d_beta,
                           float eps, const void* d_residual, int64_t ldr, int relu, void* d_y, int64_t ldc,
                           avs_stream_t stream);

/* The expanding 1x1 layers of ResNet layers 1-2 in ONE streaming pass: the BatchNorm statistics of y = a . w^T
 * follow from the Gram matrix of the (narrow) input,
 *   mean_y[n] = w_n . mean(a),  var_y[n] = w_n^T C w_n,  C = a^T a / R - mean(a) mean(a)^T  per group of R rows,
 * so avs_bn_gram_affine_bf16 reads only a (k = 64 or 128 channels; a^T a and W . C on the matrix cores, C carried as
 * bf16 hi + lo = 16 significant bits, everything else fp32; deterministic) and writes the folded affine
 * d_scale / d_shift [groups, n] of bn(conv(a)); avs_conv1x1_affine_bf16 is then the convolution with that affine
 * (+ residual, + ReLU) in its epilogue - no statistics pass over the wide output.  Same operands and meaning as
 * avs_conv1x1_bn_in_bf16 (d_in_scale / d_in_shift = NULL: a is x itself; else a = bf16(relu(x * in_scale + in_shift)),
 * the previous layer's BatchNorm applied on the way in).  Replaces avs_conv1x1_bn[_in]_bf16 for k in {64, 128},
 * n % 32 == 0 (features/extractors.py:29,65: conv3 / downsample of the train-mode ResNet bottlenecks); other
 * shapes: AVS_E_UNSUPPORTED.  Statistics agree with the two-pass kernel's to fp32 rounding, not bit for bit.
 * d_res_scale / d_res_shift (both or neither; fp32 [groups, n]): the residual is itself a RAW convolution output
 * (the downsample branch of a bottleneck, :29) whose BatchNorm is folded into the add,
 *   y = act( conv(a) * scale + shift + residual * res_scale + res_shift ),
 * so the downsample branch needs no apply pass of its own.
 * d_a_out (avs_bn_gram_affine_bf16; NULL = off; needs d_in_scale): the transformed input a is also stored, bf16
 * [rows, lda >= k] - d_a_out == d_x (lda == lin_stride) overwrites the raw input in place - so that the convolution
 * pass can read a finished input (d_in_scale = NULL) instead of transforming it once per 128-column slab.         */
int avs_bn_gram_affine_bf16(const void* d_x, int64_t lin_stride, int k, const float* d_in_scale,
                            const float* d_in_shift, const void* d_w, int64_t ldb, int n, int64_t rows_per_group,
                            int groups, const float* d_gamma, const float* d_beta, float eps, float* d_scale,
                            float* d_shift, void* d_a_out, int64_t lda, avs_stream_t stream);
int avs_conv1x1_affine_bf16(const void* d_x, int64_t lin_stride, int k, const float* d_in_scale,
                            const float* d_in_shift, const void* d_w, int64_t ldb, int n, int64_t rows_per_group,
                            int groups, const float* d_scale, const float* d_shift, const void* d_residual,
                            int64_t ldr, const float* d_res_scale, const float* d_res_shift, int relu, void* d_y,
                            int64_t ldc, avs_stream_t stream);

/* The library holds NO process-global mutable state: the tile / staging variant of a convolution is chosen per call
 * (avs_conv_desc.variant), everything else by fixed rules.  (The kernel-study build, `make study` ->
 * lib/libavsum_hip_study.so, additionally exports void avs_debug_flags(int) and the avs_tune_* setters of the rules'
 * thresholds for tools/; the shipped library has no such code paths and no such symbols.)                        */

/* Batched C[b] = act(alpha * A[b] . B[b]^T + bias):  A [M,K] (row stride lda),
 * B [N,K] (row stride ldb; the nn.Linear weight layout), C [M,N] (ldc).
 * Replaces nn.Linear (models/av_model.py:10-15,29-31; models/attention.py:8-11;
 * features/extractors.py:193), the LSTM input projections (av_model.py:18-23),
 * the in/out projections of nn.MultiheadAttention (av_model.py:26) and the two
 * einsums of models/attention.py:21,23.  K, lda, ldb multiples of 16 bytes.  */
int avs_gemm_nt(int dtype, int m, int n, int k,
                const void* d_a, int64_t lda, int64_t stride_a,
                const void* d_b, int64_t ldb, int64_t stride_b,
                void* d_c, int64_t ldc, int64_t stride_c,
                const float* d_bias, int bias_mode, int64_t stride_bias,
                float alpha, int act, int batch, avs_stream_t stream);

/* ---- visual front end (K1, K2, K4, K5, K7) ------------------------------ */

/* uint8 HWC frames -> normalised, zero-padded NHWC images with 4 channels
 * (4th = 0), in `dtype`:  out[n, pad_t+y, pad_l+x, c] = (src/denom - mean[c]) / std[c]
 * (then the per-channel affine a[c]*v + b[c] when affine6 = {a0,a1,a2,b0,b1,b2}
 * is given: torchvision inception_v3 transform_input).  Replaces _preprocess_frame
 * (features/extractors.py:126-140; denom 1, i.e. NO /255) and the arithmetic of
 * _preprocess_inception (:142-155; denom 255).  out is [n, out_h, out_w, 4].
 * mean3 / std3 / affine6 are HOST pointers (copied into the launch).         */
int avs_frames_normalize_u8(int dtype, const uint8_t* d_src, int n, int h, int w,
                            float denom, const float* mean3, const float* std3,
                            const float* affine6, void* d_out, int out_h, int out_w,
                            int pad_t, int pad_l, avs_stream_t stream);

/* Upload by a PULL kernel: `workgroups` 256-thread blocks read h_src_mapped - PINNED host memory by its device-visible
 * address (hipHostMalloc / torch pin_memory: the same pointer) - over PCIe and store to d_dst.  The upload's footprint on
 * the chip is the caller's choice, so the frames of the next pass can stream in beside the current pass's kernels
 * (the PCIe-inclusive path of the pipeline; the reference hands over host arrays, features/extractors.py:43).           */
int avs_pull_copy_u8(const uint8_t* h_src_mapped, uint8_t* d_dst, int64_t bytes, int workgroups, avs_stream_t stream);

/* OpenCV-style bilinear resize of uint8 HWC frames (cv2.resize INTER_LINEAR,
 * fixed-point 11-bit coefficients): features/extractors.py:132,147.          */
int avs_resize_bilinear_u8(const uint8_t* d_src, int n, int sh, int sw,
                           uint8_t* d_dst, int dh, int dw, avs_stream_t stream);

/* Batch-statistics BatchNorm2d, the mode the reference actually runs its
 * ResNet-50 trunk in (features/extractors.py:29 never calls .eval()):
 * statistics per (group, channel) over the rows of the group's frames, biased
 * variance, eps.  d_group_rows[g]..d_group_rows[g+1] are the row ranges of the
 * groups (a group = one micro-batch of <=4 frames, extractors.py:48-56).
 * Writes the folded affine  scale[g,c] = gamma*rstd,  shift[g,c] = beta - mean*scale. */
int avs_bn_batch_stats(int dtype, const void* d_x, int64_t rows, int c, int64_t ldx,
                       const int64_t* d_group_rows, int groups,
                       const float* d_gamma, const float* d_beta, float eps,
                       float* d_scale, float* d_shift, avs_stream_t stream);

/* y = act( x*scale[g] + shift[g] (+ residual) ), rows of group g as above;
 * max_group_rows = the largest group's row count (sizes the launch).
 * groups==0 with d_group_rows NULL: one affine for all rows (folded eval BN). */
int avs_bn_apply(int dtype, const void* d_x, int64_t rows, int c, int64_t ldx,
                 const int64_t* d_group_rows, int groups, int64_t max_group_rows,
                 const float* d_scale, const float* d_shift,
                 const void* d_residual, int64_t ldr, int act,
                 void* d_y, int64_t ldy, avs_stream_t stream);

/* 2-D pooling on NHWC.  mode 0 = max (padding = -inf), 1 = average with
 * count_include_pad (torch defaults).  ResNet maxpool 3x3/2 p1, Inception
 * max 3x3/2 p0 and avg 3x3/1 p1.  y may be a channel slice (y_px_stride).
 * d_bias (fp32 [c], may be NULL) and act (AVS_ACT_NONE / AVS_ACT_RELU) are applied after the pooling:
 * y = act(pool(x) + bias).  Inception's branch_pool (avg_pool2d 3x3/1 -> 1x1 conv -> folded BN -> ReLU,
 * torchvision inception.py as loaded at features/extractors.py:26) runs as 1x1 conv WITHOUT bias -> this call:
 * averaging and a 1x1 convolution commute (both linear; the zero padding of count_include_pad commutes too), and the
 * pooling then moves cout instead of cin channels (4-10x fewer).                                                   */
int avs_pool2d_nhwc(int dtype, int mode, const void* d_x, int n, int h, int w, int c,
                    int64_t x_px_stride, int k, int s, int p, const float* d_bias, int act, void* d_y,
                    int ho, int wo, int64_t y_px_stride, avs_stream_t stream);

/* BatchNorm apply + ReLU + max pooling in one pass: y = maxpool_{k,s,p}( act( x*scale[g] + shift[g] ) ), g = the
 * group (d_group_rows ranges over INPUT rows, as in avs_bn_apply; groups == 0 / NULL: one affine) of the image.
 * The ResNet stem's bn1 -> relu -> maxpool (features/extractors.py:29) without writing the normalised
 * full-resolution map; bit-identical to avs_bn_apply followed by avs_pool2d_nhwc(max).                      */
int avs_bn_maxpool_nhwc(int dtype, const void* d_x, int n, int h, int w, int c, int64_t x_px_stride,
                        const int64_t* d_group_rows, int groups, const float* d_scale, const float* d_shift,
                        int relu, int k, int s, int p, void* d_y, int ho, int wo, int64_t y_px_stride,
                        avs_stream_t stream);

/* Global average pool: y[n,c] = mean over h*w (fp32 out).  (adaptive avgpool) */
int avs_global_avgpool_nhwc(int dtype, const void* d_x, int n, int hw, int c,
                            float* d_y, int64_t ldy, avs_stream_t stream);

/* Segment mean: out[s,:] = mean(x[seg[s]:seg[s+1], :]) summed in row order in
 * fp32 then divided (numpy .mean(axis=0), extractors.py:108-110); empty
 * segment -> zeros (extractors.py:44-45).                                   */
int avs_segment_mean_f32(const float* d_x, int64_t ldx, int d, const int64_t* d_seg,
                         int nseg, float* d_out, int64_t ldo, avs_stream_t stream);

/* Shot-boundary scan: d_sums[f, 0..2] = sum over the (strided) pixels of |H|,|S|,|V| differences
 * between frame f and f-1 in OpenCV's 8-bit HSV (d_sums[0,:] = 0): the per-frame content score of PySceneDetect's
 * ContentDetector is (sums / pixels).mean(), thresholded on the host (features/extractors.py:388-393).          */
int avs_hsv_frame_diff_u8(const uint8_t* d_frames, int n, int h, int w, int step, uint32_t* d_sums,
                          avs_stream_t stream);

/* ---- audio front end (K8-K12) ------------------------------------------- */

/* Reflect-pad a mono waveform by `pad` samples each side (torch.stft
 * center=True, pad_mode="reflect"); out has t + 2*pad samples (+ tail zeroed
 * up to out_len).                                                           */
int avs_reflect_pad_f32(const float* d_x, int64_t t, int pad, float* d_out,
                        int64_t out_len, avs_stream_t stream);

/* STFT of a (reflect-padded) waveform as a dense real DFT on the fp64 matrix cores:
 *   spec[f, c] = float( sum_n double(xpad[f*hop + n]) * basis_t[n, c] ),  f < frames, c < ncols
 * basis_t is [nfft, ncols_pad] float64 (window folded in; columns = re | im bins, zero padded to a
 * multiple of 64).  Replaces torch.stft inside torchaudio Spectrogram (features/extractors.py:237,242).
 * fp64 because log2(mel+1e-6) is ill-conditioned in quiet bins.            */
int avs_stft_f64(const float* d_xpad, int64_t xpad_len, int64_t frames, int hop, int nfft,
                 const double* d_basis_t, int ncols, int ncols_pad, float* d_spec,
                 avs_stream_t stream);

/* torchaudio MelSpectrogram / MFCC front end in ONE kernel (features/extractors.py:236-246: MelSpectrogram(sr,
 * n_mels) and MFCC(sr, n_mfcc) defaults: n_fft = win = 400, periodic Hann, hop 200, center + reflect padding, power
 * 2, onesided 201 bins): d_wave fp32 [t] (t > 200, 16-byte aligned) -> [1 + t / 200, nmel] each, any subset of
 *   d_log2mel = log2(mel + 1e-6)   (:245),   d_db = 10 log10(max(mel, 1e-10))   (AmplitudeToDB before its top_db
 *   clamp; d_max receives max(max(mel, 1e-10)) over the call for avs_clamp_topdb_f32; zero it first),   d_power = mel.
 * A workgroup stages the waveform span of its 32 frames in LDS (reflect padding by index, 16-byte loads), runs the
 * real DFT folded to half its length (d_window = the fp32 window values as float64 [400]; d_cos [204, 208] / d_sin
 * [200, 208] = cos / -sin(2 pi k n / 400), float64) on the fp64 matrix cores, then |X|^2, the sparse mel sum
 * (d_fb [201, nmel], filter m non-zero on bins d_fb_lo[m] .. d_fb_hi[m] - 1) and the log on chip: the spectrum
 * never reaches HBM.  Replaces avs_reflect_pad_f32 + avs_stft_f64 + avs_power_mel_f32 for this front end.          */
int avs_stft_mel_fused_f32(const float* d_wave, int64_t t, const double* d_window, const double* d_cos,
                           const double* d_sin, const float* d_fb, const int* d_fb_lo, const int* d_fb_hi, int nmel,
                           float* d_log2mel, float* d_db, float* d_power, float* d_max, avs_stream_t stream);
/* The same front end reduced to TIME MEANS per segment (a shot's slice of the track): d_blocks int32 [nblocks, 3] =
 * (first STFT frame, frames <= 32, segment) - every segment cut into runs of at most 32 frames, in order -,
 * d_seg_block int32 [nseg + 1] = first block of each segment, d_seg_frames int32 [nseg] = its frame count.  Out:
 * d_mean_log2 [nseg, ld_log2] = mean over the segment's frames of log2(mel + 1e-6), d_mean_db [nseg, ld_db] = mean of
 * max(10 log10(max(mel, 1e-10)), 10 log10(*d_max) - top_db) - by linearity the DCT / mfcc_proj of that mean IS the mean
 * of the MFCC rows (features/extractors.py:232-246 pool the per-frame matrices over time).  *d_max = the largest
 * clamped mel power the clamp is relative to:
 *   find_max = 0: given by the caller (an avs_stft_mel_fused_f32 call with only d_max requested: a second pass of the
 *                 DFT over the track) - nothing per frame reaches HBM;
 *   find_max = 1: found by THIS call over the frames of its blocks (a table that covers the whole track gives the track
 *                 maximum): ONE pass of the DFT that writes the blocks' unclamped dB rows to the workspace, then a
 *                 bandwidth-bound pass clamps and sums them - the same values in the same order, bit-identical to
 *                 find_max = 0 with that maximum.  *d_max is overwritten.
 * d_ws: avs_stft_mel_segmean_workspace_bytes(...) bytes; deterministic (block partial sums folded in order).          */
int64_t avs_stft_mel_segmean_workspace_bytes(int nblocks, int nmel, int want_log2, int want_db, int find_max);
int avs_stft_mel_segmean_f32(const float* d_wave, int64_t t, const double* d_window, const double* d_cos,
                             const double* d_sin, const float* d_fb, const int* d_fb_lo, const int* d_fb_hi, int nmel,
                             const int* d_blocks, int nblocks, const int* d_seg_block, const int* d_seg_frames, int nseg,
                             float* d_max, int find_max, float top_db, float* d_mean_log2, int64_t ld_log2,
                             float* d_mean_db, int64_t ld_db, void* d_ws, int64_t ws_bytes, avs_stream_t stream);
/* The same for a BATCH of tracks in one set of launches (find_max = 1 semantics per track): d_waves holds the tracks one
 * after another - track i at sample offset d_track_off[i] (a multiple of 4: 16-byte aligned), d_track_len[i] samples -,
 * d_blocks int32 [nblocks, 4] = (first STFT frame inside its track, frames <= 32, segment, track), segments numbered over
 * the whole batch (d_seg_block / d_seg_frames as above), d_max fp32 [ntracks] receives every track's maximum.  Four
 * launches for all tracks of a batch instead of five per track (configs[2]: 50 tracks).                                */
int avs_stft_mel_segmean_batch_f32(const float* d_waves, const int64_t* d_track_off, const int64_t* d_track_len, int ntracks,
                                   const double* d_window, const double* d_cos, const double* d_sin, const float* d_fb,
                                   const int* d_fb_lo, const int* d_fb_hi, int nmel, const int* d_blocks, int nblocks,
                                   const int* d_seg_block, const int* d_seg_frames, int nseg, float* d_max, float top_db,
                                   float* d_mean_log2, int64_t ld_log2, float* d_mean_db, int64_t ld_db, void* d_ws,
                                   int64_t ws_bytes, avs_stream_t stream);

/* Power spectrum -> mel filterbank -> log.  d_spec is [frames, 2*nbins]
 * (re | im per frame, from avs_gemm_nt against the windowed DFT basis);
 * d_fb is [nbins, nmel] (torchaudio melscale_fbanks layout).
 *   mode 0: out = log2(mel + 1e-6)          (features/extractors.py:245)
 *   mode 1: out = 10*log10(max(mel,1e-10)), and atomically folds the maximum
 *           of max(mel,1e-10) into *d_max, which the caller zeroed
 *           (AmplitudeToDB, first half)
 *   mode 2: out = mel
 *   mode 3: the MAGNITUDE spectrum sqrt(re^2+im^2) through the filterbank, out = ln(mel + 0.01): the VGGish
 *           log-mel front end (torchvggish mel_features.log_mel_spectrogram; features/extractors.py:188,216) */
int avs_power_mel_f32(const float* d_spec, int64_t frames, int nbins,
                      const float* d_fb, const int* d_fb_lo, const int* d_fb_hi,
                      int nmel, int mode, float* d_out, float* d_max,
                      avs_stream_t stream);

/* x = max(x, 10*log10(*d_max) - top_db) in place (AmplitudeToDB top_db clamp). */
int avs_clamp_topdb_f32(float* d_x, int64_t count, const float* d_max, float top_db,
                        avs_stream_t stream);
int avs_fill_f32(float* d_x, int64_t count, float value, avs_stream_t stream);

/* y = rint((clamp(x, lo, hi) - lo) * scale), round-half-to-even, values kept as float: the 8-bit quantiser of the
 * VGGish post-processor (torchvggish Postprocessor.postprocess; features/extractors.py:188,216).              */
int avs_quantize_f32(const float* d_x, int64_t count, float lo, float hi, float scale, float* d_y,
                     avs_stream_t stream);

/* Channel mix-down + rational resampling by a polyphase FIR (the reference delegates to
 * pydub/ffmpeg set_channels(1).set_frame_rate(16000), features/extractors.py:364-378, and averages channels at
 * :326-328):  y[i*up + p] = sum_k mono[i*down + k - width] * d_taps[p*ntaps + k],  mono = mean over the
 * interleaved channels of d_x [t, channels], zero outside the clip.  The taps come from the host (audio.py).   */
int avs_resample_f32(const float* d_x, int64_t t, int channels, const float* d_taps, int up, int down,
                     int ntaps, int width, float* d_y, int64_t out_len, avs_stream_t stream);

/* ---- importance scorer (K17-K19) ---------------------------------------- */

/* Batched LSTM recurrences, PyTorch gate order i,f,g,o, h0=c0=0
 * (nn.LSTM at models/av_model.py:18-23,39-40).
 *   d_xproj [rows, ndir*4H]: x_t.W_ih^T + b_ih + b_hh for every direction
 *           (direction d occupies columns d*4H..), rows = sum of seq lengths;
 *   d_whh_t [ndir, H, 4H]: W_hh TRANSPOSED per direction;
 *   d_seq_rows[s]..d_seq_rows[s+1]: the rows of sequence s;
 *   reverse_mask bit d set = direction d runs t = T-1..0;
 *   d_out [rows, ldo]: h_t of direction d at columns out_col0 + d*H.
 *   variant: 0 = the library's choice.  hidden = 256 keeps part of W_hh^T on chip for the whole sequence
 *           (AVS_LSTM_RESIDENT_20_8: 20 of a thread's 64 row-vectors in registers + 8 in LDS, the default there;
 *           AVS_LSTM_RESIDENT_16_8: 16 + 8) instead of streaming all of it from L2 every step (AVS_LSTM_STREAM).
 *           Same arithmetic in the same order: bit-identical outputs.                                             */
enum { AVS_LSTM_AUTO = 0, AVS_LSTM_STREAM = 1, AVS_LSTM_RESIDENT_20_8 = 2, AVS_LSTM_RESIDENT_16_8 = 3 };
int avs_lstm_f32(const float* d_xproj, const float* d_whh_t, int hidden, int ndir,
                 unsigned reverse_mask, const int64_t* d_seq_rows, int nseq,
                 float* d_out, int64_t ldo, int out_col0, int variant, avs_stream_t stream);

/* Attention core of nn.MultiheadAttention fed [B,T,E] WITHOUT batch_first
 * (models/av_model.py:26,44): for every time-step t and head h,
 * softmax over the B axis.  d_qkv [B*T, 3E] (q|k|v, already projected),
 * d_ctx [B*T, E].  One wave per (t, head, b); wave-shuffle softmax.          */
int avs_mha_batchaxis_f32(const float* d_qkv, int b, int t, int e, int heads,
                          float* d_ctx, avs_stream_t stream);

/* scores[r] = sigmoid( dot(hid[r,:], w2) + b2 )  (scorer.2 + Sigmoid,
 * models/av_model.py:30).                                                    */
int avs_score_head_f32(const float* d_hid, int64_t rows, int d, int64_t ldh,
                       const float* d_w2, const float* d_b2, float* d_scores,
                       avs_stream_t stream);

/* Fused attention core of models/attention.py:21-24 (flash style: the [T,T] scores never reach memory):
 *   ctx[b,q,h*D+:] = softmax_k( Q[b,q,h,:] . K[b,k,h,:] / sqrt(D) ) . V[b,k,h,:]
 * d_q / d_k / d_v are the projected [b*t, heads*D] matrices (row stride ld), head_dim D in {64, 128, 256}.
 * fp32 MFMA for both products, per-query online softmax folded with one wave shuffle.                      */
int avs_mhsa_flash_f32(const float* d_q, const float* d_k, const float* d_v, int64_t ld, int b, int t,
                       int heads, int head_dim, float* d_ctx, int64_t ldo, avs_stream_t stream);
/* The same fused core with AVS_F16X2 operands (d_q / d_k / d_v: the projections after avs_f16x2_pack_f32, row stride
 * ld slots; d_ctx fp32): every product hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_f16 (2^-21 relative, fp32
 * accumulation; the probabilities are split in registers), on 16-query tiles (v_mfma_f32_16x16x32_f16: 64
 * accumulator + 64 Q-fragment registers per lane at head dim 256, two waves per SIMD).  Whole forward at E = 1024,
 * H = 4, T = 5000: 1.40 ms against 2.69 ms for the fp32 form and 2.30 ms for the batched-GEMM path that materialises
 * the 400 MB of scores (models/attention.py:21-24).                                                               */
int avs_mhsa_flash_f16x2(const void* d_q, const void* d_k, const void* d_v, int64_t ld, int b, int t, int heads,
                         int head_dim, float* d_ctx, int64_t ldo, avs_stream_t stream);

/* Row softmax in place: x[r, 0:n] for `rows` rows of stride ldx
 * (torch.softmax(dim=-1) at models/attention.py:22).                         */
int avs_softmax_rows_f32(float* d_x, int64_t rows, int n, int64_t ldx,
                         avs_stream_t stream);
/* Backward of that softmax, in place on the upstream gradient: dp[r, j] <- alpha * p[r, j] * (dp[r, j] -
 * sum_k p[r, k] dp[r, k]), the gradient with respect to the unscaled scores of softmax(alpha * q.k^T)
 * (autograd through models/attention.py:21-22; the reference class is an ordinary autograd module).       */
int avs_softmax_bwd_rows_f32(const float* d_p, float* d_dp, int64_t rows, int n, int64_t ld, float alpha,
                             avs_stream_t stream);

/* ---- scorer backward (K22; scripts/train_av_model.py:86-96) ------------------ */
/* The dense parts of the backward reuse avs_gemm_nt on transposed copies (dX = dY.W uses W^T as the [N,K]
 * operand, dW = dY^T.X uses dY^T and X^T); the entry points below are the rest.                           */

/* dst[c, r] = src[r, c]; dst row stride ld_dst >= rows (pad columns are left untouched).                  */
int avs_transpose_f32(const float* d_src, int rows, int cols, int64_t ld_src, float* d_dst, int64_t ld_dst,
                      avs_stream_t stream);
/* out[c] = sum_r x[r,c] * (row_weight ? row_weight[r] : 1): bias gradients; deterministic order.          */
int avs_colsum_f32(const float* d_x, int64_t rows, int cols, int64_t ld, const float* d_row_weight,
                   float* d_out, avs_stream_t stream);
/* out = dy * (keep ? keep : 1) * (relu_out > 0): gradient through Dropout(keep = mask/(1-p)) and ReLU
 * (models/av_model.py:10-15).                                                                             */
int avs_relu_dropout_bwd_f32(const float* d_dy, const float* d_relu_out, const float* d_keep, int64_t n,
                             float* d_out, avs_stream_t stream);
int avs_mul_f32(const float* d_a, const float* d_b, int64_t n, float* d_out, avs_stream_t stream);
/* dz[r] = ds[r]*s[r]*(1-s[r]);  dhid_pre[r,k] = dz[r]*w2[k]*(hid[r,k] > 0)   (scorer.2 + Sigmoid and the
 * ReLU of scorer.0 backwards, models/av_model.py:29-31).                                                  */
int avs_score_head_bwd_f32(const float* d_dscores, const float* d_scores, const float* d_hid, int64_t rows,
                           int d, int64_t ldh, const float* d_w2, float* d_dz, float* d_dhid_pre,
                           avs_stream_t stream);
/* avs_lstm_f32 that also saves the post-activation gates [rows, ndir*4H] (i,f,g,o) and the cell state
 * [rows, ndir*H] for the backward (the forward of scripts/train_av_model.py:88).  variant: AVS_LSTM_AUTO (hidden = 256:
 * W_hh^T partly resident in registers / LDS, as avs_lstm_f32 does) or AVS_LSTM_STREAM; bit-identical outputs.          */
int avs_lstm_train_fwd_f32(const float* d_xproj, const float* d_whh_t, int hidden, int ndir,
                           unsigned reverse_mask, const int64_t* d_seq_rows, int nseq, float* d_out,
                           int64_t ldo, int out_col0, float* d_gates, float* d_cell, int variant, avs_stream_t stream);
/* Backward through time (loss.backward() of scripts/train_av_model.py:93): from dL/dh_t (d_dout, same layout as the
 * forward's d_out) to the gradient of the pre-activations d_dxproj [rows, ndir*4H].  d_whh is W_hh in its ORIGINAL
 * layout [ndir, 4H, H].  variant: AVS_LSTM_AUTO (hidden = 256: W_hh partly resident on chip, the step's saved inputs
 * fetched one step ahead; sums in another fixed order than the streaming kernel's - equal to rounding, deterministic)
 * or AVS_LSTM_STREAM.                                                                                               */
int avs_lstm_bwd_f32(const float* d_dout, int64_t ldo, int out_col0, const float* d_gates,
                     const float* d_cell, const float* d_whh, int hidden, int ndir, unsigned reverse_mask,
                     const int64_t* d_seq_rows, int nseq, float* d_dxproj, int variant, avs_stream_t stream);

/* The hidden = 256 recurrence with ONE recurrence split over FOUR CUs (models/av_model.py:39-40 forward,
 * scripts/train_av_model.py:94-95 backward): each of the four workgroups of a recurrence owns 64 hidden units and keeps its
 * quarter of W_hh in registers for the whole sequence; per time step the four exchange the step's vector (h_t forward, the
 * gate gradients backward) through tagged 8-byte granules in the workspace.  Same arguments and bit-identical outputs as
 * avs_lstm_f32 / avs_lstm_train_fwd_f32 (d_gates and d_cell given: both or neither) / avs_lstm_bwd_f32 with AVS_LSTM_AUTO;
 * a time step costs ~1.6 us (forward) / 1.85 us (backward) instead of 5.2 / 5.3: for FEW recurrences (one video per
 * training step, a few dozen videos at inference).  A launch runs 4 * ndir * nseq workgroups of 512 threads, one per CU: up
 * to 64 recurrences are all resident at once; more run in rounds (partners are dispatched together: complete groups always
 * finish) and past ~250 recurrences avs_lstm_f32's one recurrence per CU is the faster use of the chip.
 * d_ws: avs_lstm_split_workspace_bytes(ndir, nseq) bytes, 8-byte aligned, ZEROED ONCE by the caller before its first use
 * and then left alone; epoch: tags of a launch are epoch + 1 ... epoch + longest sequence - the caller passes values whose
 * ranges do not overlap from launch to launch on the same workspace (e.g. a running sum of rows + 1) and stay below 2^32
 * (tag 0 is the zeroed workspace's: before the sum would wrap, zero the workspace again and start over).
 * The FIRST 64 bytes of the workspace hold an error word (uint32, first of them): the number of workgroups whose bounded
 * wait (2^20 polls: of the order of a second) for a partner ran out - 0 after a healthy launch; such a launch ends, its
 * outputs are incomplete.  The same single-queue assumption as avs_conv2d_nhwc_bncluster: not for GPUs time-shared between
 * processes, nor for two streams running such kernels at once.                                                       */
size_t avs_lstm_split_workspace_bytes(int ndir, int nseq);
int avs_lstm_split_f32(const float* d_xproj, const float* d_whh_t, int hidden, int ndir, unsigned reverse_mask,
                       const int64_t* d_seq_rows, int nseq, float* d_out, int64_t ldo, int out_col0, float* d_gates,
                       float* d_cell, void* d_ws, size_t ws_bytes, unsigned epoch, avs_stream_t stream);
int avs_lstm_bwd_split_f32(const float* d_dout, int64_t ldo, int out_col0, const float* d_gates, const float* d_cell,
                           const float* d_whh, int hidden, int ndir, unsigned reverse_mask, const int64_t* d_seq_rows,
                           int nseq, float* d_dxproj, void* d_ws, size_t ws_bytes, unsigned epoch, avs_stream_t stream);

/* ---- fusion (K13-K15) --------------------------------------------------- */

/* out[i,j] = sqrt(sum_d (double(v[i,d]) - double(a[j,d]))^2), float64 out
 * (scipy cdist "euclidean" at features/fusion.py:11).                        */
int avs_cdist_f64(const float* d_v, int tv, const float* d_a, int ta, int d,
                  double* d_out, avs_stream_t stream);

/* Exact DTW on a cost matrix [n,m] (features/fusion.py:15-18 intent; SURVEY
 * A.9): D = C + min(up, left, diag) with tie order up, left, diag; path from
 * (0,0) to (n-1,m-1) written as int64 pairs to d_path (capacity n+m-1 pairs),
 * its length to *d_path_len, total cost to *d_cost.  The three live
 * anti-diagonals stay in LDS (n <= 6400); the workspace holds one predecessor
 * byte per cell.                                                            */
int64_t avs_dtw_workspace_bytes(int n, int m);
int avs_dtw_path_f64(const double* d_cost_matrix, int n, int m, void* d_workspace,
                     int64_t workspace_bytes, int64_t* d_path, int64_t* d_path_len,
                     double* d_cost, avs_stream_t stream);

/* out[u,:] = x[idx[u],:] * float(w[u])  (features/fusion.py:28-32).          */
int avs_gather_scale_f32(const float* d_x, int64_t ldx, int d, const int64_t* d_idx,
                         const double* d_w, int count, float* d_out,
                         avs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AVSUM_HIP_H */
