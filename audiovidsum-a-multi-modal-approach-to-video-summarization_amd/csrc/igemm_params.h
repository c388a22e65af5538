// Shared by the contraction kernels of libavsum_hip.so (igemm.hip, local224.hip): the launch parameters, the epilogue
// forms and the inline-asm helpers of the hand-counted pipelines.
#pragma once
#include "avs_internal.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct IgemmParams {
  const char* x;
  const char* w;
  char* y;
  const float* bias;
  int M, N, K;
  int HoWo, Wo, H, W, cin, KW;
  int sh, sw, ph, pw;
  long long x_img_stride, x_row_stride, x_px_stride;
  long long ldb, ldc;
  long long sA, sB, sC, sBias;
  float alpha;
  int act, bias_mode;
  int tiles_n;
  // EPI_STATS (fused BatchNorm batch statistics): every row tile writes, for each group it overlaps, its column sums
  // and sums of squares of the fp32 accumulators to its OWN slot stat_part[tile][slot][sum | sumsq][N] (plain stores,
  // no atomics); bn_fold_kernel adds a group's slots in tile order, so the statistics are reproducible bit for bit
  float* stat_part;
  int stat_slots;    // slots per row tile = groups a tile can overlap
  int rows_per_group;
  long long lin_stride;  // >= 0: output row m reads input row m at x + m*lin_stride (no (n,ho,wo) decode needed)
  // EPI_BNLOCAL (whole BatchNorm in the epilogue): BatchNorm parameters, optional residual
  const float* gamma;
  const float* beta;
  float eps;
  const char* residual;
  long long ldr;
  int split;         // fp32 operands only: 1 = products on the bf16 matrix cores as hi*hi + hi*lo + lo*hi (AVS_F32_SPLIT)
  int tall;          // 1: the 256-row tile variants (WR = 4)
  int w_kstep;       // 1: w is stored reduction-step major (AVS_W_KSTEP32), [K / S][N][S], S = 32 bf16 / 16 fp32: the 64 bytes a B row needs
                     //    in one step sit next to the neighbouring rows' (whole cache lines per DMA instruction)
  int tile_rows;     // EPI_BNLOCAL: rows of the tile that are used (whole groups), also the pitch between tiles
  // EPI_AFFINE (AVS_F16X2): y = act((conv * scale[g] + shift[g]) + residual (* res_scale[g] + res_shift[g])), the folded
  // affines given per group of rows_per_group rows: gamma / beta point at scale / shift [groups, N]
  int affine;
  const float* res_scale;
  const float* res_shift;
  int variant;       // avs_conv_desc.variant: AVS_TILE_128 / AVS_TILE_256 (bits 0-1), AVS_STAGING_GENERIC (bit 2)
  // AVS_F16P8 operands of the AVS_F16X2 1x1 forms (avs_conv_desc.formats): the input of the convolution + statistics
  // form (fetched into registers, the lo halves rebuilt there), the output / the residual of the given-affine form
  int x_p8, y_p8, res_p8;
  // two destinations (AVS_F16X2, avs_conv2d_nhwc_split): output columns >= nsplit go to y2 (row stride ldc2, column c at
  // c - nsplit) - several 1x1 convolutions that read the same input run as ONE contraction with their filters stacked
  char* y2;
  long long ldc2;
  int nsplit;
  int relu_cols;     // > 0: only the columns below it take the ReLU of a bias + ReLU epilogue
  // clustered tile-local BatchNorm (local224.hip): a group = `cluster` consecutive tiles of tile_rows rows; every wave
  // publishes its tile's (mean, centred sum of squares) per column as 8-byte {value, epoch} granules in xchg
  // [tiles_m][tiles_n][4 waves][64] and reads its partners' - one exchange, merged by Chan's update in tile order
  int cluster;
  unsigned epoch;
  unsigned long long* xchg;
  unsigned* xerr;    // a counter of waves whose bounded wait ran out (0 = every exchange completed)
#ifdef AVS_STUDY
  int debug;  // ablation switches of the kernel-study build (tools/): 1 = skip output stores, 2 = skip A/B loads, ...
#endif
};

enum { EPI_PLAIN = 0, EPI_STATS = 1, EPI_ANY = 2, EPI_BRELU = 3, EPI_BNLOCAL = 5, EPI_AFFINE = 6 };
constexpr int BNLOCAL_MAX_GROUPS = 6;  // groups per 256-row tile (rows_per_group >= 43)
constexpr int STATS_MIN_GROUP_ROWS = 64;  // EPI_STATS: a wave's 64 rows then overlap at most two groups

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// Passes a fragment THROUGH an empty asm: every later use of it depends on this statement, so it cannot be scheduled
// ahead of the (volatile) wait that precedes the statement.
__device__ __forceinline__ void avs_pin(uint4& v) {
  u32x4 r = __builtin_bit_cast(u32x4, v);
  asm volatile("" : "+v"(r));
  v = __builtin_bit_cast(uint4, r);
}

__device__ __forceinline__ void avs_pin2(uint2& v) {
  u32x2 r = __builtin_bit_cast(u32x2, v);
  asm volatile("" : "+v"(r));
  v = __builtin_bit_cast(uint2, r);
}
__device__ __forceinline__ uint2 avs_lds_read_b64(unsigned byte_addr) {
  uint2 v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(byte_addr));
  return v;
}
__device__ __forceinline__ uint4 avs_lds_read_b128(unsigned byte_addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
  return v;
}


// local224.hip: the AVS_F16X2 tile-local BatchNorm form on 224-row tiles (one group of 193..224 rows per tile)
bool igemm_h2_local224_ok(const IgemmParams& p, int dtype);
void igemm_h2_local224_launch(const IgemmParams& p, bool spatial, dim3 grid, hipStream_t stream);
