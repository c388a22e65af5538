// AVS_F16X2: BatchNorm batch statistics of y = a . w^T WITHOUT computing y, from the second moments of a - and the
// BatchNorm + ReLU of the layer before applied on the way (the pass IS that layer's apply pass: it reads the raw
// convolution output once and writes the finished activation back in place).
//
//   mean_y[n] = w_n . mean(a),   var_y[n] = w_n^T C w_n,   C = a^T a / R - mean(a) mean(a)^T   (K x K, per group)
//
// For the expanding 1x1 layers of ResNet layers 1-2 (K = 64 / 128 input channels, N = 4K outputs, groups of 3136 / 784
// rows - far too large for a tile) this turns "convolution + statistics, then an apply pass over the 4K-wide output"
// into ONE streaming convolution pass with the affine in its epilogue (avs_conv2d_nhwc_affine): the wide output is
// written once and never re-read.  One workgroup per group:
//   1. rows come in as runs of 8 channels (32 bytes: fp16 hi | lo), consecutive lanes taking consecutive runs of a row
//      (whole cache lines per load / store instruction); a = relu(x * in_scale + in_shift) in fp32 (XF), a split into
//      fp16 hi | lo ONCE: stored back to HBM (the finished activation) and, transposed, into the LDS tile
//      [channel][64 rows] (the reduction index of a^T a is the row);  a^T a on the matrix cores as
//      hi*hi + hi*lo + lo*hi (v_mfma_f32_32x32x16_f16, fp32 accumulators over the whole group);
//   2. C in fp32 from the accumulators, scaled by a power of two so that its largest entry is <= 1, split into
//      fp16 hi | lo images in LDS;
//   3. T^T = C . W^T on the matrix cores (three MFMAs per step), var_y[n] = sum_l T[n,l] W[n,l] in the lane that
//      holds column n (its own weight row), mean_y on the VALU, then the folded affine.
// Accuracy: fp32 sums of products that are exact to 2^-21; E[a a^T] - m m^T is formed once, in fp32, on post-ReLU
// BatchNorm outputs (mean^2 ~ variance): ~1e-6 relative on var_y (tests/test_gpu_f16x2.py, against float64).
// Deterministic (fixed reduction orders, no atomics).
#include "avs_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GramH2Params {
  const char* x;
  const char* w;
  const float* in_scale;
  const float* in_shift;
  const float* gamma;
  const float* beta;
  float* scale;
  float* shift;
  char* a_out;   // XF only, may be x itself: the finished activation a (f16x2)
  long long lin_stride, ldb, lda;
  int N, rows_per_group;
  float eps;
};

template <int K, bool XF>
__global__ __launch_bounds__(256, 2) void bn_gram_affine_h2_kernel(GramH2Params p) {
  static_assert(K == 64 || K == 128, "input widths of the layer-1 / layer-2 expanding convolutions");
  constexpr int CPR = K / 8;              // runs of 8 channels per row
  constexpr int TR = 64;                  // rows per LDS tile = lanes of a wave
  constexpr int NP = CPR / 4;             // runs per thread and tile
  constexpr int RPP = 256 / CPR;          // rows per pass of the 256 threads: thread t takes run t % CPR of row t / CPR
  constexpr int PT = TR * 2 + 16;         // bytes per channel row of a transposed tile plane (16-byte padding)
  constexpr int PLANE = K * PT + CPR * 16;   // one plane (hi or lo) of a tile; the rows of run c are skewed by 16 c bytes
  constexpr int TILE_BYTES = 2 * PLANE;
  constexpr int KB = K / 32;              // 32-channel blocks per side
  constexpr int NBLK = (KB * KB) / 4;     // Gram blocks per wave: 1 (K = 64) or 4 (K = 128: one block row)
  constexpr int PC = K * 2 + 16;          // bytes per row of a C image
  constexpr int C_BYTES = K * PC;
  constexpr int MAIN_BYTES = 2 * TILE_BYTES > 2 * C_BYTES ? 2 * TILE_BYTES : 2 * C_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[MAIN_BYTES];
  __shared__ float mbar[K];
  __shared__ float q2[K];
  __shared__ __attribute__((aligned(16))) float xaf[2][XF ? K : 4];   // the input affine of this group: scale | shift

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const long long g = blockIdx.x;
  const int R = p.rows_per_group;
  const char* __restrict__ xg = p.x + g * R * p.lin_stride * 4;

  // (Channel sums and sums of squares are NOT kept per thread: the sums come from one more pair of MFMAs against a
  // ones operand, the squares are the diagonal of a^T a.)
  if constexpr (XF) {
    for (int i = t; i < K; i += 256) {
      xaf[0][i] = p.in_scale[g * K + i];
      xaf[1][i] = p.in_shift[g * K + i];
    }
    __syncthreads();
  }

  // Thread t takes run c = t % CPR (8 channels) of rows t / CPR + RPP i of a tile: a wave's loads and stores cover whole
  // rows (full cache lines; with a lane per ROW every 128-byte line was requested by eight instructions, 16 bytes at a
  // time).  The transposed 2-byte LDS writes of a wave then go to CPR different channel rows at once: the rows of run c
  // are skewed by 16 c bytes, which spreads them over the banks.
  const int c_run = t % CPR, r_loc = t / CPR;
  uint4 rh[NP], rl[NP];
  auto gload = [&](int tile) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = tile * TR + r_loc + RPP * i;
      if (row < R) {
        const uint4* src = reinterpret_cast<const uint4*>(xg + ((long long)row * p.lin_stride + 8 * c_run) * 4);
        rh[i] = src[0];
        rl[i] = src[1];
      } else {
        rh[i] = make_uint4(0u, 0u, 0u, 0u);
        rl[i] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  float xs[XF ? 8 : 1], xh[XF ? 8 : 1];   // this thread's 8 channels of the input affine
  if constexpr (XF) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xs[j] = xaf[0][8 * c_run + j];
      xh[j] = xaf[1][8 * c_run + j];
    }
  }
  auto xform_store = [&](int buf, int tile) {
    char* hiT = lds + buf * TILE_BYTES + (8 * c_run) * PT + c_run * 16;
    char* loT = hiT + PLANE;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int rt = r_loc + RPP * i;         // row inside the tile
      const int row = tile * TR + rt;
      uint4 hi = rh[i], lo = rl[i];
      if (row < R) {
        if constexpr (XF) {
          float v[8];
          avs_f16x2_join8(hi, lo, v);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaxf(fmaf(v[j], xs[j], xh[j]), 0.f);
          avs_f16x2_split8(v, hi, lo);   // the statistics below are those of the STORED activation (hi + lo)
          if (p.a_out != nullptr) {
            uint4* dst = reinterpret_cast<uint4*>(p.a_out + (((g * R + row) * p.lda) + 8 * c_run) * 4);
            dst[0] = hi;
            dst[1] = lo;
          }
        }
      }
      // transposed: channel 8 c + j, row rt (rows past the group are zeros)
      const unsigned hw[4] = {hi.x, hi.y, hi.z, hi.w}, lw[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<unsigned short*>(hiT + (2 * j) * PT + rt * 2) = (unsigned short)hw[j];
        *reinterpret_cast<unsigned short*>(hiT + (2 * j + 1) * PT + rt * 2) = (unsigned short)(hw[j] >> 16);
        *reinterpret_cast<unsigned short*>(loT + (2 * j) * PT + rt * 2) = (unsigned short)lw[j];
        *reinterpret_cast<unsigned short*>(loT + (2 * j + 1) * PT + rt * 2) = (unsigned short)(lw[j] >> 16);
      }
    }
  };
  // 32 channels (block b) x 16 rows (step s) of a plane as an MFMA operand: lane (lr, lh) takes channel 32 b + lr of
  // rows 16 s + 8 lh .. + 7 = 16 contiguous bytes
  auto frag = [&](const char* plane, int b, int s) -> avs_f16x8 {
    const int ch = 32 * b + lr;
    return __builtin_bit_cast(avs_f16x8, *reinterpret_cast<const uint4*>(plane + ch * PT + (ch >> 3) * 16 + (16 * s + 8 * lh) * 2));
  };

  f32x16 acc[NBLK], accs;   // accs: channel sums of block bi (rows of a^T . 1)
#pragma unroll
  for (int b = 0; b < NBLK; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) accs[e] = 0.f;
  const _Float16 one = (_Float16)1.0f;
  const avs_f16x8 ones = {one, one, one, one, one, one, one, one};

  // ---- 1. a^T a and the channel sums over the group's rows
  const int tiles = (R + TR - 1) / TR;
  gload(0);
  xform_store(0, 0);
  __syncthreads();
  for (int tile = 0; tile < tiles; ++tile) {
    const int buf = tile & 1;
    if (tile + 1 < tiles) gload(tile + 1);   // in flight during the matrix work below
    const char* hiT = lds + buf * TILE_BYTES;
    const char* loT = hiT + PLANE;
#pragma unroll
    for (int s = 0; s < TR / 16; ++s) {
      if constexpr (K == 64) {
        const int bi = wave >> 1, bj = wave & 1;
        const avs_f16x8 ah = frag(hiT, bi, s), al = frag(loT, bi, s), bh = frag(hiT, bj, s), bl = frag(loT, bj, s);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[0], 0, 0, 0);
        if (bj == 0) {   // (wave-uniform) waves 0 and 2 also sum the channels of blocks 0 and 1
          accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ones, accs, 0, 0, 0);
          accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ones, accs, 0, 0, 0);
        }
      } else {
        const avs_f16x8 ah = frag(hiT, wave, s), al = frag(loT, wave, s);
        accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ones, accs, 0, 0, 0);
        accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ones, accs, 0, 0, 0);
#pragma unroll
        for (int bj = 0; bj < KB; ++bj) {
          const avs_f16x8 bh = frag(hiT, bj, s), bl = frag(loT, bj, s);
          acc[bj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[bj], 0, 0, 0);
          acc[bj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[bj], 0, 0, 0);
          acc[bj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[bj], 0, 0, 0);
        }
      }
    }
    if (tile + 1 < tiles) xform_store(buf ^ 1, tile + 1);
    __syncthreads();
  }

  // channel means (every column of accs holds the same sums: column 0 writes them) and mean squares (the diagonal
  // of a^T a, in the waves that hold a diagonal block)
  const float inv_r = 1.f / (float)R;
  {
    const int bi = K == 64 ? (wave >> 1) : wave;
    const bool sums_here = K == 64 ? (wave & 1) == 0 : true;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = (e & 3) + 8 * (e >> 2) + 4 * lh;
      if (sums_here && lr == 0) mbar[32 * bi + i] = accs[e] * inv_r;
      if constexpr (K == 64) {
        if ((wave >> 1) == (wave & 1) && lr == i) q2[32 * bi + i] = acc[0][e] * inv_r;
      } else {
#pragma unroll
        for (int bj = 0; bj < KB; ++bj)   // (wave-uniform test: the block index stays a compile-time register index)
          if (bj == wave && lr == i) q2[32 * bi + i] = acc[bj][e] * inv_r;
      }
    }
  }
  __syncthreads();   // mbar / q2 visible; the tiles are dead from here on
  // power-of-two scale that brings the largest second moment (>= the largest |C| entry) to <= 1
  float dmax = 0.f;
  for (int k = lane; k < K; k += 64) dmax = fmaxf(dmax, q2[k]);
  dmax = avs_wave_max(dmax);
  const float cscale = dmax > 0.f ? exp2f(-ceilf(log2f(dmax))) : 1.f;

  // ---- 2. C = a^T a / R - mean mean^T, scaled, as fp16 hi | lo images [K][PC] (symmetric)
  char* chi = lds;
  char* clo = lds + C_BYTES;
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    const int bi = K == 64 ? (wave >> 1) : wave;
    const int bj = K == 64 ? (wave & 1) : b;
    const int l = 32 * bj + lr;
    const float ml = mbar[l];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = 32 * bi + (e & 3) + 8 * (e >> 2) + 4 * lh;
      const float cv = (acc[b][e] * inv_r - mbar[k] * ml) * cscale;
      const _Float16 h = (_Float16)cv;
      const _Float16 lo = (_Float16)(cv - (float)h);
      *reinterpret_cast<_Float16*>(chi + k * PC + l * 2) = h;
      *reinterpret_cast<_Float16*>(clo + k * PC + l * 2) = lo;
    }
  }
  __syncthreads();

  // ---- 3. var_y[n] = w_n^T C w_n, mean_y[n] = w_n . mean: 32 output channels per wave and turn.
  // T^T = C . W^T on the matrix cores (C rows as the A operand, W rows as the B operand), so a lane holds COLUMN
  // n = n0 + lr of T^T, i.e. T[n][l] for l = 32 lt + 8 r + 4 lh + q (register e = 4 r + q): the element-wise product with
  // W[n][l] then reads this lane's OWN weight row (8 bytes of hi halves + 8 of lo per run) and the sum over l is a sum
  // over registers + one shuffle - no transposition, no 32-lane reduction.
  const char* __restrict__ w = p.w;
  const float inv_cs = 1.f / cscale;
  for (int n0 = wave * 32; n0 < p.N; n0 += 128) {
    const int nrow = n0 + lr;   // N is a multiple of 32 (launcher)
    const char* wrow = w + (long long)nrow * p.ldb * 4;
    // this lane's W fragments: W[nrow][16 s + 8 lh .. + 7] as hi | lo
    uint4 wh[K / 16], wl[K / 16];
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      const uint4* src = reinterpret_cast<const uint4*>(wrow + 8 * (2 * s + lh) * 4);
      wh[s] = src[0];
      wl[s] = src[1];
    }
    float my = 0.f;
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      float v[8];
      avs_f16x2_join8(wh[s], wl[s], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) my = fmaf(v[j], mbar[16 * s + 8 * lh + j], my);
    }
    my += __shfl_xor(my, 32, 64);
    float part = 0.f;
#pragma unroll 1   // (unrolled, the scheduler hoists every C fragment read of the turn and the kernel spills)
    for (int lt = 0; lt < KB; ++lt) {
      f32x16 tt;
#pragma unroll
      for (int e = 0; e < 16; ++e) tt[e] = 0.f;
#pragma unroll
      for (int s = 0; s < K / 16; ++s) {
        const int off = (32 * lt + lr) * PC + (16 * s + 8 * lh) * 2;
        const avs_f16x8 ch = __builtin_bit_cast(avs_f16x8, *reinterpret_cast<const uint4*>(chi + off));
        const avs_f16x8 cl = __builtin_bit_cast(avs_f16x8, *reinterpret_cast<const uint4*>(clo + off));
        const avs_f16x8 bh = __builtin_bit_cast(avs_f16x8, wh[s]), bl = __builtin_bit_cast(avs_f16x8, wl[s]);
        tt = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, bh, tt, 0, 0, 0);
        tt = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl, bh, tt, 0, 0, 0);
        tt = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, bl, tt, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // elements 32 lt + 8 r + 4 lh .. + 3 of this lane's weight row: run 4 lt + r, halves 4 lh .. 4 lh + 3
        const char* run = wrow + (4 * lt + r) * 32 + 8 * lh;
        const uint2 h2 = *reinterpret_cast<const uint2*>(run), l2 = *reinterpret_cast<const uint2*>(run + 16);
        float h0, h1, h2f, h3, l0, l1, l2f, l3;
        avs_unpack_f16x2(h2.x, h0, h1);
        avs_unpack_f16x2(h2.y, h2f, h3);
        avs_unpack_f16x2(l2.x, l0, l1);
        avs_unpack_f16x2(l2.y, l2f, l3);
        part = fmaf(tt[4 * r + 0], h0 + l0, part);
        part = fmaf(tt[4 * r + 1], h1 + l1, part);
        part = fmaf(tt[4 * r + 2], h2f + l2f, part);
        part = fmaf(tt[4 * r + 3], h3 + l3, part);
      }
    }
    part += __shfl_xor(part, 32, 64);
    if (lh == 0) {
      const float var = fmaxf(part * inv_cs, 0.f);
      const float sc = p.gamma[nrow] / sqrtf(var + p.eps);
      p.scale[g * p.N + nrow] = sc;
      p.shift[g * p.N + nrow] = p.beta[nrow] - my * sc;
    }
  }
}

extern "C" int avs_bn_gram_affine_f16x2(const void* d_x, int64_t lin_stride, int k, const float* d_in_scale,
                                        const float* d_in_shift, const void* d_w, int64_t ldb, int n,
                                        int64_t rows_per_group, int groups, const float* d_gamma, const float* d_beta,
                                        float eps, float* d_scale, float* d_shift, void* d_a_out, int64_t lda,
                                        avs_stream_t stream) {
  const char* who = "avs_bn_gram_affine_f16x2";
  AVS_REQUIRE(k == 64 || k == 128, AVS_E_UNSUPPORTED, "%s: k = %d (built for 64 and 128 input channels)", who, k);
  AVS_REQUIRE(n > 0 && n % 32 == 0, AVS_E_UNSUPPORTED, "%s: n = %d must be a multiple of 32", who, n);
  AVS_REQUIRE(groups >= 0 && rows_per_group > 0 && rows_per_group < (1ll << 30), AVS_E_SHAPE,
              "%s: groups=%d rows_per_group=%lld", who, groups, (long long)rows_per_group);
  if (groups == 0) return AVS_OK;
  AVS_REQUIRE(d_x && d_w && d_gamma && d_beta && d_scale && d_shift, AVS_E_ARG, "%s: null pointer", who);
  AVS_REQUIRE((d_in_scale == nullptr) == (d_in_shift == nullptr), AVS_E_ARG, "%s: input scale and shift go together", who);
  AVS_REQUIRE(lin_stride % 8 == 0 && lin_stride >= k && ldb % 8 == 0 && ldb >= k, AVS_E_SHAPE,
              "%s: strides must be multiples of 8 slots and at least k", who);
  AVS_REQUIRE(((((uintptr_t)d_x) | ((uintptr_t)d_w)) & 31u) == 0, AVS_E_ALIGN, "%s: x / w must be 32-byte aligned", who);
  if (d_a_out != nullptr) {
    AVS_REQUIRE(d_in_scale != nullptr, AVS_E_ARG, "%s: the finished input is only stored with an input affine", who);
    AVS_REQUIRE(lda % 8 == 0 && lda >= k && (((uintptr_t)d_a_out) & 31u) == 0, AVS_E_ALIGN,
                "%s: a_out must be 32-byte aligned with a row stride that is a multiple of 8 slots and at least k", who);
    AVS_REQUIRE(d_a_out != d_x || lda == lin_stride, AVS_E_ARG, "%s: in place needs lda == lin_stride", who);
  }
  GramH2Params p{};
  p.x = (const char*)d_x;
  p.w = (const char*)d_w;
  p.in_scale = d_in_scale;
  p.in_shift = d_in_shift;
  p.gamma = d_gamma;
  p.beta = d_beta;
  p.scale = d_scale;
  p.shift = d_shift;
  p.lin_stride = lin_stride;
  p.ldb = ldb;
  p.a_out = (char*)d_a_out;
  p.lda = lda;
  p.N = n;
  p.rows_per_group = (int)rows_per_group;
  p.eps = eps;
  const dim3 grid((unsigned)groups), block(256);
  hipStream_t st = (hipStream_t)stream;
  const bool xf = d_in_scale != nullptr;
  if (k == 64) {
    if (xf) hipLaunchKernelGGL((bn_gram_affine_h2_kernel<64, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((bn_gram_affine_h2_kernel<64, false>), grid, block, 0, st, p);
  } else {
    if (xf) hipLaunchKernelGGL((bn_gram_affine_h2_kernel<128, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((bn_gram_affine_h2_kernel<128, false>), grid, block, 0, st, p);
  }
  AVS_CHECK_LAUNCH(who);
  return AVS_OK;
}
