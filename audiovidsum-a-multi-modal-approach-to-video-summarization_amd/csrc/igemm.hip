// Implicit-GEMM convolution / batched NT GEMM for gfx950 (CDNA4).
//
// One kernel serves every dense contraction of the hot path (SURVEY K3, K6,
// K16-K20): the reduction index k runs over (kh, kw, ci) of an NHWC activation
// tensor, so a 1x1/stride-1 convolution, an nn.Linear and a strided-batched
// GEMM are the same code with different address parameters.
//
// Tile: 128 x BN outputs per 256-thread workgroup (4 waves as 2 x 2), 64 BYTES
// of reduction per LDS row and step (16 fp32 / 32 bf16), two LDS buffers with a
// register-staged prefetch (global loads of step s+1 are in flight while step s
// is on the matrix cores).  An LDS row is four 16-byte chunks, XOR-swizzled by
// (row >> 2) & 3 so that the ds_read_b128 fragment reads are conflict-free.
//
//   bf16: v_mfma_f32_32x32x16_bf16 — lane (r = l & 31, h = l >> 5) feeds the
//         8 bf16 of chunk 2*ks + h.
//   fp32: v_mfma_f32_32x32x2_f32 x 4 — the same 16-byte chunk holds 4 floats,
//         used as four K=2 steps; the k order inside a tile is permuted
//         identically for A and B, which only reorders an exact-fmaf sum.
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
};

template <int ES, int BN, bool ACC64>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmParams p) {
  static_assert(!ACC64 || (ES == 4 && BN == 64), "fp64 slice accumulation: fp32 operands, narrow tile only");
  constexpr int CE = 16 / ES;   // elements per 16-byte chunk
  constexpr int BKE = 64 / ES;  // elements per LDS row
  constexpr int NB = BN / 64;   // B rows staged per thread
  constexpr int NT = BN / 64;   // 32-wide column tiles per wave
  constexpr int A_ROWS = 128;

  __shared__ uint4 lds[2][(A_ROWS + BN) * 4];

  // XCD-aware, bijective block remap: blocks that share an XCD (orig % 8)
  // take consecutive tiles, so neighbouring column tiles reuse A rows in L2.
  const unsigned nwg = gridDim.x, orig = blockIdx.x;
  const unsigned q = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
  const unsigned wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
  const int tn = wg % p.tiles_n;
  const int tm = wg / p.tiles_n;
  const int m0 = tm * A_ROWS;
  const int n0 = tn * BN;

  const long long z = blockIdx.z;
  const char* __restrict__ x = p.x + z * p.sA * ES;
  const char* __restrict__ w = p.w + z * p.sB * ES;
  char* __restrict__ y = p.y + z * p.sC * ES;
  const float* __restrict__ bias = p.bias ? p.bias + z * p.sBias : nullptr;

  const int t = threadIdx.x;
  const int c = t & 3;
  const int r = t >> 2;

  long long a_off[2];
  int hi0[2], wi0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + r + 64 * i;
    if (m < p.M) {
      const int n = m / p.HoWo;
      const int rem = m - n * p.HoWo;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      a_off[i] = (long long)n * p.x_img_stride;
      hi0[i] = ho * p.sh - p.ph;
      wi0[i] = wo * p.sw - p.pw;
    } else {
      a_off[i] = 0;
      hi0[i] = -(1 << 29);
      wi0[i] = 0;
    }
  }
  long long b_off[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int n = n0 + r + 64 * i;
    b_off[i] = n < p.N ? (long long)n * p.ldb : -1;
  }

  int kc = c * CE;
  int kk = kc / p.cin;
  int ci = kc - kk * p.cin;
  int kh = kk / p.KW;
  int kw = kk - kh * p.KW;

  uint4 va[2], vb[NB];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);

  auto gload = [&]() {
    const bool kval = kc < p.K;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int hi = hi0[i] + kh, wi = wi0[i] + kw;
      const bool ok = kval && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      va[i] = zero4;
      if (ok)
        va[i] = *reinterpret_cast<const uint4*>(
            x + (a_off[i] + hi * p.x_row_stride + wi * p.x_px_stride + ci) * ES);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      vb[i] = zero4;
      if (kval && b_off[i] >= 0)
        vb[i] = *reinterpret_cast<const uint4*>(w + (b_off[i] + kc) * ES);
    }
    kc += BKE;
    ci += BKE;
    while (ci >= p.cin) {
      ci -= p.cin;
      if (++kw == p.KW) {
        kw = 0;
        ++kh;
      }
    }
  };
  auto lwrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r + 64 * i;
      lds[buf][row * 4 + (c ^ ((row >> 2) & 3))] = va[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = r + 64 * i;
      lds[buf][(A_ROWS + row) * 4 + (c ^ ((row >> 2) & 3))] = vb[i];
    }
  };

  const int wave = t >> 6, lane = t & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ACC64: every 16-element reduction slice is an exact-fmaf MFMA chain in fp32; the slices are summed in
  // fp64, so the rounding error scales with the size of a SLICE sum, not of the whole running sum.
  double acc64[ACC64 ? 2 : 1][ACC64 ? NT : 1][ACC64 ? 16 : 1];
  if constexpr (ACC64) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc64[i][j][e] = 0.0;
  }

  const int steps = (p.K + BKE - 1) / BKE;
  gload();
  lwrite(0);
  __syncthreads();

  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    if (s + 1 < steps) gload();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int chunk = 2 * ks + lh;
      uint4 fa[2], fb[NT];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = wr * 64 + mt * 32 + lr;
        fa[mt] = lds[buf][row * 4 + (chunk ^ ((row >> 2) & 3))];
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int row = wc * (BN / 2) + nt * 32 + lr;
        fb[nt] = lds[buf][(A_ROWS + row) * 4 + (chunk ^ ((row >> 2) & 3))];
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (ES == 2) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, fa[mt]), __builtin_bit_cast(bf16x8, fb[nt]),
                acc[mt][nt], 0, 0, 0);
          } else {
            const float4 a4 = __builtin_bit_cast(float4, fa[mt]);
            const float4 b4 = __builtin_bit_cast(float4, fb[nt]);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[mt][nt], 0, 0, 0);
          }
        }
    }
    if constexpr (ACC64) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            acc64[i][j][e] += (double)acc[i][j][e];
            acc[i][j][e] = 0.f;
          }
    }
    if (s + 1 < steps) lwrite(buf ^ 1);
    __syncthreads();
  }

  // Epilogue straight from the accumulators: register e of a 32x32 tile is
  // row (e&3) + 8*(e>>2) + 4*lh, column lr — lanes 0..31 of one register store
  // 32 consecutive output channels of one row.
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = n0 + wc * (BN / 2) + nt * 32 + lr;
      if (col >= p.N) continue;
      const float bcol = (p.bias_mode == AVS_BIAS_COL) ? bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wr * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row >= p.M) continue;
        float v;
        if constexpr (ACC64)
          v = (float)(acc64[mt][nt][e] * (double)p.alpha) + bcol;
        else
          v = acc[mt][nt][e] * p.alpha + bcol;
        if (p.bias_mode == AVS_BIAS_ROW) v += bias[row];
        if (p.act == AVS_ACT_RELU) v = fmaxf(v, 0.f);
        char* dst = y + ((long long)row * p.ldc + col) * ES;
        if constexpr (ES == 2)
          *reinterpret_cast<unsigned short*>(dst) = avs_f32_to_bf16(v);
        else
          *reinterpret_cast<float*>(dst) = v;
      }
    }
}

static int igemm_launch(int dtype, IgemmParams& p, int batch, hipStream_t stream, const char* who) {
  const int es = dtype == AVS_BF16 ? 2 : 4;
  const int ce = 16 / es;
  AVS_REQUIRE(dtype == AVS_F32 || dtype == AVS_BF16 || dtype == AVS_F32_ACC64, AVS_E_ARG, "%s: bad dtype %d", who,
              dtype);
  AVS_REQUIRE(p.M >= 0 && p.N > 0 && p.K > 0 && batch > 0, AVS_E_SHAPE, "%s: bad sizes M=%d N=%d K=%d batch=%d", who,
              p.M, p.N, p.K, batch);
  if (p.M == 0) return AVS_OK;
  AVS_REQUIRE(p.x && p.w && p.y, AVS_E_ARG, "%s: null operand", who);
  AVS_REQUIRE(p.bias_mode == AVS_BIAS_NONE || p.bias, AVS_E_ARG, "%s: bias mode %d without bias", who, p.bias_mode);
  AVS_REQUIRE(p.cin % ce == 0 && p.K % ce == 0, AVS_E_SHAPE,
              "%s: cin=%d / K=%d must be multiples of %d elements (16 bytes)", who, p.cin, p.K, ce);
  AVS_REQUIRE(avs_aligned16(p.x) && avs_aligned16(p.w), AVS_E_ALIGN, "%s: x / w must be 16-byte aligned", who);
  AVS_REQUIRE((p.x_img_stride * es) % 16 == 0 && (p.x_row_stride * es) % 16 == 0 && (p.x_px_stride * es) % 16 == 0 &&
                  (p.ldb * es) % 16 == 0 && (p.sA * es) % 16 == 0 && (p.sB * es) % 16 == 0,
              AVS_E_ALIGN, "%s: strides must be multiples of 16 bytes", who);
  AVS_REQUIRE(p.ldc >= p.N, AVS_E_SHAPE, "%s: output row stride %lld < N=%d", who, p.ldc, p.N);
  AVS_REQUIRE(batch <= 65535, AVS_E_SHAPE, "%s: batch %d > 65535", who, batch);

  const bool narrow = p.N <= 64 || dtype == AVS_F32_ACC64;
  const int bn = narrow ? 64 : 128;
  p.tiles_n = (p.N + bn - 1) / bn;
  const long long tiles_m = ((long long)p.M + 127) / 128;
  const long long total = tiles_m * p.tiles_n;
  AVS_REQUIRE(total < (1ll << 31), AVS_E_SHAPE, "%s: too many tiles", who);
  dim3 grid((unsigned)total, 1, (unsigned)batch), block(256);
  if (dtype == AVS_BF16) {
    if (narrow)
      hipLaunchKernelGGL((igemm_kernel<2, 64, false>), grid, block, 0, stream, p);
    else
      hipLaunchKernelGGL((igemm_kernel<2, 128, false>), grid, block, 0, stream, p);
  } else if (dtype == AVS_F32_ACC64) {
    hipLaunchKernelGGL((igemm_kernel<4, 64, true>), grid, block, 0, stream, p);
  } else {
    if (narrow)
      hipLaunchKernelGGL((igemm_kernel<4, 64, false>), grid, block, 0, stream, p);
    else
      hipLaunchKernelGGL((igemm_kernel<4, 128, false>), grid, block, 0, stream, p);
  }
  AVS_CHECK_LAUNCH(who);
  return AVS_OK;
}

extern "C" int avs_conv2d_nhwc(const avs_conv_desc* d, const void* d_x, const void* d_w, const float* d_bias,
                               void* d_y, avs_stream_t stream) {
  AVS_REQUIRE(d != nullptr, AVS_E_ARG, "avs_conv2d_nhwc: null descriptor");
  AVS_REQUIRE(d->n >= 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->kh > 0 && d->kw > 0 && d->sh > 0 && d->sw > 0 &&
                  d->ph >= 0 && d->pw >= 0 && d->ho > 0 && d->wo > 0 && d->cout > 0,
              AVS_E_SHAPE, "avs_conv2d_nhwc: non-positive extent");
  // every output tap row/pixel must stay inside the padded input
  AVS_REQUIRE((d->ho - 1) * d->sh - d->ph + d->kh - 1 < d->h + d->ph &&
                  (d->wo - 1) * d->sw - d->pw + d->kw - 1 < d->w + d->pw,
              AVS_E_SHAPE, "avs_conv2d_nhwc: output extent %dx%d exceeds what input %dx%d allows", d->ho, d->wo, d->h,
              d->w);
  const long long rows = (long long)d->n * d->ho * d->wo;
  AVS_REQUIRE(rows < (1ll << 31), AVS_E_SHAPE, "avs_conv2d_nhwc: %lld output pixels exceed int32", rows);
  IgemmParams p{};
  p.x = (const char*)d_x;
  p.w = (const char*)d_w;
  p.y = (char*)d_y;
  p.bias = d_bias;
  p.M = (int)rows;
  p.N = d->cout;
  p.K = d->kh * d->kw * d->cin;
  p.HoWo = d->ho * d->wo;
  p.Wo = d->wo;
  p.H = d->h;
  p.W = d->w;
  p.cin = d->cin;
  p.KW = d->kw;
  p.sh = d->sh;
  p.sw = d->sw;
  p.ph = d->ph;
  p.pw = d->pw;
  p.x_img_stride = d->x_img_stride;
  p.x_row_stride = d->x_row_stride;
  p.x_px_stride = d->x_px_stride;
  p.ldb = d->w_row_stride;
  p.ldc = d->y_px_stride;
  p.alpha = d->alpha;
  p.act = d->act;
  p.bias_mode = d_bias ? AVS_BIAS_COL : AVS_BIAS_NONE;
  AVS_REQUIRE(p.ldb >= p.K, AVS_E_SHAPE, "avs_conv2d_nhwc: w_row_stride %lld < kh*kw*cin=%d", p.ldb, p.K);
  return igemm_launch(d->dtype, p, 1, (hipStream_t)stream, "avs_conv2d_nhwc");
}

extern "C" int avs_gemm_nt(int dtype, int m, int n, int k, const void* d_a, int64_t lda, int64_t stride_a,
                           const void* d_b, int64_t ldb, int64_t stride_b, void* d_c, int64_t ldc, int64_t stride_c,
                           const float* d_bias, int bias_mode, int64_t stride_bias, float alpha, int act, int batch,
                           avs_stream_t stream) {
  IgemmParams p{};
  p.x = (const char*)d_a;
  p.w = (const char*)d_b;
  p.y = (char*)d_c;
  p.bias = d_bias;
  p.M = m;
  p.N = n;
  p.K = k;
  p.HoWo = 1;
  p.Wo = 1;
  p.H = 1;
  p.W = 1;
  p.cin = k;
  p.KW = 1;
  p.sh = p.sw = 1;
  p.ph = p.pw = 0;
  p.x_img_stride = lda;
  p.x_row_stride = 0;
  p.x_px_stride = 0;
  p.ldb = ldb;
  p.ldc = ldc;
  p.sA = stride_a;
  p.sB = stride_b;
  p.sC = stride_c;
  p.sBias = stride_bias;
  p.alpha = alpha;
  p.act = act;
  p.bias_mode = bias_mode;
  AVS_REQUIRE(bias_mode >= AVS_BIAS_NONE && bias_mode <= AVS_BIAS_ROW, AVS_E_ARG, "avs_gemm_nt: bad bias mode %d",
              bias_mode);
  AVS_REQUIRE(lda >= 0 && ldb >= k, AVS_E_SHAPE, "avs_gemm_nt: lda=%lld ldb=%lld k=%d", (long long)lda,
              (long long)ldb, k);
  return igemm_launch(dtype, p, batch, (hipStream_t)stream, "avs_gemm_nt");
}
