// 1x1 convolution + BatchNorm in batch-statistics mode (+ residual, + ReLU) in ONE kernel, bf16.
//
// Why: with per-micro-batch statistics the normalisation cannot ride in a convolution's epilogue —
// every output of the group must exist before its mean/variance do — so the plain path writes the
// raw convolution, reads it back (statistics), and reads + rewrites it (apply): 8 bytes of HBM per
// element for a 1x1 layer that computes almost nothing.  Here one workgroup owns ALL rows of one
// group for a slab of output channels and walks them twice:
//   pass 1  tiles of 128 rows x BN channels on the matrix cores, only column sums / sums of squares kept
//           (registers -> one shuffle -> LDS across the two row-waves): deterministic, no atomics;
//   pass 2  the same tiles again (operands now come from L2), epilogue = scale/shift from pass 1,
//           residual add, ReLU, bf16, LDS-staged 16-byte stores of the FINAL activations.
// HBM per element: one write (+ one residual read).  The second pass doubles the matrix work of
// layers whose cost is their memory traffic, not their FLOPs.
//
// Tile machinery = igemm.hip's (LDS-DMA staging, 64-byte rows, XOR-swizzled slots, 2 x 2 waves of
// 64 x BN/2 outputs, v_mfma_f32_32x32x16_bf16); rows are linear (output row m reads input row m).
#include "avs_internal.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __attribute__((aligned(16))) unsigned int avs_zero16_cb[4];  // zero source for padded lanes

struct ConvBnParams {
  const char* x;
  const char* w;
  char* y;
  const char* res;
  const float* gamma;
  const float* beta;
  float eps;
  int N, K;
  long long lin_stride, ldb, ldc, ldr;
  int rows_per_group, groups, tiles_n;
  int relu;
  // XF: the input is a RAW convolution output whose BatchNorm (+ ReLU) is applied on the way in:
  // a[m,k] = relu(x[m,k] * in_scale[g,k] + in_shift[g,k]) rounded to bf16 (avs_bn_apply's arithmetic)
  const float* in_scale;
  const float* in_shift;
  // PRE: the output BatchNorm's folded affine [groups, N] is given (avs_bn_gram_affine_bf16 computed it from the
  // input's Gram matrix), so the statistics pass is skipped: one streaming pass over the group
  const float* pre_scale;
  const float* pre_shift;
  // RA: the residual is a RAW convolution output too (the downsample branch): its BatchNorm's folded affine
  // [groups, N] is applied while it is added, y = act(bn(conv) + res * res_scale + res_shift)
  const float* res_scale;
  const float* res_shift;
};

#define AVS_CONVBN_MAX_K 512

#define AVS_GLDS16(src, dst)                                                                        \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),            \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

template <int BN, bool XF, bool PRE = false, bool RA = false>
// (Four workgroups per CU for the streaming form - 40 KB of LDS each fits - need <= 128 VGPRs: the 128-column
//  variants then spill 11-16 registers and the kernel ran at 3.1 instead of 4.2 TB/s; three it is.)
__global__ __launch_bounds__(256, 3) void conv1x1_bn_kernel(ConvBnParams p) {
  constexpr int ES = 2, ROWB = 64;
  constexpr int CE = 16 / ES, BKE = ROWB / ES, CPRR = ROWB / 16, RPP = 256 / CPRR, SH = 2, KS = ROWB / 32;
  constexpr int A_ROWS = 128, NA = A_ROWS / RPP, NB = BN / RPP, NT = BN / 64;
  constexpr int BUF = (A_ROWS + BN) * CPRR;
  constexpr int CT_PITCH = BN * 2 + 16;
  constexpr int CT_SLOTS = (A_ROWS * CT_PITCH) / 16;
  constexpr int LDS_SLOTS = 2 * BUF > CT_SLOTS ? 2 * BUF : CT_SLOTS;
  constexpr int CPRW = BN / 8, RSTEP = 256 / CPRW;

  // (A variant that kept the (tile, step) sequence as one continuous pipeline — next tile's first DMA issued
  // before the current epilogue, staging area not aliased — needed > 168 VGPRs and lost more to the drop
  // from 3 to 2 workgroups per CU than it won: 29.9 k vs 32.1 k frames/s end to end.)
  __shared__ uint4 lds[LDS_SLOTS];
  __shared__ float red[2][BN][2];
  __shared__ __attribute__((aligned(16))) float xf[XF ? 2 : 1][XF ? AVS_CONVBN_MAX_K : 4];  // input scale | shift

  const unsigned nwg = gridDim.x, orig = blockIdx.x;
  const unsigned q = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
  const unsigned wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
  const int tn = wg % p.tiles_n;
  const int g = wg / p.tiles_n;
  const int n0 = tn * BN;
  const long long m_lo = (long long)g * p.rows_per_group;
  const long long m_hi = m_lo + p.rows_per_group;
  const int tiles_m = (p.rows_per_group + A_ROWS - 1) / A_ROWS;

  const char* __restrict__ x = p.x;
  const char* __restrict__ w = p.w;
  const char* zsrc = reinterpret_cast<const char*>(avs_zero16_cb);

  const int t = threadIdx.x;
  const int wave = t >> 6, lane = t & 63;
  const int c = t & (CPRR - 1);
  const int rb = t / CPRR;
  const int cq = c ^ ((rb >> SH) & (CPRR - 1));
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  const char* b_base[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int n = n0 + rb + RPP * i;
    b_base[i] = n < p.N ? w + (long long)n * p.ldb * ES : nullptr;
  }

  const int steps = (p.K + BKE - 1) / BKE;
  if constexpr (XF) {
    for (int i = t; i < p.K; i += 256) {
      xf[0][i] = p.in_scale[(long long)g * p.K + i];
      xf[1][i] = p.in_shift[(long long)g * p.K + i];
    }
    __syncthreads();
  }
  float s1[NT], s2[NT], scale[NT], shift[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) s1[nt] = s2[nt] = scale[nt] = shift[nt] = 0.f;

  if constexpr (PRE) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = n0 + wc * (BN / 2) + nt * 32 + lr;
      scale[nt] = col < p.N ? p.pre_scale[(long long)g * p.N + col] : 0.f;
      shift[nt] = col < p.N ? p.pre_shift[(long long)g * p.N + col] : 0.f;
    }
  }
  for (int pass = PRE ? 1 : 0; pass < 2; ++pass) {
    for (int tm = 0; tm < tiles_m; ++tm) {
      const long long m0 = m_lo + (long long)tm * A_ROWS;
      const char* a_base[NA];
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const long long m = m0 + rb + RPP * i;
        a_base[i] = m < m_hi ? x + m * p.lin_stride * ES : nullptr;
      }
      int kc = cq * CE;
      auto stage = [&](int buf) {
        uint4* abuf = lds + buf * BUF + wave * 64;
        uint4* bbuf = abuf + A_ROWS * CPRR;
        const bool kval = kc < p.K;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const char* src = (kval && a_base[i] != nullptr) ? a_base[i] + (long long)kc * ES : zsrc;
          AVS_GLDS16(src, abuf + 256 * i);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const char* src = (kval && b_base[i] != nullptr) ? b_base[i] + (long long)kc * ES : zsrc;
          AVS_GLDS16(src, bbuf + 256 * i);
        }
        kc += BKE;
      };
      // XF: every thread normalises the A chunks IT staged (its own LDS-DMA has landed after vmcnt(0)), in place,
      // before the barrier that publishes the buffer.  Rows past the group and the K tail stay zero.
      auto transform = [&](int buf, int kfirst) {
        if constexpr (XF) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const int kk = kfirst + cq * CE;
          if (kk < p.K) {
            const float4 sa = *reinterpret_cast<const float4*>(&xf[0][kk]);
            const float4 sb = *reinterpret_cast<const float4*>(&xf[0][kk + 4]);
            const float4 ha = *reinterpret_cast<const float4*>(&xf[1][kk]);
            const float4 hb = *reinterpret_cast<const float4*>(&xf[1][kk + 4]);
            const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
            const float sh[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
#pragma unroll
            for (int i = 0; i < NA; ++i) {
              if (a_base[i] == nullptr) continue;
              uint4* slot = lds + buf * BUF + t + 256 * i;
              const uint4 v = *slot;
              unsigned vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float lo = __uint_as_float(vv[j] << 16) * sc[2 * j] + sh[2 * j];
                float hi = __uint_as_float(vv[j] & 0xffff0000u) * sc[2 * j + 1] + sh[2 * j + 1];
                lo = fmaxf(lo, 0.f);
                hi = fmaxf(hi, 0.f);
                vv[j] = (unsigned)avs_f32_to_bf16(lo) | ((unsigned)avs_f32_to_bf16(hi) << 16);
              }
              *slot = make_uint4(vv[0], vv[1], vv[2], vv[3]);
            }
          }
        }
      };

      f32x16 acc[2][NT];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

      // a group's last tile is usually short (784 rows = 6 tiles + 16 rows): 32-row blocks of this wave that lie past the
      // group's end (wave-uniform) skip their fragment reads, matrix work and conversion - their rows were staged as
      // zeros, are never stored and add nothing to the statistics
      const int live = PRE ? (int)(m_hi - m0) - wr * 64 : 64;   // rows of this wave's 64 that exist (one-pass form only)
      stage(0);
      transform(0, 0);
      __syncthreads();
      for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        const uint4* abuf = lds + buf * BUF;
        const uint4* bbuf = abuf + A_ROWS * CPRR;
        uint4 fa[KS][2], fb[KS][NT];
        if (live > 0) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const int chunk = 2 * ks + lh;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              const int row = wr * 64 + mt * 32 + lr;
              if (mt == 0 || live > 32) fa[ks][mt] = abuf[row * CPRR + (chunk ^ ((row >> SH) & (CPRR - 1)))];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int row = wc * (BN / 2) + nt * 32 + lr;
              fb[ks][nt] = bbuf[row * CPRR + (chunk ^ ((row >> SH) & (CPRR - 1)))];
            }
          }
        }
        if (s + 1 < steps) stage(buf ^ 1);
        if (live > 0) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
              if (mt == 0 || live > 32) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                  acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[ks][mt]),
                                                                        __builtin_bit_cast(bf16x8, fb[ks][nt]),
                                                                        acc[mt][nt], 0, 0, 0);
              }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < steps) transform(buf ^ 1, (s + 1) * BKE);
        __syncthreads();
      }

      if (pass == 0) {
        // rows past the group's end were staged as zeros: they add nothing
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const float v = acc[mt][nt][e];
              s1[nt] += v;
              s2[nt] = fmaf(v, v, s2[nt]);
            }
      } else {
        char* ct = reinterpret_cast<char*>(lds);
        char* cbase = ct + (wr * 64 + 4 * lh) * CT_PITCH + (wc * (BN / 2) + lr) * 2;
        const bool relu_now = p.relu && p.res == nullptr;
        // two accumulator registers (rows r, r + 1 of one column) per conversion; ReLU on the packed pair
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if (live <= mt * 32) continue;   // no row of this 32-row block exists: nothing to convert (never stored)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
              const int roff = mt * 32 + (e & 3) + 8 * (e >> 2);
              unsigned pk = avs_pack_bf16x2(fmaf(acc[mt][nt][e], scale[nt], shift[nt]),
                                            fmaf(acc[mt][nt][e + 1], scale[nt], shift[nt]));
              if (relu_now) pk = avs_relu_bf16x2(pk);
              *reinterpret_cast<unsigned short*>(cbase + roff * CT_PITCH + nt * 64) = (unsigned short)pk;
              *reinterpret_cast<unsigned short*>(cbase + (roff + 1) * CT_PITCH + nt * 64) = (unsigned short)(pk >> 16);
            }
        }
        __syncthreads();
        const int srow = t / CPRW, sch = t - srow * CPRW;
        const int col = n0 + sch * 8;
        if (col < p.N) {  // N is a multiple of 8 (checked on the host)
          constexpr int NIT = A_ROWS / RSTEP;
          float rs[RA ? 8 : 1], rh[RA ? 8 : 1];   // the residual's affine: this thread's 8 channels, the workgroup's group
          if constexpr (RA) {
            const float4 s0 = *reinterpret_cast<const float4*>(p.res_scale + (long long)g * p.N + col);
            const float4 s1 = *reinterpret_cast<const float4*>(p.res_scale + (long long)g * p.N + col + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(p.res_shift + (long long)g * p.N + col);
            const float4 h1 = *reinterpret_cast<const float4*>(p.res_shift + (long long)g * p.N + col + 4);
            rs[0] = s0.x, rs[1] = s0.y, rs[2] = s0.z, rs[3] = s0.w, rs[4] = s1.x, rs[5] = s1.y, rs[6] = s1.z, rs[7] = s1.w;
            rh[0] = h0.x, rh[1] = h0.y, rh[2] = h0.z, rh[3] = h0.w, rh[4] = h1.x, rh[5] = h1.y, rh[6] = h1.z, rh[7] = h1.w;
          }
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const long long row = m0 + srow + it * RSTEP;
            if (row >= m_hi) break;
            uint4 v = *reinterpret_cast<const uint4*>(ct + (srow + it * RSTEP) * CT_PITCH + sch * 16);
            if (p.res != nullptr) {
              unsigned vv[4] = {v.x, v.y, v.z, v.w};
              const uint4 r1 = *reinterpret_cast<const uint4*>(p.res + (row * p.ldr + col) * 2);
              const unsigned rw[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float r0 = __uint_as_float(rw[j] << 16), r1 = __uint_as_float(rw[j] & 0xffff0000u);
                if constexpr (RA) {
                  r0 = r0 * rs[2 * j] + rh[2 * j];
                  r1 = r1 * rs[2 * j + 1] + rh[2 * j + 1];
                }
                const float a0 = __uint_as_float(vv[j] << 16) + r0;
                const float a1 = __uint_as_float(vv[j] & 0xffff0000u) + r1;
                vv[j] = avs_pack_bf16x2(a0, a1);
                if (p.relu) vv[j] = avs_relu_bf16x2(vv[j]);
              }
              v = make_uint4(vv[0], vv[1], vv[2], vv[3]);
            }
            *reinterpret_cast<uint4*>(p.y + (row * p.ldc + col) * 2) = v;
          }
        }
        __syncthreads();  // the staging tile aliases the operand buffers of the next tile
      }
    }

    if (pass == 0) {
      // fold the two lane halves, then the two row-waves through LDS; every lane then holds the group's
      // statistics of its own column(s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        s1[nt] += __shfl_xor(s1[nt], 32, 64);
        s2[nt] += __shfl_xor(s2[nt], 32, 64);
        if (lh == 0) {
          const int lc = wc * (BN / 2) + nt * 32 + lr;
          red[wr][lc][0] = s1[nt];
          red[wr][lc][1] = s2[nt];
        }
      }
      __syncthreads();
      const float inv_n = 1.f / (float)p.rows_per_group;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int lc = wc * (BN / 2) + nt * 32 + lr;
        const int col = n0 + lc;
        const float mean = (red[0][lc][0] + red[1][lc][0]) * inv_n;
        const float var = fmaxf((red[0][lc][1] + red[1][lc][1]) * inv_n - mean * mean, 0.f);
        const float ga = col < p.N ? p.gamma[col] : 0.f;
        const float be = col < p.N ? p.beta[col] : 0.f;
        scale[nt] = ga / sqrtf(var + p.eps);
        shift[nt] = be - mean * scale[nt];
      }
      __syncthreads();
    }
  }
}

// 1 = 64-channel slabs whatever n is (kernel-study build only; the shipped library has no mutable global state)
#ifdef AVS_STUDY
static int g_convbn_narrow = 0;
extern "C" void avs_tune_convbn_narrow(int enabled) { g_convbn_narrow = enabled; }
#else
static constexpr int g_convbn_narrow = 0;
#endif

static int conv1x1_bn_launch(const char* who, const void* d_x, int64_t lin_stride, int k, const void* d_w, int64_t ldb,
                             int n, int64_t rows_per_group, int groups, const float* d_gamma, const float* d_beta,
                             float eps, const void* d_residual, int64_t ldr, int relu, void* d_y, int64_t ldc,
                             const float* d_in_scale, const float* d_in_shift, const float* d_pre_scale,
                             const float* d_pre_shift, const float* d_res_scale, const float* d_res_shift,
                             avs_stream_t stream) {
  AVS_REQUIRE(k > 0 && n > 0 && groups >= 0 && rows_per_group > 0 && rows_per_group < (1ll << 30), AVS_E_SHAPE,
              "%s: k=%d n=%d groups=%d rows_per_group=%lld", who, k, n, groups, (long long)rows_per_group);
  if (groups == 0) return AVS_OK;
  const bool pre = d_pre_scale != nullptr;
  AVS_REQUIRE(d_x && d_w && d_y && (pre ? d_pre_shift != nullptr : (d_gamma && d_beta)), AVS_E_ARG, "%s: null pointer",
              who);
  AVS_REQUIRE(k % 8 == 0 && n % 8 == 0 && lin_stride % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && ldb >= k &&
                  ldc >= n && (!d_residual || (ldr % 8 == 0 && ldr >= n)),
              AVS_E_SHAPE, "%s: k, n and every stride must be multiples of 8 elements (16 bytes)", who);
  AVS_REQUIRE(avs_aligned16(d_x) && avs_aligned16(d_w) && avs_aligned16(d_y) && avs_aligned16(d_residual),
              AVS_E_ALIGN, "%s: operands must be 16-byte aligned", who);
  const bool ra = d_res_scale != nullptr;
  if (ra) {
    AVS_REQUIRE(pre && d_residual && d_res_shift, AVS_E_ARG,
                "%s: a residual affine needs the residual, its shift and the one-pass form", who);
    AVS_REQUIRE(avs_aligned16(d_res_scale) && avs_aligned16(d_res_shift), AVS_E_ALIGN,
                "%s: residual scale / shift must be 16-byte aligned", who);
  }
  const bool xf = d_in_scale != nullptr;
  if (xf) {
    AVS_REQUIRE(d_in_shift, AVS_E_ARG, "%s: null input shift", who);
    AVS_REQUIRE(k <= AVS_CONVBN_MAX_K, AVS_E_UNSUPPORTED, "%s: k=%d > %d", who, k, AVS_CONVBN_MAX_K);
    AVS_REQUIRE(avs_aligned16(d_in_scale) && avs_aligned16(d_in_shift), AVS_E_ALIGN,
                "%s: input scale / shift must be 16-byte aligned", who);
  }
  ConvBnParams p{};
  p.x = (const char*)d_x;
  p.w = (const char*)d_w;
  p.y = (char*)d_y;
  p.res = (const char*)d_residual;
  p.gamma = d_gamma;
  p.beta = d_beta;
  p.eps = eps;
  p.N = n;
  p.K = k;
  p.lin_stride = lin_stride;
  p.ldb = ldb;
  p.ldc = ldc;
  p.ldr = ldr;
  p.rows_per_group = (int)rows_per_group;
  p.groups = groups;
  p.relu = relu;
  p.in_scale = d_in_scale;
  p.in_shift = d_in_shift;
  p.pre_scale = d_pre_scale;
  p.pre_shift = d_pre_shift;
  p.res_scale = d_res_scale;
  p.res_shift = d_res_shift;
  const bool narrow = n <= 64 || g_convbn_narrow;
  const int bn = narrow ? 64 : 128;
  p.tiles_n = (n + bn - 1) / bn;
  const long long total = (long long)groups * p.tiles_n;
  AVS_REQUIRE(total < (1ll << 31), AVS_E_SHAPE, "%s: too many workgroups", who);
  const dim3 grid((unsigned)total), block(256);
  hipStream_t st = (hipStream_t)stream;
#define AVS_CONVBN_LAUNCH(BN_, XF_, PRE_) hipLaunchKernelGGL((conv1x1_bn_kernel<BN_, XF_, PRE_>), grid, block, 0, st, p)
  if (ra) {
#define AVS_CONVBN_LAUNCH_RA(BN_, XF_) hipLaunchKernelGGL((conv1x1_bn_kernel<BN_, XF_, true, true>), grid, block, 0, st, p)
    if (narrow && xf) AVS_CONVBN_LAUNCH_RA(64, true);
    else if (narrow) AVS_CONVBN_LAUNCH_RA(64, false);
    else if (xf) AVS_CONVBN_LAUNCH_RA(128, true);
    else AVS_CONVBN_LAUNCH_RA(128, false);
#undef AVS_CONVBN_LAUNCH_RA
  } else if (pre) {
    if (narrow && xf) AVS_CONVBN_LAUNCH(64, true, true);
    else if (narrow) AVS_CONVBN_LAUNCH(64, false, true);
    else if (xf) AVS_CONVBN_LAUNCH(128, true, true);
    else AVS_CONVBN_LAUNCH(128, false, true);
  } else {
    if (narrow && xf) AVS_CONVBN_LAUNCH(64, true, false);
    else if (narrow) AVS_CONVBN_LAUNCH(64, false, false);
    else if (xf) AVS_CONVBN_LAUNCH(128, true, false);
    else AVS_CONVBN_LAUNCH(128, false, false);
  }
#undef AVS_CONVBN_LAUNCH
  AVS_CHECK_LAUNCH(who);
  return AVS_OK;
}

extern "C" int avs_conv1x1_bn_bf16(const void* d_x, int64_t lin_stride, int k, const void* d_w, int64_t ldb, int n,
                                   int64_t rows_per_group, int groups, const float* d_gamma, const float* d_beta,
                                   float eps, const void* d_residual, int64_t ldr, int relu, void* d_y, int64_t ldc,
                                   avs_stream_t stream) {
  return conv1x1_bn_launch("avs_conv1x1_bn_bf16", d_x, lin_stride, k, d_w, ldb, n, rows_per_group, groups, d_gamma,
                           d_beta, eps, d_residual, ldr, relu, d_y, ldc, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, stream);
}

extern "C" int avs_conv1x1_bn_in_bf16(const void* d_x, int64_t lin_stride, int k, const float* d_in_scale,
                                      const float* d_in_shift, const void* d_w, int64_t ldb, int n,
                                      int64_t rows_per_group, int groups, const float* d_gamma, const float* d_beta,
                                      float eps, const void* d_residual, int64_t ldr, int relu, void* d_y,
                                      int64_t ldc, avs_stream_t stream) {
  AVS_REQUIRE(d_in_scale && d_in_shift, AVS_E_ARG, "avs_conv1x1_bn_in_bf16: null input scale / shift");
  return conv1x1_bn_launch("avs_conv1x1_bn_in_bf16", d_x, lin_stride, k, d_w, ldb, n, rows_per_group, groups, d_gamma,
                           d_beta, eps, d_residual, ldr, relu, d_y, ldc, d_in_scale, d_in_shift, nullptr, nullptr,
                           nullptr, nullptr, stream);
}

extern "C" int avs_conv1x1_affine_bf16(const void* d_x, int64_t lin_stride, int k, const float* d_in_scale,
                                       const float* d_in_shift, const void* d_w, int64_t ldb, int n,
                                       int64_t rows_per_group, int groups, const float* d_scale, const float* d_shift,
                                       const void* d_residual, int64_t ldr, const float* d_res_scale,
                                       const float* d_res_shift, int relu, void* d_y, int64_t ldc,
                                       avs_stream_t stream) {
  AVS_REQUIRE(d_scale && d_shift, AVS_E_ARG, "avs_conv1x1_affine_bf16: null output scale / shift");
  AVS_REQUIRE((d_res_scale == nullptr) == (d_res_shift == nullptr), AVS_E_ARG,
              "avs_conv1x1_affine_bf16: residual scale and shift go together");
  return conv1x1_bn_launch("avs_conv1x1_affine_bf16", d_x, lin_stride, k, d_w, ldb, n, rows_per_group, groups, nullptr,
                           nullptr, 0.f, d_residual, ldr, relu, d_y, ldc, d_in_scale, d_in_shift, d_scale, d_shift,
                           d_res_scale, d_res_shift, stream);
}

// ============================================================================================================
// BatchNorm batch statistics of y = a . w^T WITHOUT computing y: from the Gram matrix of a.
//
//   mean_y[n] = w_n . mean(a),   var_y[n] = w_n^T C w_n,   C = a^T a / R - mean(a) mean(a)^T   (K x K, per group)
//
// For the expanding 1x1 layers of ResNet layers 1-2 (K = 64 / 128 input channels, N = 4K outputs) the Gram matrix
// costs K/N = 1/4 of the convolution's MACs and reads only the narrow input, so the convolution itself becomes ONE
// streaming pass with the affine in its epilogue (avs_conv1x1_affine_bf16) instead of a statistics pass + an output
// pass over the same group (avs_conv1x1_bn_bf16).  One workgroup per group:
//   1. rows staged through registers (16-byte loads; the previous layer's BatchNorm + ReLU applied on the way, as in
//      conv1x1_bn_kernel<XF>) into [64 rows][K] LDS tiles; per-channel sums on the VALU; a^T a on the matrix cores
//      (v_mfma_f32_32x32x16_bf16: both operands are 32 channels x 16 ROWS, gathered from the row-major tile with
//      2-byte LDS reads - the reduction index is the row);
//   2. C in fp32 from the accumulators, split into bf16 hi + lo (16 significant bits) in LDS;
//   3. T = W . C on the matrix cores (2 MFMAs per step: hi, lo; C is symmetric, so its rows are the B operand),
//      var_y[n] = sum_l T[n,l] W[n,l] by a lane reduction, mean_y from the same W fragments, then the folded affine.
// Deterministic (fixed reduction orders, no atomics).
struct GramParams {
  const char* x;
  const char* w;
  const float* in_scale;
  const float* in_shift;
  const float* gamma;
  const float* beta;
  float* scale;
  float* shift;
  char* a_out;   // XF only, may be x itself: the transformed input a = bf16(relu(x * in_scale + in_shift)) is stored here
  long long lin_stride, ldb, lda;
  int N, rows_per_group;
  float eps;
};

template <int K, bool XF>
__global__ __launch_bounds__(256, 2) void bn_gram_affine_kernel(GramParams p) {
  static_assert(K == 64 || K == 128, "input widths of the layer-1 / layer-2 expanding convolutions");
  constexpr int CPR = K / 8;              // 16-byte chunks per row
  constexpr int TR = 64;                  // rows per LDS tile
  constexpr int RPT = 256 / CPR;          // rows one pass of the 256 threads stages
  constexpr int NP = TR / RPT;            // chunks per thread per tile
  constexpr int PITCH = K * 2 + 16;       // bytes per LDS row (tiles and C images)
  constexpr int KB = K / 32;              // 32-channel blocks per side
  constexpr int NBLK = (KB * KB) / 4;     // Gram blocks per wave: 1 (K = 64) or 4 (K = 128: one block row)
  constexpr int TILE_BYTES = TR * PITCH;
  constexpr int C_BYTES = K * PITCH;      // one bf16 K x K image
  constexpr int MAIN_BYTES = 2 * C_BYTES > 2 * TILE_BYTES + RPT * K * 4 ? 2 * C_BYTES : 2 * TILE_BYTES + RPT * K * 4;
  __shared__ __attribute__((aligned(16))) char lds[MAIN_BYTES];
  __shared__ float mbar[K];
  __shared__ float qbuf[4][32];

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const int c = t % CPR, r0t = t / CPR;
  const long long g = blockIdx.x;
  const int R = p.rows_per_group;
  const char* __restrict__ xg = p.x + g * R * p.lin_stride * 2;

  float xsc[8], xsh[8];
  if constexpr (XF) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xsc[j] = p.in_scale[g * K + 8 * c + j];
      xsh[j] = p.in_shift[g * K + 8 * c + j];
    }
  }
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;

  uint4 regs[NP];
  auto gload = [&](int tile) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int row = tile * TR + r0t + RPT * q;
      regs[q] = row < R ? *reinterpret_cast<const uint4*>(xg + (long long)row * p.lin_stride * 2 + c * 16)
                        : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto xform_store = [&](int buf, int tile) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int row = tile * TR + r0t + RPT * q;
      uint4 v = regs[q];
      if (row < R) {
        unsigned vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float lo = __uint_as_float(vv[j] << 16), hi = __uint_as_float(vv[j] & 0xffff0000u);
          if constexpr (XF) {   // conv1x1_bn_kernel<XF>'s arithmetic: a = bf16(relu(x * scale + shift))
            lo = fmaxf(lo * xsc[2 * j] + xsh[2 * j], 0.f);
            hi = fmaxf(hi * xsc[2 * j + 1] + xsh[2 * j + 1], 0.f);
            const unsigned short blo = avs_f32_to_bf16(lo), bhi = avs_f32_to_bf16(hi);
            vv[j] = (unsigned)blo | ((unsigned)bhi << 16);
            lo = avs_bf16_to_f32(blo);
            hi = avs_bf16_to_f32(bhi);
          }
          cs[2 * j] += lo;
          cs[2 * j + 1] += hi;
        }
        v = make_uint4(vv[0], vv[1], vv[2], vv[3]);
        if constexpr (XF) {
          // every chunk is read by exactly one thread before it is written: in place (a_out == x) is safe
          if (p.a_out != nullptr)
            *reinterpret_cast<uint4*>(p.a_out + ((g * R + row) * p.lda) * 2 + c * 16) = v;
        }
      }
      *reinterpret_cast<uint4*>(lds + buf * TILE_BYTES + (r0t + RPT * q) * PITCH + c * 16) = v;
    }
  };
  // 32 channels (block b) x 16 rows (step s) of a tile as an MFMA operand: lane (lr, lh) takes channel 32 b + lr of
  // rows 16 s + 8 lh .. + 7
  auto gather = [&](const char* tile, int b, int s) -> bf16x8 {
    const char* base = tile + (16 * s + 8 * lh) * PITCH + (32 * b + lr) * 2;
    unsigned w4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned lo = *reinterpret_cast<const unsigned short*>(base + (2 * j) * PITCH);
      const unsigned hi = *reinterpret_cast<const unsigned short*>(base + (2 * j + 1) * PITCH);
      w4[j] = lo | (hi << 16);
    }
    return __builtin_bit_cast(bf16x8, make_uint4(w4[0], w4[1], w4[2], w4[3]));
  };

  f32x16 acc[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;

  // ---- 1. a^T a and the channel sums over the group's rows
  const int tiles = (R + TR - 1) / TR;
  gload(0);
  xform_store(0, 0);
  __syncthreads();
  for (int tile = 0; tile < tiles; ++tile) {
    const int buf = tile & 1;
    if (tile + 1 < tiles) gload(tile + 1);   // in flight during the matrix work below
    const char* tb = lds + buf * TILE_BYTES;
#pragma unroll
    for (int s = 0; s < TR / 16; ++s) {
      if constexpr (K == 64) {
        const bf16x8 a = gather(tb, wave >> 1, s);
        const bf16x8 b = gather(tb, wave & 1, s);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
      } else {
        const bf16x8 a = gather(tb, wave, s);
#pragma unroll
        for (int bj = 0; bj < KB; ++bj) {
          const bf16x8 b = gather(tb, bj, s);
          acc[bj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[bj], 0, 0, 0);
        }
      }
    }
    if (tile + 1 < tiles) xform_store(buf ^ 1, tile + 1);
    __syncthreads();
  }

  // channel sums: the RPT threads that share a chunk column, added in a fixed order
  float* scratch = reinterpret_cast<float*>(lds + 2 * TILE_BYTES);   // [RPT][K]
#pragma unroll
  for (int j = 0; j < 8; ++j) scratch[r0t * K + 8 * c + j] = cs[j];
  __syncthreads();
  const float inv_r = 1.f / (float)R;
  if (t < K) {
    float m = 0.f;
    for (int r = 0; r < RPT; ++r) m += scratch[r * K + t];
    mbar[t] = m * inv_r;
  }
  __syncthreads();   // mbar visible; tiles and scratch are dead from here on

  // ---- 2. C = a^T a / R - mean mean^T, as bf16 hi + lo images [K][PITCH] (symmetric)
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
      const float cv = acc[b][e] * inv_r - mbar[k] * ml;
      const unsigned short h = avs_f32_to_bf16(cv);
      const unsigned short lo = avs_f32_to_bf16(cv - avs_bf16_to_f32(h));
      *reinterpret_cast<unsigned short*>(chi + k * PITCH + l * 2) = h;
      *reinterpret_cast<unsigned short*>(clo + k * PITCH + l * 2) = lo;
    }
  }
  __syncthreads();

  // ---- 3. var_y[n] = w_n^T C w_n, mean_y[n] = w_n . mean: 32 output channels per wave and turn
  const char* __restrict__ w = p.w;
  for (int n0 = wave * 32; n0 < p.N; n0 += 128) {
    const int nrow = n0 + lr;   // N is a multiple of 32 (launcher)
    uint4 wf[K / 16];           // this lane's W fragments: W[nrow][16 s + 8 lh .. + 7]
#pragma unroll
    for (int s = 0; s < K / 16; ++s)
      wf[s] = *reinterpret_cast<const uint4*>(w + ((long long)nrow * p.ldb + 16 * s + 8 * lh) * 2);
    float my = 0.f;
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      const unsigned vv[4] = {wf[s].x, wf[s].y, wf[s].z, wf[s].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        my = fmaf(__uint_as_float(vv[j] << 16), mbar[16 * s + 8 * lh + 2 * j], my);
        my = fmaf(__uint_as_float(vv[j] & 0xffff0000u), mbar[16 * s + 8 * lh + 2 * j + 1], my);
      }
    }
    my += __shfl_xor(my, 32, 64);
    float part[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) part[e] = 0.f;
#pragma unroll
    for (int lt = 0; lt < KB; ++lt) {
      f32x16 tt;
#pragma unroll
      for (int e = 0; e < 16; ++e) tt[e] = 0.f;
#pragma unroll
      for (int s = 0; s < K / 16; ++s) {
        const int off = (32 * lt + lr) * PITCH + (16 * s + 8 * lh) * 2;
        const uint4 bh = *reinterpret_cast<const uint4*>(chi + off);
        const uint4 bl = *reinterpret_cast<const uint4*>(clo + off);
        tt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[s]), __builtin_bit_cast(bf16x8, bh),
                                                     tt, 0, 0, 0);
        tt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[s]), __builtin_bit_cast(bf16x8, bl),
                                                     tt, 0, 0, 0);
      }
      // tt[e] = T[n0 + rowoff(e) + 4 lh][32 lt + lr]; times W at the same place
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const unsigned short wb = *reinterpret_cast<const unsigned short*>(w + ((long long)n * p.ldb + 32 * lt + lr) * 2);
        part[e] = fmaf(tt[e], avs_bf16_to_f32(wb), part[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = part[e];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lr == 0) qbuf[wave][(e & 3) + 8 * (e >> 2) + 4 * lh] = v;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lh == 0) {
      const float var = fmaxf(qbuf[wave][lr], 0.f);
      const float sc = p.gamma[nrow] / sqrtf(var + p.eps);
      p.scale[g * p.N + nrow] = sc;
      p.shift[g * p.N + nrow] = p.beta[nrow] - my * sc;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

extern "C" int avs_bn_gram_affine_bf16(const void* d_x, int64_t lin_stride, int k, const float* d_in_scale,
                                       const float* d_in_shift, const void* d_w, int64_t ldb, int n,
                                       int64_t rows_per_group, int groups, const float* d_gamma, const float* d_beta,
                                       float eps, float* d_scale, float* d_shift, void* d_a_out, int64_t lda,
                                       avs_stream_t stream) {
  const char* who = "avs_bn_gram_affine_bf16";
  AVS_REQUIRE(k == 64 || k == 128, AVS_E_UNSUPPORTED, "%s: k = %d (built for 64 and 128 input channels)", who, k);
  AVS_REQUIRE(n > 0 && n % 32 == 0, AVS_E_UNSUPPORTED, "%s: n = %d must be a multiple of 32", who, n);
  AVS_REQUIRE(groups >= 0 && rows_per_group > 0 && rows_per_group < (1ll << 30), AVS_E_SHAPE,
              "%s: groups=%d rows_per_group=%lld", who, groups, (long long)rows_per_group);
  if (groups == 0) return AVS_OK;
  AVS_REQUIRE(d_x && d_w && d_gamma && d_beta && d_scale && d_shift, AVS_E_ARG, "%s: null pointer", who);
  AVS_REQUIRE((d_in_scale == nullptr) == (d_in_shift == nullptr), AVS_E_ARG, "%s: input scale and shift go together", who);
  AVS_REQUIRE(lin_stride % 8 == 0 && lin_stride >= k && ldb % 8 == 0 && ldb >= k, AVS_E_SHAPE,
              "%s: strides must be multiples of 8 elements (16 bytes) and at least k", who);
  AVS_REQUIRE(avs_aligned16(d_x) && avs_aligned16(d_w), AVS_E_ALIGN, "%s: x / w must be 16-byte aligned", who);
  if (d_a_out != nullptr) {
    AVS_REQUIRE(d_in_scale != nullptr, AVS_E_ARG, "%s: the transformed input is only stored with an input affine", who);
    AVS_REQUIRE(lda % 8 == 0 && lda >= k && avs_aligned16(d_a_out), AVS_E_ALIGN,
                "%s: a_out must be 16-byte aligned with a row stride that is a multiple of 8 elements and at least k", who);
    AVS_REQUIRE(d_a_out != d_x || lda == lin_stride, AVS_E_ARG, "%s: in place needs lda == lin_stride", who);
  }
  GramParams p{};
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
    if (xf) hipLaunchKernelGGL((bn_gram_affine_kernel<64, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((bn_gram_affine_kernel<64, false>), grid, block, 0, st, p);
  } else {
    if (xf) hipLaunchKernelGGL((bn_gram_affine_kernel<128, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((bn_gram_affine_kernel<128, false>), grid, block, 0, st, p);
  }
  AVS_CHECK_LAUNCH(who);
  return AVS_OK;
}
