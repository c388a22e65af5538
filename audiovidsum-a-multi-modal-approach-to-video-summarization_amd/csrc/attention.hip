// Fused multi-head self-attention core (models/attention.py:21-24), fp32, flash style: the [T, T] score
// matrix never exists in memory (the reference materialises [B, H, T, T]: 400 MB at T = 5000).
//
//   ctx[b, q, h*D + :] = softmax_k( Q[b,q,h,:] . K[b,k,h,:] / sqrt(D) ) . V[b,k,h,:]
//
// Workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries and sweeps the keys in
// tiles of 32 that the workgroup stages in LDS (K padded to a pitch of D+1 so that the per-lane column reads
// are conflict-free; V row-major).  Everything is kept TRANSPOSED so that a query lives on a lane:
//   S^T = K . Q^T     v_mfma_f32_32x32x2_f32, A = K tile from LDS, B = the wave's Q rows held in registers;
//                     result: key rows in the 16 accumulator registers, query column on the lane
//   row softmax       online (running max / sum per query = per lane): 16 registers + ONE wave shuffle (xor 32)
//                     fold the two lane halves — no LDS, no cross-lane loops
//   O^T += V^T . P    the S^T accumulator is already the MFMA B operand of this product (it sums over the
//                     accumulator's ROW index); A = V tile from LDS read along d (lane = d, conflict-free).
//                     Register e of lane half h is key (e&3) + 8*(e>>2) + 4*h of the tile, and the A operand
//                     reads exactly that key, so no data moves between the two products.
// fp32 MFMA is an exact fmaf chain (no TF32-like path on gfx950), so the result differs from the CPU only by
// summation order and the online-softmax rescaling.
#include "avs_internal.h"
#include <math.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int D, int NW>
__global__ __launch_bounds__(64 * NW, 1) void flash_mhsa_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, long long ld, int T,
                                                            float sqrt_d, float* __restrict__ ctx, long long ldo) {
  constexpr int KP = D + 1;          // K tile pitch (floats)
  constexpr int DT = D / 32;         // 32-wide d tiles of the output
  constexpr int OP = 33;             // output staging pitch
  __shared__ float ks[32 * KP];
  __shared__ float vs[32 * D];
  __shared__ float os[NW][32 * OP];

  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int lj = lane & 31, lh = lane >> 5;
  const int h = blockIdx.y;
  const long long boff = (long long)blockIdx.z * T;   // first row of this batch entry
  const int q0 = blockIdx.x * (32 * NW) + wave * 32;
  const int qrow = q0 + lj;
  const bool q_ok = qrow < T;

  // this lane's query row, the half of its D entries that its lane half feeds to the MFMA (k = 2s + lh)
  float qreg[D / 2];
  {
    const float* qp = q + (boff + (q_ok ? qrow : 0)) * ld + h * D;
#pragma unroll
    for (int s = 0; s < D / 2; ++s) qreg[s] = q_ok ? qp[2 * s + lh] : 0.f;
  }
  f32x16 oacc[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int k0 = 0; k0 < T; k0 += 32) {
    __syncthreads();  // the previous tile is no longer read
    // staging: float4 per thread, D/4 of them per key row, 32 rows
#pragma unroll 4
    for (int idx = t; idx < 32 * (D / 4); idx += 64 * NW) {
      const int lrow = idx / (D / 4), c = 4 * (idx - lrow * (D / 4));
      const int krow = k0 + lrow;
      const bool ok = krow < T;
      const long long roff = (boff + (ok ? krow : 0)) * ld + h * D + c;
      const float4 kv = ok ? *reinterpret_cast<const float4*>(k + roff) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 vv = ok ? *reinterpret_cast<const float4*>(v + roff) : make_float4(0.f, 0.f, 0.f, 0.f);
      float* kd = ks + lrow * KP + c;
      kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
      *reinterpret_cast<float4*>(vs + lrow * D + c) = vv;
    }
    __syncthreads();

    // S^T[key][query] = sum_d K[key][d] * Q[query][d]
    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
    const float* krow_p = ks + lj * KP + lh;
#pragma unroll 16
    for (int s = 0; s < D / 2; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(krow_p[2 * s], qreg[s], sacc, 0, 0, 0);

    // scale, mask the keys past T, online softmax per query (= per lane; the two lane halves hold different keys)
    float mt = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      const float sv = key < T ? sacc[e] / sqrt_d : -INFINITY;
      sacc[e] = sv;
      mt = fmaxf(mt, sv);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);           // finite: every tile holds at least one valid key
    const float corr = expf(m_run - m_new);         // exp(-inf) = 0 on the first tile
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pe = expf(sacc[e] - m_new);       // masked keys: exp(-inf) = 0
      sacc[e] = pe;
      ls += pe;
    }
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * corr + ls;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[i][e] *= corr;

    // O^T[d][query] += sum_key V[key][d] * P[key][query]; register e = key (e&3) + 8*(e>>2) + 4*lh
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float* vrow = vs + ((e & 3) + 8 * (e >> 2) + 4 * lh) * D + lj;
#pragma unroll
      for (int i = 0; i < DT; ++i)
        oacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * i], sacc[e], oacc[i], 0, 0, 0);
    }
  }

  // normalise and store: O^T tile (d rows in registers, query on the lane) -> LDS [query][d] -> 128-byte rows
  const float inv_l = 1.f / l_run;
  float* osw = os[wave];
#pragma unroll
  for (int i = 0; i < DT; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int dd = (e & 3) + 8 * (e >> 2) + 4 * lh;
      osw[lj * OP + dd] = oacc[i][e] * inv_l;
    }
    // a wave only touches its own staging slice: wave-level ordering is enough
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = (lane >> 3) + 8 * it, c = (lane & 7) * 4;
      if (q0 + r < T) {
        float4 o4;
        o4.x = osw[r * OP + c];
        o4.y = osw[r * OP + c + 1];
        o4.z = osw[r * OP + c + 2];
        o4.w = osw[r * OP + c + 3];
        *reinterpret_cast<float4*>(ctx + (boff + q0 + r) * ldo + h * D + i * 32 + c) = o4;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same fused core on the fp16 matrix cores, split precision (AVS_F16X2 operands): Q, K, V come in as fp16 hi | lo
// runs (avs_f16x2_pack_f32 of the fp32 projections), every product is hi*hi + lo*hi + hi*lo (2^-21 relative, fp32
// accumulation), the probabilities are split once per tile in registers.  K tile: rows of f16x2 as they are
// ([key][D slots], pitch D * 4 + 16 bytes); V tile: TRANSPOSED planes Vt_hi / Vt_lo [d][32 key slots] (2-byte writes
// when staging, a lane per key so that a wave's writes are contiguous): the reduction index of O^T += V^T . P is the key.
// (A first version on 32-query tiles, v_mfma_f32_32x32x16_f16, needed 128 accumulator + 128 Q-fragment registers per
// lane at head dim 256 - one wave per SIMD, every latency exposed: 2.98 ms whole forward at T = 5000 against 2.29 ms for
// the batched GEMMs.  Replaced by:)
// The split-precision core on 16-QUERY tiles (v_mfma_f32_16x16x32_f16): a wave owns 16 queries, so its state is 64
// accumulator + 64 Q-fragment registers at head dim 256 (the 32-query form above needs 128 + 128 and runs one wave per
// SIMD, every latency exposed) - two waves per SIMD, twice the workgroups.  Lane l = (query l & 15, group g = l >> 4):
//   S^T block  16 keys x 16 queries, reduction over 32 d per MFMA: A = K rows (lane = key, 8 d of group g: one hi and
//              one lo 16-byte LDS read), B = the Q fragments in registers; the lane holds keys 4 g + e of each block;
//   softmax    8 scores per lane and tile, the row reductions are two shuffles (xor 16, xor 32);
//   O^T block  16 d x 16 queries, reduction over the tile's 32 keys: B = P, whose 8 slots of group g are the lane's own
//              registers (slot (g, j) = key 16 (j >> 2) + 4 g + (j & 3)) - split, not moved; A = V^T from the
//              transposed planes in that slot order;
//   output     the lane holds O[query][16 b + 4 g + 0..3]: four consecutive floats - stored from registers.
template <int D, int NW>
__global__ __launch_bounds__(64 * NW, 512 / (64 * NW)) void flash_mhsa_h2q16_kernel(
    const char* __restrict__ q, const char* __restrict__ k, const char* __restrict__ v, long long ld, int T, float sqrt_d,
    float* __restrict__ ctx, long long ldo) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int KPB = D * 4 + 16;    // K tile pitch (bytes)
  constexpr int VPB = 32 * 2 + 16;   // V^T plane pitch (bytes)
  constexpr int DB = D / 16;         // 16-wide d blocks of the output
  constexpr int KS = D / 32;         // reduction steps of S^T
  constexpr int NTH = 64 * NW;
  constexpr int NCK = (32 * D * 4 / 16) / NTH;   // 16-byte K chunks per thread and tile
  constexpr int NPV = (32 * D / 8) / NTH;        // (key, run) pairs of V per thread and tile
  static_assert(NCK >= 1 && NPV >= 1, "tile staging needs at least one chunk / pair per thread");
  __shared__ __attribute__((aligned(16))) char ks[32 * KPB];
  __shared__ __attribute__((aligned(16))) char vth[D * VPB];
  __shared__ __attribute__((aligned(16))) char vtl[D * VPB];

  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int lq = lane & 15, lg = lane >> 4;
  const int h = blockIdx.y;
  const long long boff = (long long)blockIdx.z * T;
  const int q0 = blockIdx.x * (16 * NW) + wave * 16;
  const int qrow = q0 + lq;
  const bool q_ok = qrow < T;

  // Q as MFMA B fragments: step s covers d = 32 s + 8 g .. + 7 = run 4 s + g (hi | lo)
  uint4 qh[KS], ql[KS];
  {
    const char* qp = q + ((boff + (q_ok ? qrow : 0)) * ld + h * D) * 4;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const uint4* r = reinterpret_cast<const uint4*>(qp + (4 * s + lg) * 32);
      qh[s] = q_ok ? r[0] : make_uint4(0u, 0u, 0u, 0u);
      ql[s] = q_ok ? r[1] : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  f32x4 oacc[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) oacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float inv_sqrt_d = 1.f / sqrt_d;   // (head dims are powers of four: exact)

  const int vkey = t & 31;
  const int vslot = 8 * ((vkey >> 2) & 3) + 4 * (vkey >> 4) + (vkey & 3);
  uint4 kreg[NCK], vreg[NPV][2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      const int idx = t + NTH * c;
      const int lrow = idx / (D / 4), ch = idx - lrow * (D / 4);
      const bool ok = k0 + lrow < T;
      kreg[c] = *reinterpret_cast<const uint4*>(k + ((boff + (ok ? k0 + lrow : 0)) * ld + h * D) * 4 + ch * 16);
      if (!ok) kreg[c] = make_uint4(0u, 0u, 0u, 0u);
    }
    const bool vok = k0 + vkey < T;
#pragma unroll
    for (int c = 0; c < NPV; ++c) {
      const int run = (t >> 5) + (NTH / 32) * c;
      const uint4* src = reinterpret_cast<const uint4*>(v + ((boff + (vok ? k0 + vkey : 0)) * ld + h * D + 8 * run) * 4);
      vreg[c][0] = src[0];
      vreg[c][1] = src[1];
      if (!vok) vreg[c][0] = vreg[c][1] = make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      const int idx = t + NTH * c;
      const int lrow = idx / (D / 4), ch = idx - lrow * (D / 4);
      *reinterpret_cast<uint4*>(ks + lrow * KPB + ch * 16) = kreg[c];
    }
#pragma unroll
    for (int c = 0; c < NPV; ++c) {
      const int run = (t >> 5) + (NTH / 32) * c;
      const unsigned hw[4] = {vreg[c][0].x, vreg[c][0].y, vreg[c][0].z, vreg[c][0].w};
      const unsigned lw[4] = {vreg[c][1].x, vreg[c][1].y, vreg[c][1].z, vreg[c][1].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d0 = 8 * run + 2 * j;
        *reinterpret_cast<unsigned short*>(vth + d0 * VPB + vslot * 2) = (unsigned short)hw[j];
        *reinterpret_cast<unsigned short*>(vth + (d0 + 1) * VPB + vslot * 2) = (unsigned short)(hw[j] >> 16);
        *reinterpret_cast<unsigned short*>(vtl + d0 * VPB + vslot * 2) = (unsigned short)lw[j];
        *reinterpret_cast<unsigned short*>(vtl + (d0 + 1) * VPB + vslot * 2) = (unsigned short)(lw[j] >> 16);
      }
    }
  };

  gload(0);
  for (int k0 = 0; k0 < T; k0 += 32) {
    __syncthreads();   // the previous tile is no longer read
    lstore();
    __syncthreads();
    if (k0 + 32 < T) gload(k0 + 32);   // the next tile's rows are in flight during this tile's matrix work

    // S^T[key][query], two blocks of 16 keys
    f32x4 sacc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const avs_f16x8 bh = __builtin_bit_cast(avs_f16x8, qh[s]), bl = __builtin_bit_cast(avs_f16x8, ql[s]);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const uint4* kr = reinterpret_cast<const uint4*>(ks + (16 * kb + lq) * KPB + (4 * s + lg) * 32);
        const avs_f16x8 kh = __builtin_bit_cast(avs_f16x8, kr[0]), kl = __builtin_bit_cast(avs_f16x8, kr[1]);
        sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, bh, sacc[kb], 0, 0, 0);
        sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, bh, sacc[kb], 0, 0, 0);
        sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, bl, sacc[kb], 0, 0, 0);
      }
      if ((s & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    // scale, mask the keys past T, online softmax per query (the four lane groups hold different keys)
    float p8[8];
    float mt = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = k0 + 16 * kb + 4 * lg + e;
        const float sv = key < T ? sacc[kb][e] * inv_sqrt_d : -INFINITY;
        p8[4 * kb + e] = sv;
        mt = fmaxf(mt, sv);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float corr = expf(m_run - m_new);
    float ls = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      p8[j] = expf(p8[j] - m_new);
      ls += p8[j];
    }
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * corr + ls;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DB; ++i) oacc[i] *= corr;
    // P as fp16 hi | lo B fragment: slot j of this lane's group = its own register j
    uint4 ph, pl;
    avs_f16x2_split8(p8, ph, pl);
    const avs_f16x8 bh = __builtin_bit_cast(avs_f16x8, ph), bl = __builtin_bit_cast(avs_f16x8, pl);
    // O^T[d][query] += sum_key V[key][d] P[key][query]
#pragma unroll
    for (int i = 0; i < DB; ++i) {
      const int off = (16 * i + lq) * VPB + lg * 16;
      const avs_f16x8 vh = __builtin_bit_cast(avs_f16x8, *reinterpret_cast<const uint4*>(vth + off));
      const avs_f16x8 vl = __builtin_bit_cast(avs_f16x8, *reinterpret_cast<const uint4*>(vtl + off));
      oacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, bh, oacc[i], 0, 0, 0);
      oacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, bh, oacc[i], 0, 0, 0);
      oacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, bl, oacc[i], 0, 0, 0);
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }

  // normalise and store from registers: the lane holds O[query][16 i + 4 g + 0..3]
  if (q_ok) {
    const float inv_l = 1.f / l_run;
    float* orow = ctx + (boff + qrow) * ldo + h * D + 4 * lg;
#pragma unroll
    for (int i = 0; i < DB; ++i)
      *reinterpret_cast<float4*>(orow + 16 * i) = make_float4(oacc[i][0] * inv_l, oacc[i][1] * inv_l, oacc[i][2] * inv_l,
                                                             oacc[i][3] * inv_l);
  }
}

extern "C" int avs_mhsa_flash_f16x2(const void* d_q, const void* d_k, const void* d_v, int64_t ld, int b, int t,
                                    int heads, int head_dim, float* d_ctx, int64_t ldo, avs_stream_t stream) {
  const char* who = "avs_mhsa_flash_f16x2";
  AVS_REQUIRE(b > 0 && t >= 0 && heads > 0 && (head_dim == 64 || head_dim == 128 || head_dim == 256), AVS_E_SHAPE,
              "%s: b=%d t=%d heads=%d head_dim=%d (head_dim must be 64, 128 or 256)", who, b, t, heads, head_dim);
  AVS_REQUIRE(ld >= (int64_t)heads * head_dim && ldo >= (int64_t)heads * head_dim && ld % 8 == 0 && ldo % 4 == 0,
              AVS_E_SHAPE, "%s: row strides too small, or not multiples of 8 slots (q, k, v) / 4 floats (ctx)", who);
  if (t == 0) return AVS_OK;
  AVS_REQUIRE(d_q && d_k && d_v && d_ctx, AVS_E_ARG, "%s: null pointer", who);
  AVS_REQUIRE(((((uintptr_t)d_q) | ((uintptr_t)d_k) | ((uintptr_t)d_v)) & 31u) == 0 && avs_aligned16(d_ctx), AVS_E_ALIGN,
              "%s: q, k, v must be 32-byte aligned (AVS_F16X2), ctx 16-byte aligned", who);
  AVS_REQUIRE(b <= 65535 && heads <= 65535, AVS_E_SHAPE, "%s: grid too large", who);
  // 64 queries (4 waves of 16) per workgroup, two workgroups per CU
  dim3 grid((unsigned)avs_cdiv(t, 64), (unsigned)heads, (unsigned)b);
  const float sq = sqrtf((float)head_dim);
#define AVS_FLASH_H2_LAUNCH(DD)                                                                                       \
  hipLaunchKernelGGL((flash_mhsa_h2q16_kernel<DD, 4>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)d_q,     \
                     (const char*)d_k, (const char*)d_v, (long long)ld, t, sq, d_ctx, (long long)ldo)
  if (head_dim == 64) AVS_FLASH_H2_LAUNCH(64);
  else if (head_dim == 128) AVS_FLASH_H2_LAUNCH(128);
  else AVS_FLASH_H2_LAUNCH(256);
#undef AVS_FLASH_H2_LAUNCH
  AVS_CHECK_LAUNCH(who);
  return AVS_OK;
}

extern "C" int avs_mhsa_flash_f32(const float* d_q, const float* d_k, const float* d_v, int64_t ld, int b, int t,
                                  int heads, int head_dim, float* d_ctx, int64_t ldo, avs_stream_t stream) {
  AVS_REQUIRE(b > 0 && t >= 0 && heads > 0 && (head_dim == 64 || head_dim == 128 || head_dim == 256), AVS_E_SHAPE,
              "avs_mhsa_flash_f32: b=%d t=%d heads=%d head_dim=%d (head_dim must be 64, 128 or 256)", b, t, heads,
              head_dim);
  AVS_REQUIRE(ld >= (int64_t)heads * head_dim && ldo >= (int64_t)heads * head_dim && ld % 4 == 0 && ldo % 4 == 0,
              AVS_E_SHAPE, "avs_mhsa_flash_f32: row strides too small or not multiples of 4");
  if (t == 0) return AVS_OK;
  AVS_REQUIRE(d_q && d_k && d_v && d_ctx, AVS_E_ARG, "avs_mhsa_flash_f32: null pointer");
  AVS_REQUIRE(avs_aligned16(d_q) && avs_aligned16(d_k) && avs_aligned16(d_v) && avs_aligned16(d_ctx), AVS_E_ALIGN,
              "avs_mhsa_flash_f32: operands must be 16-byte aligned");
  AVS_REQUIRE(b <= 65535 && heads <= 65535, AVS_E_SHAPE, "avs_mhsa_flash_f32: grid too large");
  // 128 queries (4 waves) per workgroup; the 64-query form was measured slower at every size tried
  // (T = 1800: 1.24 vs 1.11 ms, T = 5000: 3.15 vs 2.75 ms at E = 1024, H = 4) and is kept for tiny grids only
  const bool small = avs_cdiv(t, 128) * heads * b < 16;
  const int qpw = small ? 64 : 128;
  dim3 grid((unsigned)avs_cdiv(t, qpw), (unsigned)heads, (unsigned)b);
  const float sq = sqrtf((float)head_dim);
#define AVS_FLASH_LAUNCH(DD, NWV)                                                                                    \
  hipLaunchKernelGGL((flash_mhsa_kernel<DD, NWV>), grid, dim3(64 * NWV), 0, (hipStream_t)stream, d_q, d_k, d_v,        \
                     (long long)ld, t, sq, d_ctx, (long long)ldo)
  if (head_dim == 64) {
    if (small) AVS_FLASH_LAUNCH(64, 2); else AVS_FLASH_LAUNCH(64, 4);
  } else if (head_dim == 128) {
    if (small) AVS_FLASH_LAUNCH(128, 2); else AVS_FLASH_LAUNCH(128, 4);
  } else {
    if (small) AVS_FLASH_LAUNCH(256, 2); else AVS_FLASH_LAUNCH(256, 4);
  }
#undef AVS_FLASH_LAUNCH
  AVS_CHECK_LAUNCH("avs_mhsa_flash_f32");
  return AVS_OK;
}
