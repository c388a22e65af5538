// Importance-scorer kernels: the LSTM recurrence, the
// batch-axis attention core of nn.MultiheadAttention fed without batch_first,
// the scoring head and a row softmax.  The GEMM-shaped parts (input
// projections, in/out projections, scorer.0) go through avs_gemm_nt.
#include "avs_internal.h"
#include "lstm_h256.h"
#include <math.h>

// ---------------------------------------------------------------------------
// LSTM recurrence.  One 1024-thread workgroup per (sequence, direction).
// Per step: g = xproj[t] + W_hh . h_{t-1}.  W_hh^T [H][4H] is streamed from L2
// with 16-byte loads (thread = 4 consecutive gate rows x one slice of k), h
// lives in LDS and is read as a broadcast; partial sums cross k-slices through
// LDS, then H threads apply the gates.  Latency-bound by design: sequences
// are independent, so the grid is sequences x directions.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void lstm_kernel(const float* __restrict__ xproj,
                                                    const float* __restrict__ whh_t, int H, int ndir,
                                                    unsigned reverse_mask, const int64_t* __restrict__ seq_rows,
                                                    float* __restrict__ out, long long ldo, int out_col0, int KQ,
                                                    int kpq) {
  extern __shared__ float sm[];
  const int G = 4 * H;
  const int RV = H;  // float4 row-vectors (G / 4)
  float* h_s = sm;             // [H]
  float* part = sm + H;        // [KQ][G]

  const int seq = blockIdx.x, dir = blockIdx.y;
  const long long r0 = seq_rows[seq], r1 = seq_rows[seq + 1];
  const long long T = r1 - r0;
  const bool rev = (reverse_mask >> dir) & 1u;
  const float* __restrict__ W = whh_t + (long long)dir * H * G;
  const long long ldx = (long long)ndir * G;
  const float* __restrict__ xp = xproj + (long long)dir * G;

  const int tid = threadIdx.x;
  const int kq = tid / RV, jv = tid - kq * RV;
  const bool mv = kq < KQ;
  const int k0 = kq * kpq;
  const int k1 = (k0 + kpq) < H ? (k0 + kpq) : H;

  float c_state = 0.f;
  if (tid < H) h_s[tid] = 0.f;
  __syncthreads();

  for (long long s = 0; s < T; ++s) {
    const long long row = rev ? (r1 - 1 - s) : (r0 + s);
    float xi = 0.f, xf = 0.f, xg = 0.f, xo = 0.f;
    if (tid < H) {
      const float* xr = xp + row * ldx;
      xi = xr[tid];
      xf = xr[H + tid];
      xg = xr[2 * H + tid];
      xo = xr[3 * H + tid];
    }
    if (mv) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4* wp = reinterpret_cast<const float4*>(W) + (long long)k0 * RV + jv;
#pragma unroll 8
      for (int k = k0; k < k1; ++k) {
        const float4 wv = *wp;
        wp += RV;
        const float hk = h_s[k];
        a.x = fmaf(wv.x, hk, a.x);
        a.y = fmaf(wv.y, hk, a.y);
        a.z = fmaf(wv.z, hk, a.z);
        a.w = fmaf(wv.w, hk, a.w);
      }
      reinterpret_cast<float4*>(part + (long long)kq * G)[jv] = a;
    }
    __syncthreads();
    if (tid < H) {
      float gi = xi, gf = xf, gg = xg, go = xo;
      for (int q = 0; q < KQ; ++q) {
        const float* pq = part + q * G;
        gi += pq[tid];
        gf += pq[H + tid];
        gg += pq[2 * H + tid];
        go += pq[3 * H + tid];
      }
      const float ig = avs_sigmoid(gi), fg = avs_sigmoid(gf), cg = tanhf(gg), og = avs_sigmoid(go);
      c_state = fg * c_state + ig * cg;
      const float hv = og * tanhf(c_state);
      h_s[tid] = hv;
      out[row * ldo + out_col0 + dir * H + tid] = hv;
    }
    __syncthreads();
  }
}

extern "C" int avs_lstm_f32(const float* d_xproj, const float* d_whh_t, int hidden, int ndir, unsigned reverse_mask,
                            const int64_t* d_seq_rows, int nseq, float* d_out, int64_t ldo, int out_col0, int variant,
                            avs_stream_t stream) {
  AVS_REQUIRE(variant >= AVS_LSTM_AUTO && variant <= AVS_LSTM_RESIDENT_16_8, AVS_E_ARG, "avs_lstm_f32: bad variant %d",
              variant);
  AVS_REQUIRE(variant <= AVS_LSTM_STREAM || hidden == 256, AVS_E_UNSUPPORTED,
              "avs_lstm_f32: the resident variants are built for hidden = 256");
  AVS_REQUIRE(hidden > 0 && hidden <= 1024 && ndir > 0 && ndir <= 32 && nseq >= 0 && out_col0 >= 0 &&
                  ldo >= out_col0 + (int64_t)ndir * hidden,
              AVS_E_SHAPE, "avs_lstm_f32: hidden=%d ndir=%d nseq=%d ldo=%lld out_col0=%d", hidden, ndir, nseq,
              (long long)ldo, out_col0);
  if (nseq == 0) return AVS_OK;
  AVS_REQUIRE(d_xproj && d_whh_t && d_seq_rows && d_out, AVS_E_ARG, "avs_lstm_f32: null pointer");
  AVS_REQUIRE(avs_aligned16(d_whh_t), AVS_E_ALIGN, "avs_lstm_f32: whh_t not 16-byte aligned");
  AVS_REQUIRE(nseq <= 65535 * 32767, AVS_E_SHAPE, "avs_lstm_f32: too many sequences");
  if (hidden == 256 && variant != AVS_LSTM_STREAM) {
    // 1: 20 row-vectors in registers + 8 in LDS, batches of 4;  2: 16 + 8, batches of 8
    const bool ok = variant == AVS_LSTM_RESIDENT_16_8
                        ? lstm_h256_launch<16, 8, 8, false>(d_xproj, d_whh_t, ndir, reverse_mask, d_seq_rows, nseq, d_out,
                                                            (long long)ldo, out_col0, nullptr, nullptr, (hipStream_t)stream)
                        : lstm_h256_launch<20, 8, 4, false>(d_xproj, d_whh_t, ndir, reverse_mask, d_seq_rows, nseq, d_out,
                                                            (long long)ldo, out_col0, nullptr, nullptr, (hipStream_t)stream);
    AVS_REQUIRE(ok, AVS_E_HIP, "avs_lstm_f32: cannot reserve the LDS of the resident form");
    AVS_CHECK_LAUNCH("avs_lstm_f32");
    return AVS_OK;
  }
  int KQ = 1024 / hidden;
  if (KQ > hidden) KQ = hidden;
  if (KQ < 1) KQ = 1;
  const int kpq = (hidden + KQ - 1) / KQ;
  const size_t shmem = ((size_t)hidden + (size_t)KQ * 4 * hidden) * sizeof(float);
  hipLaunchKernelGGL(lstm_kernel, dim3(nseq, ndir), dim3(1024), shmem, (hipStream_t)stream, d_xproj, d_whh_t, hidden,
                     ndir, reverse_mask, d_seq_rows, d_out, (long long)ldo, out_col0, KQ, kpq);
  AVS_CHECK_LAUNCH("avs_lstm_f32");
  return AVS_OK;
}

// ---------------------------------------------------------------------------
// nn.MultiheadAttention(E, heads) fed [B,T,E] without batch_first: the attended
// axis is B, every time-step is an independent "batch" entry.
// One wave per (t, head, b): lanes over the head dimension, dot products by
// wave shuffle, online softmax over the B keys.
// ---------------------------------------------------------------------------
#define AVS_MHA_MAX_PER_LANE 8
__global__ __launch_bounds__(256) void mha_batchaxis_kernel(const float* __restrict__ qkv, int B, int T, int E,
                                                            int heads, float scale, float* __restrict__ ctx) {
  const int dh = E / heads;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const long long total = (long long)T * heads * B;
  if (wave >= total) return;
  const int b = (int)(wave % B);
  const long long th = wave / B;
  const int hd = (int)(th % heads);
  const int t = (int)(th / heads);
  const long long ld = 3ll * E;
  const float* qrow = qkv + ((long long)b * T + t) * ld + hd * dh;
  float q[AVS_MHA_MAX_PER_LANE], acc[AVS_MHA_MAX_PER_LANE];
#pragma unroll
  for (int j = 0; j < AVS_MHA_MAX_PER_LANE; ++j) {
    const int d = lane + 64 * j;
    q[j] = d < dh ? qrow[d] * scale : 0.f;
    acc[j] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int b2 = 0; b2 < B; ++b2) {
    const float* krow = qkv + ((long long)b2 * T + t) * ld + E + hd * dh;
    const float* vrow = krow + E;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < AVS_MHA_MAX_PER_LANE; ++j) {
      const int d = lane + 64 * j;
      if (d < dh) s = fmaf(q[j], krow[d], s);
    }
    s = avs_wave_sum(s);
    const float mn = fmaxf(m, s);
    const float corr = expf(m - mn);  // exp(-inf) = 0 on the first key
    const float pexp = expf(s - mn);
    l = l * corr + pexp;
#pragma unroll
    for (int j = 0; j < AVS_MHA_MAX_PER_LANE; ++j) {
      const int d = lane + 64 * j;
      if (d < dh) acc[j] = acc[j] * corr + pexp * vrow[d];
    }
    m = mn;
  }
  float* orow = ctx + ((long long)b * T + t) * E + hd * dh;
#pragma unroll
  for (int j = 0; j < AVS_MHA_MAX_PER_LANE; ++j) {
    const int d = lane + 64 * j;
    if (d < dh) orow[d] = acc[j] / l;
  }
}

extern "C" int avs_mha_batchaxis_f32(const float* d_qkv, int b, int t, int e, int heads, float* d_ctx,
                                     avs_stream_t stream) {
  AVS_REQUIRE(b > 0 && t >= 0 && e > 0 && heads > 0 && e % heads == 0 && e / heads <= 64 * AVS_MHA_MAX_PER_LANE,
              AVS_E_SHAPE, "avs_mha_batchaxis_f32: b=%d t=%d e=%d heads=%d (head dim <= %d)", b, t, e, heads,
              64 * AVS_MHA_MAX_PER_LANE);
  if (t == 0) return AVS_OK;
  AVS_REQUIRE(d_qkv && d_ctx, AVS_E_ARG, "avs_mha_batchaxis_f32: null pointer");
  const long long waves = (long long)t * heads * b;
  const long long blocks = avs_cdiv(waves, 4);
  AVS_REQUIRE(blocks < (1ll << 31), AVS_E_SHAPE, "avs_mha_batchaxis_f32: too many waves");
  const float scale = (float)(1.0 / sqrt((double)(e / heads)));
  hipLaunchKernelGGL(mha_batchaxis_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_qkv, b, t, e,
                     heads, scale, d_ctx);
  AVS_CHECK_LAUNCH("avs_mha_batchaxis_f32");
  return AVS_OK;
}

// scores[r] = sigmoid(dot(hid[r,:], w2) + b2): one wave per row.
__global__ __launch_bounds__(256) void score_head_kernel(const float* __restrict__ hid, long long rows, int d,
                                                         long long ldh, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, float* __restrict__ scores) {
  const long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int k = lane; k < d; k += 64) s = fmaf(hid[row * ldh + k], w2[k], s);
  s = avs_wave_sum(s);
  if (lane == 0) scores[row] = avs_sigmoid(s + b2[0]);
}

extern "C" int avs_score_head_f32(const float* d_hid, int64_t rows, int d, int64_t ldh, const float* d_w2,
                                  const float* d_b2, float* d_scores, avs_stream_t stream) {
  AVS_REQUIRE(rows >= 0 && d > 0 && ldh >= d, AVS_E_SHAPE, "avs_score_head_f32: rows=%lld d=%d ldh=%lld",
              (long long)rows, d, (long long)ldh);
  if (rows == 0) return AVS_OK;
  AVS_REQUIRE(d_hid && d_w2 && d_b2 && d_scores, AVS_E_ARG, "avs_score_head_f32: null pointer");
  const long long blocks = avs_cdiv(rows, 4);
  AVS_REQUIRE(blocks < (1ll << 31), AVS_E_SHAPE, "avs_score_head_f32: too many rows");
  hipLaunchKernelGGL(score_head_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_hid,
                     (long long)rows, d, (long long)ldh, d_w2, d_b2, d_scores);
  AVS_CHECK_LAUNCH("avs_score_head_f32");
  return AVS_OK;
}

// In-place row softmax, one 256-thread block per row: max and sum by wave
// shuffle, then across the 4 waves through LDS.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, int n, long long ldx) {
  __shared__ float red[4];
  float* row = x + (long long)blockIdx.x * ldx;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, row[i]);
  m = avs_wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float e = expf(row[i] - m);
    row[i] = e;
    s += e;
  }
  s = avs_wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  for (int i = threadIdx.x; i < n; i += 256) row[i] = row[i] / s;
}

extern "C" int avs_softmax_rows_f32(float* d_x, int64_t rows, int n, int64_t ldx, avs_stream_t stream) {
  AVS_REQUIRE(rows >= 0 && n > 0 && ldx >= n, AVS_E_SHAPE, "avs_softmax_rows_f32: rows=%lld n=%d ldx=%lld",
              (long long)rows, n, (long long)ldx);
  if (rows == 0) return AVS_OK;
  AVS_REQUIRE(d_x, AVS_E_ARG, "avs_softmax_rows_f32: null pointer");
  AVS_REQUIRE(rows < (1ll << 31), AVS_E_SHAPE, "avs_softmax_rows_f32: too many rows");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, d_x, n,
                     (long long)ldx);
  AVS_CHECK_LAUNCH("avs_softmax_rows_f32");
  return AVS_OK;
}

// Backward of the row softmax, in place on the upstream gradient: ds[r, j] = alpha * p[r, j] * (dp[r, j] - sum_k p[r, k]
// dp[r, k]) - the gradient with respect to the UNSCALED scores q.k of softmax(alpha * q.k) (models/attention.py:21-22).
// One 256-thread block per row; the dot product by wave shuffle, the four waves added in a fixed order.
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ p, float* __restrict__ dp,
                                                               int n, long long ld, float alpha) {
  __shared__ float red[4];
  const float* prow = p + (long long)blockIdx.x * ld;
  float* drow = dp + (long long)blockIdx.x * ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s = fmaf(prow[i], drow[i], s);
  s = avs_wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  for (int i = threadIdx.x; i < n; i += 256) drow[i] = alpha * prow[i] * (drow[i] - s);
}

extern "C" int avs_softmax_bwd_rows_f32(const float* d_p, float* d_dp, int64_t rows, int n, int64_t ld, float alpha,
                                        avs_stream_t stream) {
  AVS_REQUIRE(rows >= 0 && n > 0 && ld >= n, AVS_E_SHAPE, "avs_softmax_bwd_rows_f32: rows=%lld n=%d ld=%lld",
              (long long)rows, n, (long long)ld);
  if (rows == 0) return AVS_OK;
  AVS_REQUIRE(d_p && d_dp, AVS_E_ARG, "avs_softmax_bwd_rows_f32: null pointer");
  AVS_REQUIRE(rows < (1ll << 31), AVS_E_SHAPE, "avs_softmax_bwd_rows_f32: too many rows");
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, d_p, d_dp, n,
                     (long long)ld, alpha);
  AVS_CHECK_LAUNCH("avs_softmax_bwd_rows_f32");
  return AVS_OK;
}
