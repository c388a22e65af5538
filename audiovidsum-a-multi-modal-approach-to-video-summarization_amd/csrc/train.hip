// Backward-pass kernels of the importance scorer (scripts/train_av_model.py:86-96).
// The dense parts of the backward (dX = dY.W, dW = dY^T.X) reuse avs_gemm_nt on transposed copies made by
// avs_transpose_f32; this file holds the rest: transposes, column sums (bias gradients), the element-wise
// gradient gates and the LSTM backward-through-time recurrence.
#include "avs_internal.h"
#include "lstm_h256.h"
#include <math.h>

// ---------------------------------------------------------------------------
// dst[c, r] = src[r, c]  (32x32 LDS tile, padded against bank conflicts).  Columns r >= rows of dst are left
// untouched: the caller zero-fills a buffer whose row stride is padded to 16 bytes.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int rows, int cols,
                                                        long long lds_, float* __restrict__ dst, long long ldd) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 8 * i][tx] = src[(long long)r * lds_ + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < rows && c < cols) dst[(long long)c * ldd + r] = tile[tx][ty + 8 * i];
  }
}

extern "C" int avs_transpose_f32(const float* d_src, int rows, int cols, int64_t ld_src, float* d_dst, int64_t ld_dst,
                                 avs_stream_t stream) {
  AVS_REQUIRE(rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= rows, AVS_E_SHAPE,
              "avs_transpose_f32: rows=%d cols=%d ld_src=%lld ld_dst=%lld", rows, cols, (long long)ld_src,
              (long long)ld_dst);
  if (rows == 0 || cols == 0) return AVS_OK;
  AVS_REQUIRE(d_src && d_dst, AVS_E_ARG, "avs_transpose_f32: null pointer");
  dim3 grid((unsigned)avs_cdiv(cols, 32), (unsigned)avs_cdiv(rows, 32));
  AVS_REQUIRE(grid.y <= 65535, AVS_E_SHAPE, "avs_transpose_f32: too many rows");
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_src, rows, cols, (long long)ld_src,
                     d_dst, (long long)ld_dst);
  AVS_CHECK_LAUNCH("avs_transpose_f32");
  return AVS_OK;
}

// ---------------------------------------------------------------------------
// out[c] = sum_r x[r, c] * (w ? w[r] : 1).  Bias gradients and the weighted sum of the scoring head
// (scripts/train_av_model.py:94: loss.backward()).  One block of 1024 threads per 16 columns: a thread = column c of the 16
// and row phase ph of 64 (a wave's load = 4 rows x 64 bytes), four independent accumulators per thread (rows ph, ph + 64,
// ph + 128, ph + 192 of every 256) - 128 blocks x 64 loads in flight for the 2048-column gate gradients, where one block
// per 64 columns with 4 dependent row chains (round 3) ran at 53 us per call.  Fixed summation order: deterministic.
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, long long rows, int cols,
                                                      long long ld, const float* __restrict__ w,
                                                      float* __restrict__ out) {
  __shared__ float red[64][16];
  const int t = threadIdx.x;
  const int cl = t & 15, ph = t >> 4;
  const int c = blockIdx.x * 16 + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < cols) {
    long long r = ph;
    for (; r + 192 < rows; r += 256) {
      const float x0 = x[r * ld + c], x1 = x[(r + 64) * ld + c], x2 = x[(r + 128) * ld + c], x3 = x[(r + 192) * ld + c];
      a0 = fmaf(x0, w ? w[r] : 1.f, a0);
      a1 = fmaf(x1, w ? w[r + 64] : 1.f, a1);
      a2 = fmaf(x2, w ? w[r + 128] : 1.f, a2);
      a3 = fmaf(x3, w ? w[r + 192] : 1.f, a3);
    }
    for (; r < rows; r += 64) a0 = fmaf(x[r * ld + c], w ? w[r] : 1.f, a0);
  }
  red[ph][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (t < 16 && c < cols) {
    float s = 0.f;
#pragma unroll 8
    for (int q = 0; q < 64; ++q) s += red[q][t];
    out[c] = s;
  }
}

extern "C" int avs_colsum_f32(const float* d_x, int64_t rows, int cols, int64_t ld, const float* d_row_weight,
                              float* d_out, avs_stream_t stream) {
  AVS_REQUIRE(rows >= 0 && cols > 0 && ld >= cols, AVS_E_SHAPE, "avs_colsum_f32: rows=%lld cols=%d ld=%lld",
              (long long)rows, cols, (long long)ld);
  AVS_REQUIRE(d_x && d_out, AVS_E_ARG, "avs_colsum_f32: null pointer");
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)avs_cdiv(cols, 16)), dim3(1024), 0, (hipStream_t)stream, d_x,
                     (long long)rows, cols, (long long)ld, d_row_weight, d_out);
  AVS_CHECK_LAUNCH("avs_colsum_f32");
  return AVS_OK;
}

// ---------------------------------------------------------------------------
// Gradient gate of Linear -> ReLU -> Dropout:  dpre = dy * (keep ? keep[i] : 1) * (y_relu[i] > 0)
// (keep = the inverted-dropout multiplier mask/(1-p) that the forward applied).  Also the forward
// y = x * keep when dy == nullptr is not needed: dropout forward is avs_mul_f32.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relu_drop_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ yr,
                                                            const float* __restrict__ keep, long long n,
                                                            float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float g = dy[i];
    if (keep) g *= keep[i];
    out[i] = yr[i] > 0.f ? g : 0.f;
  }
}

extern "C" int avs_relu_dropout_bwd_f32(const float* d_dy, const float* d_relu_out, const float* d_keep, int64_t n,
                                        float* d_out, avs_stream_t stream) {
  AVS_REQUIRE(n >= 0, AVS_E_SHAPE, "avs_relu_dropout_bwd_f32: negative size");
  if (n == 0) return AVS_OK;
  AVS_REQUIRE(d_dy && d_relu_out && d_out, AVS_E_ARG, "avs_relu_dropout_bwd_f32: null pointer");
  long long gx = avs_cdiv(n, 256);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, d_dy, d_relu_out,
                     d_keep, (long long)n, d_out);
  AVS_CHECK_LAUNCH("avs_relu_dropout_bwd_f32");
  return AVS_OK;
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  long long n, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = a[i] * b[i];
}

extern "C" int avs_mul_f32(const float* d_a, const float* d_b, int64_t n, float* d_out, avs_stream_t stream) {
  AVS_REQUIRE(n >= 0, AVS_E_SHAPE, "avs_mul_f32: negative size");
  if (n == 0) return AVS_OK;
  AVS_REQUIRE(d_a && d_b && d_out, AVS_E_ARG, "avs_mul_f32: null pointer");
  long long gx = avs_cdiv(n, 256);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(mul_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, d_a, d_b, (long long)n, d_out);
  AVS_CHECK_LAUNCH("avs_mul_f32");
  return AVS_OK;
}

// ---------------------------------------------------------------------------
// Scoring head backward: s = sigmoid(hid . w2 + b2).  dz[r] = ds[r] * s[r] * (1 - s[r]);
// dhid_pre[r, k] = dz[r] * w2[k] * (hid[r, k] > 0)   (hid is the ReLU output of scorer.0).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_head_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ s,
                                                             const float* __restrict__ hid, long long rows, int d,
                                                             long long ldh, const float* __restrict__ w2,
                                                             float* __restrict__ dz, float* __restrict__ dhid_pre) {
  const long long total = rows * d;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / d;
    const int k = (int)(i - r * d);
    const float sv = s[r];
    const float z = ds[r] * sv * (1.f - sv);
    if (k == 0) dz[r] = z;
    dhid_pre[r * d + k] = hid[r * ldh + k] > 0.f ? z * w2[k] : 0.f;
  }
}

extern "C" int avs_score_head_bwd_f32(const float* d_dscores, const float* d_scores, const float* d_hid, int64_t rows,
                                      int d, int64_t ldh, const float* d_w2, float* d_dz, float* d_dhid_pre,
                                      avs_stream_t stream) {
  AVS_REQUIRE(rows >= 0 && d > 0 && ldh >= d, AVS_E_SHAPE, "avs_score_head_bwd_f32: bad extents");
  if (rows == 0) return AVS_OK;
  AVS_REQUIRE(d_dscores && d_scores && d_hid && d_w2 && d_dz && d_dhid_pre, AVS_E_ARG,
              "avs_score_head_bwd_f32: null pointer");
  long long gx = avs_cdiv(rows * d, 256);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(score_head_bwd_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, d_dscores,
                     d_scores, d_hid, (long long)rows, d, (long long)ldh, d_w2, d_dz, d_dhid_pre);
  AVS_CHECK_LAUNCH("avs_score_head_bwd_f32");
  return AVS_OK;
}

// ---------------------------------------------------------------------------
// LSTM forward that also saves what the backward needs (post-activation gates i,f,g,o and the cell state).
// Same thread layout as lstm_kernel (scorer.hip).
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(1024) void lstm_train_fwd_kernel(const float* __restrict__ xproj,
                                                              const float* __restrict__ whh_t, int H, int ndir,
                                                              unsigned reverse_mask,
                                                              const int64_t* __restrict__ seq_rows,
                                                              float* __restrict__ out, long long ldo, int out_col0,
                                                              float* __restrict__ gates, float* __restrict__ cell,
                                                              int KQ, int kpq) {
  extern __shared__ float sm[];
  const int G = 4 * H, RV = H;
  float* h_s = sm;
  float* part = sm + H;
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
      float* gr = gates + row * ldx + (long long)dir * G;
      gr[tid] = ig;
      gr[H + tid] = fg;
      gr[2 * H + tid] = cg;
      gr[3 * H + tid] = og;
      cell[row * ((long long)ndir * H) + dir * H + tid] = c_state;
    }
    __syncthreads();
  }
}

extern "C" int avs_lstm_train_fwd_f32(const float* d_xproj, const float* d_whh_t, int hidden, int ndir,
                                      unsigned reverse_mask, const int64_t* d_seq_rows, int nseq, float* d_out,
                                      int64_t ldo, int out_col0, float* d_gates, float* d_cell, int variant,
                                      avs_stream_t stream) {
  AVS_REQUIRE(variant == AVS_LSTM_AUTO || variant == AVS_LSTM_STREAM, AVS_E_ARG, "avs_lstm_train_fwd_f32: bad variant %d", variant);
  AVS_REQUIRE(hidden > 0 && hidden <= 1024 && ndir > 0 && ndir <= 32 && nseq >= 0 && out_col0 >= 0 &&
                  ldo >= out_col0 + (int64_t)ndir * hidden,
              AVS_E_SHAPE, "avs_lstm_train_fwd_f32: bad extents");
  if (nseq == 0) return AVS_OK;
  AVS_REQUIRE(d_xproj && d_whh_t && d_seq_rows && d_out && d_gates && d_cell, AVS_E_ARG,
              "avs_lstm_train_fwd_f32: null pointer");
  if (hidden == 256 && variant != AVS_LSTM_STREAM) {
    // the scorer's size: the inference kernel's resident form (20 of a thread's 64 row-vectors of W_hh^T in registers, 8 in
    // LDS) that also stores the gates and the cell state - the same fmaf chain, bit-identical to the streaming kernel
    AVS_REQUIRE(avs_aligned16(d_whh_t), AVS_E_ALIGN, "avs_lstm_train_fwd_f32: whh_t not 16-byte aligned");
    const bool ok = lstm_h256_launch<20, 8, 4, true>(d_xproj, d_whh_t, ndir, reverse_mask, d_seq_rows, nseq, d_out, (long long)ldo,
                                                     out_col0, d_gates, d_cell, (hipStream_t)stream);
    AVS_REQUIRE(ok, AVS_E_HIP, "avs_lstm_train_fwd_f32: cannot reserve the LDS of the resident form");
    AVS_CHECK_LAUNCH("avs_lstm_train_fwd_f32");
    return AVS_OK;
  }
  int KQ = 1024 / hidden;
  if (KQ > hidden) KQ = hidden;
  if (KQ < 1) KQ = 1;
  const int kpq = (hidden + KQ - 1) / KQ;
  const size_t shmem = ((size_t)hidden + (size_t)KQ * 4 * hidden) * sizeof(float);
  hipLaunchKernelGGL(lstm_train_fwd_kernel, dim3(nseq, ndir), dim3(1024), shmem, (hipStream_t)stream, d_xproj,
                     d_whh_t, hidden, ndir, reverse_mask, d_seq_rows, d_out, (long long)ldo, out_col0, d_gates, d_cell,
                     KQ, kpq);
  AVS_CHECK_LAUNCH("avs_lstm_train_fwd_f32");
  return AVS_OK;
}

// ---------------------------------------------------------------------------
// LSTM backward through time.  One workgroup per (sequence, direction), steps in reverse processing order.
//   dh = dout[t] + W_hh^T . da_{next};  do = dh*tanh(c);  dc = dh*o*(1-tanh(c)^2) + dc_next;
//   di = dc*g; dg = dc*i; df = dc*c_prev;  dc_next = dc*f;
//   da = [di*i(1-i), df*f(1-f), dg*(1-g^2), do*o(1-o)]   -> d_dxproj[t]
// W_hh in its ORIGINAL layout [dir][4H][H]: dh_next[k] = sum_j W_hh[j][k] * da[j] reads rows j with lanes over k.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void lstm_bwd_kernel(const float* __restrict__ dout, long long ldo, int out_col0,
                                                        const float* __restrict__ gates,
                                                        const float* __restrict__ cell,
                                                        const float* __restrict__ whh, int H, int ndir,
                                                        unsigned reverse_mask,
                                                        const int64_t* __restrict__ seq_rows,
                                                        float* __restrict__ dxproj, int JQ, int jpq) {
  extern __shared__ float sm[];
  const int G = 4 * H;
  float* da_s = sm;          // [G]
  float* part = sm + G;      // [JQ][H]
  const int seq = blockIdx.x, dir = blockIdx.y;
  const long long r0 = seq_rows[seq], r1 = seq_rows[seq + 1];
  const long long T = r1 - r0;
  const bool rev = (reverse_mask >> dir) & 1u;
  const float* __restrict__ W = whh + (long long)dir * G * H;
  const long long ldg = (long long)ndir * G, ldc = (long long)ndir * H;
  const int tid = threadIdx.x;
  // matvec layout: thread = (slice jq of the 4H gate rows, hidden unit k)
  const int jq = tid / H, kk = tid - jq * H;
  const bool mv = jq < JQ;
  const int j0 = jq * jpq;
  const int j1 = (j0 + jpq) < G ? (j0 + jpq) : G;
  float dc_next = 0.f, dh_next = 0.f;
  for (long long s = T - 1; s >= 0; --s) {
    const long long row = rev ? (r1 - 1 - s) : (r0 + s);
    if (tid < H) {
      const long long prow = rev ? row + 1 : row - 1;  // the step processed just before this one in the forward
      const float c_prev = s > 0 ? cell[prow * ldc + dir * H + tid] : 0.f;
      const float* gr = gates + row * ldg + (long long)dir * G;
      const float ig = gr[tid], fg = gr[H + tid], cg = gr[2 * H + tid], og = gr[3 * H + tid];
      const float c = cell[row * ldc + dir * H + tid];
      const float tc = tanhf(c);
      const float dh = dout[row * ldo + out_col0 + dir * H + tid] + dh_next;
      const float d_o = dh * tc;
      const float dc = dh * og * (1.f - tc * tc) + dc_next;
      const float d_i = dc * cg, d_g = dc * ig, d_f = dc * c_prev;
      dc_next = dc * fg;
      const float ai = d_i * ig * (1.f - ig), af = d_f * fg * (1.f - fg), ag = d_g * (1.f - cg * cg),
                  ao = d_o * og * (1.f - og);
      da_s[tid] = ai;
      da_s[H + tid] = af;
      da_s[2 * H + tid] = ag;
      da_s[3 * H + tid] = ao;
      float* dx = dxproj + row * ldg + (long long)dir * G;
      dx[tid] = ai;
      dx[H + tid] = af;
      dx[2 * H + tid] = ag;
      dx[3 * H + tid] = ao;
    }
    __syncthreads();
    if (mv) {
      float a = 0.f;
      const float* wp = W + (long long)j0 * H + kk;
#pragma unroll 8
      for (int j = j0; j < j1; ++j) {
        a = fmaf(*wp, da_s[j], a);
        wp += H;
      }
      part[jq * H + kk] = a;
    }
    __syncthreads();
    if (tid < H) {
      float a = 0.f;
      for (int q = 0; q < JQ; ++q) a += part[q * H + tid];
      dh_next = a;
    }
    // part / da_s are rewritten only after the next barrier pair
  }
}

// Backward through time at the scorer's size (hidden = 256), W_hh partly resident on chip like the forward's lstm_h256_kernel.
// dh_next[k] = sum_j W_hh[j][k] da[j]: W_hh [4H][H] has the shape of the forward's W_hh^T with rows and columns exchanged, so
// the same decomposition applies - a thread owns 64 row-vectors (rows j of one of 16 slices x 4 consecutive k), the first RK
// in registers, the next LK in LDS, the rest streamed in batches of DEPTH by buffer loads (a wave = one slice, so the slice base
// is scalar and a wave's load is one whole 1 KB row of W_hh); da is broadcast from LDS.  The step's inputs (saved gates, cell
// state, dL/dh) are needed BEFORE its matrix-vector product, so they are fetched one step ahead, under the previous product.
// Summation order: rows ascending inside a slice, then the 16 slices ascending (fixed: deterministic).
template <int RK, int LK, int DEPTH>
__global__ __launch_bounds__(1024) void lstm_bwd_h256_kernel(const float* __restrict__ dout, long long ldo, int out_col0,
                                                             const float* __restrict__ gates,
                                                             const float* __restrict__ cell,
                                                             const float* __restrict__ whh, int ndir,
                                                             unsigned reverse_mask, const int64_t* __restrict__ seq_rows,
                                                             float* __restrict__ dxproj) {
  constexpr int H = 256, G = 4 * H, JS = 16, JPS = 64;
  static_assert(RK + LK <= JPS, "a thread owns 64 row-vectors");
  extern __shared__ float sm[];
  float* da_s = sm;                                          // [G]
  float* part = sm + G;                                      // [JS][H]
  float4* wl = reinterpret_cast<float4*>(sm + G + JS * H);   // [LK][1024]: thread-private slots
  const int seq = blockIdx.x, dir = blockIdx.y;
  const long long r0 = seq_rows[seq], r1 = seq_rows[seq + 1];
  const long long T = r1 - r0;
  const bool rev = (reverse_mask >> dir) & 1u;
  const long long ldg = (long long)ndir * G, ldc = (long long)ndir * H;
  const int tid = threadIdx.x;
  const int js = __builtin_amdgcn_readfirstlane(tid >> 6), kv = tid & 63;
  const int j0 = js * JPS;
  const float* __restrict__ W = whh + (long long)dir * G * H;
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(W)) + (size_t)j0 * H * 4, 0, JPS * H * 4, 0x00020000);
  const int voff = kv * 16;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  auto ldw = [&](int i) -> float4 {
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, voff, i * (H * 4), 0));
    return make_float4(v[0], v[1], v[2], v[3]);
  };
  float4 wr[RK];
#pragma unroll
  for (int i = 0; i < RK; ++i) wr[i] = ldw(i);
#pragma unroll
  for (int i = 0; i < LK; ++i) wl[i * 1024 + tid] = ldw(RK + i);

  const float* __restrict__ gbase = gates + (long long)dir * G + tid;
  const float* __restrict__ cbase = cell + dir * H + tid;
  const float* __restrict__ dbase = dout + out_col0 + dir * H + tid;
  auto row_of = [&](long long s) -> long long { return rev ? (r1 - 1 - s) : (r0 + s); };
  // the inputs of the step about to be processed (threads < H)
  float p_i = 0.f, p_f = 0.f, p_g = 0.f, p_o = 0.f, p_c = 0.f, p_cprev = 0.f, p_d = 0.f;
  if (tid < H && T > 0) {
    const long long row = row_of(T - 1);
    p_i = gbase[row * ldg];
    p_f = gbase[row * ldg + H];
    p_g = gbase[row * ldg + 2 * H];
    p_o = gbase[row * ldg + 3 * H];
    p_c = cbase[row * ldc];
    p_d = dbase[row * ldo];
    if (T > 1) p_cprev = cbase[row_of(T - 2) * ldc];
  }
  float dc_next = 0.f, dh_next = 0.f;
  for (long long s = T - 1; s >= 0; --s) {
    const long long row = row_of(s);
    if (tid < H) {
      const float ig = p_i, fg = p_f, cg = p_g, og = p_o, c = p_c, c_prev = s > 0 ? p_cprev : 0.f;
      const float tc = tanhf(c);
      const float dh = p_d + dh_next;
      const float d_o = dh * tc;
      const float dc = dh * og * (1.f - tc * tc) + dc_next;
      const float d_i = dc * cg, d_g = dc * ig, d_f = dc * c_prev;
      dc_next = dc * fg;
      const float ai = d_i * ig * (1.f - ig), af = d_f * fg * (1.f - fg), ag = d_g * (1.f - cg * cg),
                  ao = d_o * og * (1.f - og);
      da_s[tid] = ai;
      da_s[H + tid] = af;
      da_s[2 * H + tid] = ag;
      da_s[3 * H + tid] = ao;
      float* dx = dxproj + row * ldg + (long long)dir * G;
      dx[tid] = ai;
      dx[H + tid] = af;
      dx[2 * H + tid] = ag;
      dx[3 * H + tid] = ao;
      if (s > 0) {   // the next step's inputs, in flight under this step's product
        const long long nrow = row_of(s - 1);
        p_i = gbase[nrow * ldg];
        p_f = gbase[nrow * ldg + H];
        p_g = gbase[nrow * ldg + 2 * H];
        p_o = gbase[nrow * ldg + 3 * H];
        p_d = dbase[nrow * ldo];
        p_c = p_cprev;
        if (s > 1) p_cprev = cbase[row_of(s - 2) * ldc];
      }
    }
    __syncthreads();
    constexpr int NS = JPS - RK - LK;
    static_assert(NS % DEPTH == 0, "whole batches");
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < RK; ++i) {
      const float dj = da_s[j0 + i];
      a.x = fmaf(wr[i].x, dj, a.x);
      a.y = fmaf(wr[i].y, dj, a.y);
      a.z = fmaf(wr[i].z, dj, a.z);
      a.w = fmaf(wr[i].w, dj, a.w);
    }
#pragma unroll 2
    for (int i = 0; i < LK; ++i) {
      const float4 wv = wl[i * 1024 + tid];
      const float dj = da_s[j0 + RK + i];
      a.x = fmaf(wv.x, dj, a.x);
      a.y = fmaf(wv.y, dj, a.y);
      a.z = fmaf(wv.z, dj, a.z);
      a.w = fmaf(wv.w, dj, a.w);
    }
#pragma unroll 1
    for (int b0 = 0; b0 < NS; b0 += DEPTH) {
      float4 ws[DEPTH];
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) ws[i] = ldw(RK + LK + b0 + i);
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) {
        const float dj = da_s[j0 + RK + LK + b0 + i];
        a.x = fmaf(ws[i].x, dj, a.x);
        a.y = fmaf(ws[i].y, dj, a.y);
        a.z = fmaf(ws[i].z, dj, a.z);
        a.w = fmaf(ws[i].w, dj, a.w);
      }
    }
    reinterpret_cast<float4*>(part + js * H)[kv] = a;
    __syncthreads();
    if (tid < H) {
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < JS; ++q) acc += part[q * H + tid];
      dh_next = acc;
    }
    // da_s / part are rewritten only behind the next barrier pair
  }
}

extern "C" int avs_lstm_bwd_f32(const float* d_dout, int64_t ldo, int out_col0, const float* d_gates,
                                const float* d_cell, const float* d_whh, int hidden, int ndir, unsigned reverse_mask,
                                const int64_t* d_seq_rows, int nseq, float* d_dxproj, int variant, avs_stream_t stream) {
  AVS_REQUIRE(variant == AVS_LSTM_AUTO || variant == AVS_LSTM_STREAM, AVS_E_ARG, "avs_lstm_bwd_f32: bad variant %d", variant);
  AVS_REQUIRE(hidden > 0 && hidden <= 1024 && ndir > 0 && ndir <= 32 && nseq >= 0 && out_col0 >= 0 &&
                  ldo >= out_col0 + (int64_t)ndir * hidden,
              AVS_E_SHAPE, "avs_lstm_bwd_f32: bad extents");
  if (nseq == 0) return AVS_OK;
  AVS_REQUIRE(d_dout && d_gates && d_cell && d_whh && d_seq_rows && d_dxproj, AVS_E_ARG, "avs_lstm_bwd_f32: null pointer");
  if (hidden == 256 && variant != AVS_LSTM_STREAM) {
    AVS_REQUIRE(avs_aligned16(d_whh), AVS_E_ALIGN, "avs_lstm_bwd_f32: whh not 16-byte aligned");
    constexpr int RK = 20, LK = 8, DEPTH = 4;
    const size_t shm = ((size_t)4 * 256 + 16 * 256) * sizeof(float) + (size_t)LK * 1024 * sizeof(float4);
    AVS_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd_h256_kernel<RK, LK, DEPTH>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess,
                AVS_E_HIP, "avs_lstm_bwd_f32: cannot reserve %zu bytes of LDS", shm);
    hipLaunchKernelGGL((lstm_bwd_h256_kernel<RK, LK, DEPTH>), dim3(nseq, ndir), dim3(1024), shm, (hipStream_t)stream, d_dout,
                       (long long)ldo, out_col0, d_gates, d_cell, d_whh, ndir, reverse_mask, d_seq_rows, d_dxproj);
    AVS_CHECK_LAUNCH("avs_lstm_bwd_f32");
    return AVS_OK;
  }
  int JQ = 1024 / hidden;
  if (JQ < 1) JQ = 1;
  if (JQ > 4 * hidden) JQ = 4 * hidden;
  const int jpq = (4 * hidden + JQ - 1) / JQ;
  const size_t shmem = ((size_t)4 * hidden + (size_t)JQ * hidden) * sizeof(float);
  hipLaunchKernelGGL(lstm_bwd_kernel, dim3(nseq, ndir), dim3(1024), shmem, (hipStream_t)stream, d_dout, (long long)ldo,
                     out_col0, d_gates, d_cell, d_whh, hidden, ndir, reverse_mask, d_seq_rows, d_dxproj, JQ, jpq);
  AVS_CHECK_LAUNCH("avs_lstm_bwd_f32");
  return AVS_OK;
}
