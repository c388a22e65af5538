// The hidden = 256 LSTM recurrence with part of W_hh^T resident in registers / LDS: shared by the inference entry
// (scorer.hip: avs_lstm_f32) and the training forward (train.hip: avs_lstm_train_fwd_f32).
#pragma once
#include "avs_internal.h"
#include <math.h>

__device__ __forceinline__ float avs_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// The scorer's own size (hidden = 256: W_hh^T is 1 MB, a thread's share 64 float4): one CU cannot pull 1 MB out of L2
// faster than ~8 us, which WAS the step.  Here a thread keeps the first RK of its 64 row-vectors in registers and the
// next LK in LDS for the whole sequence, and streams only the rest: the same fmaf chain in the same order as
// lstm_kernel (bit-identical results), half the bytes per step.
// TRAIN: the forward of the training step (avs_lstm_train_fwd_f32) - the same recurrence, and the post-activation gates
// i, f, g, o [rows, ndir * 4H] and the cell state [rows, ndir * H] are written for the backward sweep.
template <int RK, int LK, int DEPTH, bool TRAIN>
__global__ __launch_bounds__(1024) void lstm_h256_kernel(const float* __restrict__ xproj,
                                                         const float* __restrict__ whh_t, int ndir,
                                                         unsigned reverse_mask, const int64_t* __restrict__ seq_rows,
                                                         float* __restrict__ out, long long ldo, int out_col0,
                                                         float* __restrict__ gates, float* __restrict__ cell) {
  constexpr int H = 256, G = 4 * H, RV = H, KQ = 4, KPQ = 64;
  static_assert(RK + LK <= KPQ, "a thread owns 64 row-vectors");
  extern __shared__ float sm[];
  float* h_s = sm;                                        // [H]
  float* part = sm + H;                                   // [KQ][G]
  float4* wl = reinterpret_cast<float4*>(sm + H + KQ * G);  // [LK][1024]: thread-private slots, conflict-free

  const int seq = blockIdx.x, dir = blockIdx.y;
  const long long r0 = seq_rows[seq], r1 = seq_rows[seq + 1];
  const long long T = r1 - r0;
  const bool rev = (reverse_mask >> dir) & 1u;
  const float* __restrict__ W = whh_t + (long long)dir * H * G;
  const long long ldx = (long long)ndir * G;
  const float* __restrict__ xp = xproj + (long long)dir * G;

  const int tid = threadIdx.x;
  // a k-slice is shared by 256 threads = 4 whole waves: its base is wave-uniform (scalar registers), a thread adds only
  // its 32-bit column offset - 36 distinct 64-bit row addresses per thread would otherwise live across the time loop
  const int kq = __builtin_amdgcn_readfirstlane(tid >> 8), jv = tid & (RV - 1);
  const int k0 = kq * KPQ;
  // (buffer loads: resource = the k-slice in scalar registers, one VGPR of column offset, the row as scalar offset)
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(W)) + (size_t)k0 * RV * 16, 0, KPQ * RV * 16, 0x00020000);
  const int voff = jv * 16;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  auto ldw = [&](int i) -> float4 {
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, voff, i * (RV * 16), 0));
    return make_float4(v[0], v[1], v[2], v[3]);
  };

  float4 wr[RK];
#pragma unroll
  for (int i = 0; i < RK; ++i) wr[i] = ldw(i);
#pragma unroll
  for (int i = 0; i < LK; ++i) wl[i * 1024 + tid] = ldw(RK + i);

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
    // resident parts first, then the streamed rest in batches of DEPTH row-vectors (a rolled loop: DEPTH loads in flight
    // per thread, 16 waves -> ~100 KB in flight per CU).  The accumulation order is k ascending, as in lstm_kernel.
    constexpr int NS = KPQ - RK - LK;
    static_assert(NS % DEPTH == 0, "whole batches");
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < RK; ++i) {
      const float hk = h_s[k0 + i];
      a.x = fmaf(wr[i].x, hk, a.x);
      a.y = fmaf(wr[i].y, hk, a.y);
      a.z = fmaf(wr[i].z, hk, a.z);
      a.w = fmaf(wr[i].w, hk, a.w);
    }
#pragma unroll 2
    for (int i = 0; i < LK; ++i) {
      const float4 wv = wl[i * 1024 + tid];
      const float hk = h_s[k0 + RK + i];
      a.x = fmaf(wv.x, hk, a.x);
      a.y = fmaf(wv.y, hk, a.y);
      a.z = fmaf(wv.z, hk, a.z);
      a.w = fmaf(wv.w, hk, a.w);
    }
#pragma unroll 1
    for (int j0 = 0; j0 < NS; j0 += DEPTH) {
      float4 ws[DEPTH];
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) ws[i] = ldw(RK + LK + j0 + i);
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) {
        const float hk = h_s[k0 + RK + LK + j0 + i];
        a.x = fmaf(ws[i].x, hk, a.x);
        a.y = fmaf(ws[i].y, hk, a.y);
        a.z = fmaf(ws[i].z, hk, a.z);
        a.w = fmaf(ws[i].w, hk, a.w);
      }
    }
    reinterpret_cast<float4*>(part + (long long)kq * G)[jv] = a;
    __syncthreads();
    if (tid < H) {
      float gi = xi, gf = xf, gg = xg, go = xo;
#pragma unroll
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
      if constexpr (TRAIN) {
        float* gr = gates + row * ldx + (long long)dir * G;
        gr[tid] = ig;
        gr[H + tid] = fg;
        gr[2 * H + tid] = cg;
        gr[3 * H + tid] = og;
        cell[row * ((long long)ndir * H) + dir * H + tid] = c_state;
      }
    }
    __syncthreads();
  }
}


// launches lstm_h256_kernel<RK, LK, DEPTH, TRAIN> with its LDS reservation; returns false when the reservation fails
template <int RK, int LK, int DEPTH, bool TRAIN>
static bool lstm_h256_launch(const float* d_xproj, const float* d_whh_t, int ndir, unsigned reverse_mask,
                             const int64_t* d_seq_rows, int nseq, float* d_out, long long ldo, int out_col0, float* d_gates,
                             float* d_cell, hipStream_t stream) {
  const size_t shm = ((size_t)256 + 4 * 1024) * sizeof(float) + (size_t)LK * 1024 * sizeof(float4);
  // per launch: the attribute belongs to the current device's copy of the kernel
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_h256_kernel<RK, LK, DEPTH, TRAIN>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
    return false;
  hipLaunchKernelGGL((lstm_h256_kernel<RK, LK, DEPTH, TRAIN>), dim3(nseq, ndir), dim3(1024), shm, stream, d_xproj, d_whh_t,
                     ndir, reverse_mask, d_seq_rows, d_out, ldo, out_col0, d_gates, d_cell);
  return true;
}
