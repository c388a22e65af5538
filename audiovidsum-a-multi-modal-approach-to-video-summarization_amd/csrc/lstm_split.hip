// The hidden = 256 LSTM recurrence of the scorer (models/av_model.py:39-40: nn.LSTM(512, 256, bidirectional=True)) and its
// backward sweep (scripts/train_av_model.py:94-95) with ONE recurrence SPLIT OVER FOUR CUs.
//
// One CU cannot hold W_hh (1 MB of fp32) on chip: lstm_h256_kernel keeps 28 of a thread's 64 row-vectors in registers / LDS
// and streams the other 36 from L2 every time step - 590 KB per step, which IS its 5.2 us step.  Here the four workgroups of
// a recurrence own 64 hidden units each: the four gate columns of its units (forward) / the columns of W_hh under its units
// (backward) are 256 KB = 128 registers of each of its 512 threads, loaded once.  A step then costs the product out of
// registers (128 fmaf per thread) plus ONE exchange of the step's vector among the four CUs - h_t (256 floats) forward, the
// gate gradients (1024 floats) backward - through tagged 8-byte granules {value, tag} in global memory: one agent-scope
// store per value, untorn, so the tag IS the flag (no fence, no separate flag word); a reader polls the granule until its
// tag is this step's.  Two slots per recurrence (step parity): a CU is never more than one step ahead of its partners.
// tools/probes/xcu_exchange_latency.hip: such an exchange costs 0.85 us per step with the partners on one XCD, 1.04 us
// across XCDs.
//
// Arithmetic: the same fmaf chains in the same order as lstm_kernel / lstm_h256_kernel / lstm_bwd_h256_kernel (k ascending
// inside a slice, slices ascending) - bit-identical outputs.
// Safety: the four workgroups of a recurrence are 8 block ids apart inside 32 consecutive ids (dispatched together; blocks
// 8 apart share an XCD under round-robin dispatch - speed only); every wait is bounded, a workgroup whose wait ran out
// counts itself into the workspace's error word and leaves (its partners then run out once too): the launch always ends.
#include "avs_internal.h"
#include "lstm_h256.h"
#include <math.h>

namespace {
constexpr int SP_H = 256, SP_G = 1024, SP_PARTS = 4, SP_UNITS = SP_H / SP_PARTS;   // 64 hidden units per workgroup
constexpr int SP_SLOT = 1024;               // granules per step parity (forward uses the first 256)
constexpr int SP_SPIN = 1 << 20;            // bounded wait: 2^20 polls, of the order of a second

__device__ __forceinline__ void sp_map(int b, int& rec, int& part) {
  // 32 consecutive blocks = 8 recurrences x 4 parts; a recurrence's parts are 8 ids apart
  rec = (b & 7) + 8 * (b >> 5);
  part = (b >> 3) & 3;
}

__device__ __forceinline__ unsigned long long sp_pack(float v, unsigned tag) {
  return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}

// polls granule g until its tag is `tag`; false when the bounded wait ran out
__device__ __forceinline__ bool sp_wait(const unsigned long long* g, unsigned tag, float& v) {
  unsigned long long q = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int spin = 0; (unsigned)(q >> 32) != tag && spin < SP_SPIN; ++spin) {
    __builtin_amdgcn_s_sleep(1);
    q = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  v = __uint_as_float((unsigned)q);
  return (unsigned)(q >> 32) == tag;
}

// ---- forward (inference; TRAIN: the gates and the cell state are kept for the backward sweep) ----
// thread (kq, jv2): reduction slice k in [64 kq, 64 kq + 64), gate columns c0, c0 + 1 with c0 = gate * 256 + 64 part + 2 pr
template <bool TRAIN>
__global__ __launch_bounds__(512) void lstm_split_fwd_kernel(const float* __restrict__ xproj, const float* __restrict__ whh_t,
                                                             int ndir, unsigned reverse_mask,
                                                             const int64_t* __restrict__ seq_rows, int nrec,
                                                             float* __restrict__ out, long long ldo, int out_col0,
                                                             float* __restrict__ gates, float* __restrict__ cell,
                                                             unsigned long long* __restrict__ xchg, unsigned epoch,
                                                             unsigned* __restrict__ err) {
  int rec, part;
  sp_map(blockIdx.x, rec, part);
  if (rec >= nrec) return;
  const int seq = rec / ndir, dir = rec - seq * ndir;
  const long long r0 = seq_rows[seq], r1 = seq_rows[seq + 1];
  const long long T = r1 - r0;
  const bool rev = (reverse_mask >> dir) & 1u;
  const float* __restrict__ W = whh_t + (long long)dir * SP_H * SP_G;
  const long long ldx = (long long)ndir * SP_G;
  const float* __restrict__ xp = xproj + (long long)dir * SP_G;

  __shared__ float h_s[SP_H];
  __shared__ float part_s[4 * SP_H];   // [kq][gate * 64 + unit]
  __shared__ float act_s[SP_H];        // [gate][unit]: the step's post-activation gates of this workgroup's units
  __shared__ int bail;

  const int tid = threadIdx.x;
  const int kq = tid >> 7, jv2 = tid & 127;
  const int gate = jv2 >> 5, pr = jv2 & 31;
  const int c0 = gate * SP_H + SP_UNITS * part + 2 * pr;
  float2 w[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) w[i] = *reinterpret_cast<const float2*>(W + (long long)(64 * kq + i) * SP_G + c0);

  // the gate activations of a step run on FOUR waves at once (wave gw: gate gw of the 64 units), the cell update on one
  const int gw = __builtin_amdgcn_readfirstlane(tid >> 6) & 3, gu = tid & 63;
  const int gcol = gw * SP_H + SP_UNITS * part + gu;   // threads < 256: the gate column this thread activates
  const int unit = SP_UNITS * part + tid;              // threads < 64: the hidden unit this thread finishes
  float c_state = 0.f;
  if (tid < SP_H) h_s[tid] = 0.f;
  if (tid == 0) bail = 0;
  __syncthreads();
  unsigned long long* const slots = xchg + (long long)rec * 2 * SP_SLOT;

  for (long long s = 0; s < T; ++s) {
    const long long row = rev ? (r1 - 1 - s) : (r0 + s);
    float xv = 0.f;
    if (tid < SP_H) xv = xp[row * ldx + gcol];
    float ax = 0.f, ay = 0.f;
    const float4* h4 = reinterpret_cast<const float4*>(h_s + 64 * kq);
#pragma unroll
    for (int i4 = 0; i4 < 16; ++i4) {
      const float4 hk = h4[i4];
      ax = fmaf(w[4 * i4].x, hk.x, ax);
      ay = fmaf(w[4 * i4].y, hk.x, ay);
      ax = fmaf(w[4 * i4 + 1].x, hk.y, ax);
      ay = fmaf(w[4 * i4 + 1].y, hk.y, ay);
      ax = fmaf(w[4 * i4 + 2].x, hk.z, ax);
      ay = fmaf(w[4 * i4 + 2].y, hk.z, ay);
      ax = fmaf(w[4 * i4 + 3].x, hk.w, ax);
      ay = fmaf(w[4 * i4 + 3].y, hk.w, ay);
    }
    *reinterpret_cast<float2*>(part_s + kq * SP_H + gate * SP_UNITS + 2 * pr) = make_float2(ax, ay);
    __syncthreads();
    const unsigned tag = epoch + (unsigned)s + 1u;
    unsigned long long* const slot = slots + (s & 1) * SP_SLOT;
    if (tid < SP_H) {
      float pre = xv;   // x + the four slices' partial sums, slices ascending (as lstm_kernel adds them)
#pragma unroll
      for (int q = 0; q < 4; ++q) pre += part_s[q * SP_H + gw * SP_UNITS + gu];
      const float act = gw == 2 ? tanhf(pre) : avs_sigmoid(pre);
      act_s[gw * SP_UNITS + gu] = act;
      if constexpr (TRAIN) gates[row * ldx + (long long)dir * SP_G + gcol] = act;
    }
    __syncthreads();
    if (tid < SP_UNITS) {
      const float ig = act_s[tid], fg = act_s[SP_UNITS + tid], cg = act_s[2 * SP_UNITS + tid], og = act_s[3 * SP_UNITS + tid];
      c_state = fg * c_state + ig * cg;
      const float hv = og * tanhf(c_state);
      __hip_atomic_store(slot + unit, sp_pack(hv, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      out[row * ldo + out_col0 + dir * SP_H + unit] = hv;
      if constexpr (TRAIN) cell[row * ((long long)ndir * SP_H) + dir * SP_H + unit] = c_state;
    }
    if (s + 1 < T && tid < SP_H) {   // the whole h_t, from the four owners (this workgroup's own 64 values included)
      float v;
      if (!sp_wait(slot + tid, tag, v)) bail = 1;
      h_s[tid] = v;
    }
    __syncthreads();
    if (bail) {   // (uniform: read behind the barrier) a partner never showed up
      if (tid == 0) atomicAdd(err, 1u);
      return;
    }
  }
}

// ---- backward ----
// thread (js, kv2): rows j in [64 js, 64 js + 64) of W_hh [4H, H], columns k0, k0 + 1 with k0 = 64 part + 2 kv2
__global__ __launch_bounds__(512) void lstm_split_bwd_kernel(const float* __restrict__ dout, long long ldo, int out_col0,
                                                             const float* __restrict__ gates, const float* __restrict__ cell,
                                                             const float* __restrict__ whh, int ndir, unsigned reverse_mask,
                                                             const int64_t* __restrict__ seq_rows, int nrec,
                                                             float* __restrict__ dxproj, unsigned long long* __restrict__ xchg,
                                                             unsigned epoch, unsigned* __restrict__ err) {
  int rec, part;
  sp_map(blockIdx.x, rec, part);
  if (rec >= nrec) return;
  const int seq = rec / ndir, dir = rec - seq * ndir;
  const long long r0 = seq_rows[seq], r1 = seq_rows[seq + 1];
  const long long T = r1 - r0;
  const bool rev = (reverse_mask >> dir) & 1u;
  const long long ldg = (long long)ndir * SP_G, ldc = (long long)ndir * SP_H;
  const float* __restrict__ W = whh + (long long)dir * SP_G * SP_H;

  __shared__ float da_s[SP_G];
  __shared__ float part_s[16 * SP_UNITS];   // [js][k - 64 part]
  __shared__ int bail;

  const int tid = threadIdx.x;
  const int js = tid >> 5, kv2 = tid & 31;
  const int k0 = SP_UNITS * part + 2 * kv2;
  float2 w[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) w[i] = *reinterpret_cast<const float2*>(W + (long long)(64 * js + i) * SP_H + k0);

  const int unit = SP_UNITS * part + tid;   // threads < 64
  const float* __restrict__ gbase = gates + (long long)dir * SP_G + unit;
  const float* __restrict__ cbase = cell + dir * SP_H + unit;
  const float* __restrict__ dbase = dout + out_col0 + dir * SP_H + unit;
  auto row_of = [&](long long s) -> long long { return rev ? (r1 - 1 - s) : (r0 + s); };
  float p_i = 0.f, p_f = 0.f, p_g = 0.f, p_o = 0.f, p_c = 0.f, p_cprev = 0.f, p_d = 0.f;
  if (tid < SP_UNITS && T > 0) {
    const long long row = row_of(T - 1);
    p_i = gbase[row * ldg];
    p_f = gbase[row * ldg + SP_H];
    p_g = gbase[row * ldg + 2 * SP_H];
    p_o = gbase[row * ldg + 3 * SP_H];
    p_c = cbase[row * ldc];
    p_d = dbase[row * ldo];
    if (T > 1) p_cprev = cbase[row_of(T - 2) * ldc];
  }
  float p_tc = tanhf(p_c);   // tanh of the step's cell state: computed a step ahead, under the exchange's wait
  if (tid == 0) bail = 0;
  __syncthreads();
  unsigned long long* const slots = xchg + (long long)rec * 2 * SP_SLOT;
  float dc_next = 0.f, dh_next = 0.f;
  for (long long s = T - 1; s >= 0; --s) {
    const long long row = row_of(s);
    const long long n = T - s;                      // 1, 2, ...: this sweep's step count
    const unsigned tag = epoch + (unsigned)n;
    unsigned long long* const slot = slots + (n & 1) * SP_SLOT;
    if (tid < SP_UNITS) {
      const float ig = p_i, fg = p_f, cg = p_g, og = p_o, c_prev = s > 0 ? p_cprev : 0.f;
      const float tc = p_tc;   // tanh(cell state of this step)
      const float dh = p_d + dh_next;
      const float d_o = dh * tc;
      const float dc = dh * og * (1.f - tc * tc) + dc_next;
      const float d_i = dc * cg, d_g = dc * ig, d_f = dc * c_prev;
      dc_next = dc * fg;
      const float ai = d_i * ig * (1.f - ig), af = d_f * fg * (1.f - fg), ag = d_g * (1.f - cg * cg),
                  ao = d_o * og * (1.f - og);
      __hip_atomic_store(slot + unit, sp_pack(ai, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(slot + SP_H + unit, sp_pack(af, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(slot + 2 * SP_H + unit, sp_pack(ag, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(slot + 3 * SP_H + unit, sp_pack(ao, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float* dx = dxproj + row * ldg + (long long)dir * SP_G;
      dx[unit] = ai;
      dx[SP_H + unit] = af;
      dx[2 * SP_H + unit] = ag;
      dx[3 * SP_H + unit] = ao;
      if (s > 0) {   // the next step's inputs, in flight under this step's exchange and product
        const long long nrow = row_of(s - 1);
        p_i = gbase[nrow * ldg];
        p_f = gbase[nrow * ldg + SP_H];
        p_g = gbase[nrow * ldg + 2 * SP_H];
        p_o = gbase[nrow * ldg + 3 * SP_H];
        p_d = dbase[nrow * ldo];
        p_c = p_cprev;
        if (s > 1) p_cprev = cbase[row_of(s - 2) * ldc];
        p_tc = tanhf(p_c);   // (p_c arrived a step ago: no memory wait here)
      }
    }
    if (s > 0) {   // (the last step's product would feed nothing)
      // both granules of this thread in flight at once, then each is verified (and polled on if its tag is not there yet)
      unsigned long long q0 = __hip_atomic_load(slot + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long q1 = __hip_atomic_load(slot + 512 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float v0 = __uint_as_float((unsigned)q0), v1 = __uint_as_float((unsigned)q1);
      bool ok0 = (unsigned)(q0 >> 32) == tag, ok1 = (unsigned)(q1 >> 32) == tag;
      if (!ok0) ok0 = sp_wait(slot + tid, tag, v0);
      if (!ok1) ok1 = sp_wait(slot + 512 + tid, tag, v1);
      if (!(ok0 && ok1)) bail = 1;
      da_s[tid] = v0;
      da_s[512 + tid] = v1;
    }
    __syncthreads();
    if (bail) {
      if (tid == 0) atomicAdd(err, 1u);
      return;
    }
    if (s == 0) break;
    float ax = 0.f, ay = 0.f;
    const float4* d4 = reinterpret_cast<const float4*>(da_s + 64 * js);
#pragma unroll
    for (int i4 = 0; i4 < 16; ++i4) {
      const float4 dj = d4[i4];
      ax = fmaf(w[4 * i4].x, dj.x, ax);
      ay = fmaf(w[4 * i4].y, dj.x, ay);
      ax = fmaf(w[4 * i4 + 1].x, dj.y, ax);
      ay = fmaf(w[4 * i4 + 1].y, dj.y, ay);
      ax = fmaf(w[4 * i4 + 2].x, dj.z, ax);
      ay = fmaf(w[4 * i4 + 2].y, dj.z, ay);
      ax = fmaf(w[4 * i4 + 3].x, dj.w, ax);
      ay = fmaf(w[4 * i4 + 3].y, dj.w, ay);
    }
    *reinterpret_cast<float2*>(part_s + js * SP_UNITS + 2 * kv2) = make_float2(ax, ay);
    __syncthreads();
    if (tid < SP_UNITS) {
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc += part_s[q * SP_UNITS + tid];
      dh_next = acc;
    }
    // da_s is rewritten behind the next step's polls (every thread is past this barrier), part_s behind its first barrier
  }
}

int sp_check(const char* who, int hidden, int ndir, int nseq, int64_t ldo, int out_col0, const void* d_ws, size_t ws_bytes) {
  AVS_REQUIRE(hidden == SP_H, AVS_E_UNSUPPORTED, "%s: the split recurrence is built for hidden = 256", who);
  AVS_REQUIRE(ndir > 0 && ndir <= 32 && nseq >= 0 && out_col0 >= 0 && ldo >= out_col0 + (int64_t)ndir * hidden, AVS_E_SHAPE,
              "%s: bad extents", who);
  AVS_REQUIRE((long long)nseq * ndir <= 4096, AVS_E_SHAPE, "%s: at most 4096 recurrences per launch", who);
  AVS_REQUIRE(nseq == 0 || (d_ws && (((uintptr_t)d_ws) & 7u) == 0 && ws_bytes >= avs_lstm_split_workspace_bytes(ndir, nseq)),
              AVS_E_WORKSPACE, "%s: workspace of %zu bytes needed (8-byte aligned)", who, avs_lstm_split_workspace_bytes(ndir, nseq));
  return AVS_OK;
}
}  // namespace

// the error word (64 bytes) + [recurrence][step parity][1024] granules
extern "C" size_t avs_lstm_split_workspace_bytes(int ndir, int nseq) {
  if (ndir <= 0 || nseq <= 0) return 0;
  return (size_t)ndir * nseq * 2 * SP_SLOT * sizeof(unsigned long long) + 64;
}

extern "C" int avs_lstm_split_f32(const float* d_xproj, const float* d_whh_t, int hidden, int ndir, unsigned reverse_mask,
                                  const int64_t* d_seq_rows, int nseq, float* d_out, int64_t ldo, int out_col0,
                                  float* d_gates, float* d_cell, void* d_ws, size_t ws_bytes, unsigned epoch,
                                  avs_stream_t stream) {
  const char* who = "avs_lstm_split_f32";
  int st = sp_check(who, hidden, ndir, nseq, ldo, out_col0, d_ws, ws_bytes);
  if (st != AVS_OK) return st;
  if (nseq == 0) return AVS_OK;
  AVS_REQUIRE(d_xproj && d_whh_t && d_seq_rows && d_out, AVS_E_ARG, "%s: null pointer", who);
  AVS_REQUIRE((d_gates == nullptr) == (d_cell == nullptr), AVS_E_ARG, "%s: gates and cell go together", who);
  AVS_REQUIRE((((uintptr_t)d_whh_t) & 7u) == 0, AVS_E_ALIGN, "%s: whh_t not 8-byte aligned", who);
  const int nrec = nseq * ndir;
  unsigned* err = static_cast<unsigned*>(d_ws);   // the workspace's first 64 bytes: the error word
  unsigned long long* xchg = reinterpret_cast<unsigned long long*>(static_cast<char*>(d_ws) + 64);
  const dim3 grid((unsigned)((nrec + 7) / 8 * 32));
  if (d_gates)
    hipLaunchKernelGGL(lstm_split_fwd_kernel<true>, grid, dim3(512), 0, (hipStream_t)stream, d_xproj, d_whh_t, ndir,
                       reverse_mask, d_seq_rows, nrec, d_out, (long long)ldo, out_col0, d_gates, d_cell, xchg, epoch, err);
  else
    hipLaunchKernelGGL(lstm_split_fwd_kernel<false>, grid, dim3(512), 0, (hipStream_t)stream, d_xproj, d_whh_t, ndir,
                       reverse_mask, d_seq_rows, nrec, d_out, (long long)ldo, out_col0, nullptr, nullptr, xchg, epoch, err);
  AVS_CHECK_LAUNCH(who);
  return AVS_OK;
}

extern "C" int avs_lstm_bwd_split_f32(const float* d_dout, int64_t ldo, int out_col0, const float* d_gates,
                                      const float* d_cell, const float* d_whh, int hidden, int ndir, unsigned reverse_mask,
                                      const int64_t* d_seq_rows, int nseq, float* d_dxproj, void* d_ws, size_t ws_bytes,
                                      unsigned epoch, avs_stream_t stream) {
  const char* who = "avs_lstm_bwd_split_f32";
  int st = sp_check(who, hidden, ndir, nseq, ldo, out_col0, d_ws, ws_bytes);
  if (st != AVS_OK) return st;
  if (nseq == 0) return AVS_OK;
  AVS_REQUIRE(d_dout && d_gates && d_cell && d_whh && d_seq_rows && d_dxproj, AVS_E_ARG, "%s: null pointer", who);
  AVS_REQUIRE((((uintptr_t)d_whh) & 7u) == 0, AVS_E_ALIGN, "%s: whh not 8-byte aligned", who);
  const int nrec = nseq * ndir;
  unsigned* err = static_cast<unsigned*>(d_ws);   // the workspace's first 64 bytes: the error word
  unsigned long long* xchg = reinterpret_cast<unsigned long long*>(static_cast<char*>(d_ws) + 64);
  const dim3 grid((unsigned)((nrec + 7) / 8 * 32));
  hipLaunchKernelGGL(lstm_split_bwd_kernel, grid, dim3(512), 0, (hipStream_t)stream, d_dout, (long long)ldo, out_col0, d_gates,
                     d_cell, d_whh, ndir, reverse_mask, d_seq_rows, nrec, d_dxproj, xchg, epoch, err);
  AVS_CHECK_LAUNCH(who);
  return AVS_OK;
}
