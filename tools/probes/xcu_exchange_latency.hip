// Probe: the per-step cost of an all-to-all exchange of 256 floats among 4 workgroups (one per CU) through tagged 8-byte
// granules {value, step} in global memory, agent-scope relaxed atomics - the exchange a recurrence split over 4 CUs would
// pay every time step (each CU owns 64 of the 256 hidden units, W_hh's quarter resident in its registers).
//   hipcc -O3 --offload-arch=gfx950 -o xcu_exchange_latency tools/probes/xcu_exchange_latency.hip && ./xcu_exchange_latency
// placement 0: the 4 partners are consecutive block ids (4 different XCDs); 1: block ids 8 apart (the same XCD);
// 2, 3: the same two with PIPELINED polling (four loads in flight instead of one at a time): measured 8 - 15 % SLOWER
// (0.65 against 0.57 us with one group on an XCD) - one load at a time is what lstm_split.hip does
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int PARTS>
__global__ __launch_bounds__(256) void exch(unsigned long long* g, int steps, int placement, int groups, unsigned* err, float* out) {
  const int b = blockIdx.x;
  int grp, part;
  if ((placement & 1) == 0) {
    grp = b / PARTS;
    part = b % PARTS;
  } else {   // 8 * PARTS consecutive ids hold 8 groups; a group's parts are 8 ids apart
    grp = (b % 8) + 8 * (b / (8 * PARTS));
    part = (b / 8) % PARTS;
  }
  if (grp >= groups) return;
  constexpr int PER = 256 / PARTS;
  __shared__ float h_s[256];
  __shared__ int bail;
  if (threadIdx.x == 0) bail = 0;
  __syncthreads();
  unsigned long long* base = g + (long long)grp * 2 * 256;
  const int tid = threadIdx.x;
  float v = (float)tid;
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    unsigned long long* slot = base + (s & 1) * 256;
    if (tid < PER) {
      const unsigned long long gr = ((unsigned long long)(unsigned)(s + 1) << 32) | (unsigned long long)__float_as_uint(v + (float)s);
      __hip_atomic_store(slot + part * PER + tid, gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned long long q;
    int spin = 0;
    if (placement < 2) {
      q = __hip_atomic_load(slot + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while ((unsigned)(q >> 32) != (unsigned)(s + 1) && spin < (1 << 20)) {
        q = __hip_atomic_load(slot + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++spin;
      }
    } else {
      // pipelined polling: four loads in flight, a new one issued as the oldest is examined
      unsigned long long r0 = __hip_atomic_load(slot + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_sleep(1);
      unsigned long long r1 = __hip_atomic_load(slot + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_sleep(1);
      unsigned long long r2 = __hip_atomic_load(slot + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_sleep(1);
      unsigned long long r3 = __hip_atomic_load(slot + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      q = r0;
      while ((unsigned)(q >> 32) != (unsigned)(s + 1) && spin < (1 << 20)) {
        r0 = r1; r1 = r2; r2 = r3;
        r3 = __hip_atomic_load(slot + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q = r0;
        ++spin;
      }
    }
    if ((unsigned)(q >> 32) != (unsigned)(s + 1)) {
      atomicAdd(err, 1u);
      bail = 1;
    }
    h_s[tid] = __uint_as_float((unsigned)q);
    __syncthreads();
    if (bail) break;   // (uniform: read after the barrier) a partner never showed up - leave, the others time out once too
    acc += h_s[(tid * 7 + s) & 255];
    v = acc * 1e-9f + (float)tid;
    __syncthreads();
  }
  if (tid == 0) out[b] = acc;
}

int main() {
  const int steps = 4000;
  unsigned long long* g;
  unsigned* err;
  float* out;
  hipMalloc(&g, 64 * 2 * 256 * 8);
  hipMalloc(&err, 4);
  hipMalloc(&out, 4096 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int parts = 2; parts <= 8; parts *= 2)
    for (int placement = 0; placement < 4; ++placement)
      for (int groups : {1, 4, 16}) {
        if ((placement & 1) == 1 && groups < 8 && groups != 1) continue;
        hipMemset(g, 0, 64 * 2 * 256 * 8);
        hipMemset(err, 0, 4);
        const int nb = (placement & 1) == 0 ? groups * parts : ((groups + 7) / 8) * 8 * parts;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          hipMemset(g, 0, 64 * 2 * 256 * 8);
          hipEventRecord(e0);
          if (parts == 2) hipLaunchKernelGGL(exch<2>, dim3(nb), dim3(256), 0, 0, g, steps, placement, groups, err, out);
          if (parts == 4) hipLaunchKernelGGL(exch<4>, dim3(nb), dim3(256), 0, 0, g, steps, placement, groups, err, out);
          if (parts == 8) hipLaunchKernelGGL(exch<8>, dim3(nb), dim3(256), 0, 0, g, steps, placement, groups, err, out);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
        }
        unsigned herr;
        hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        printf("parts %d placement %d groups %2d: %.3f us per step (timeouts %u)\n", parts, placement, groups, best * 1e3f / steps, herr);
      }
  return 0;
}
