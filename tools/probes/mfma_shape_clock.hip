// Probe: v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 on RANDOM operands read from LDS - the same FLOPs per
// wave and step, the same LDS traffic.  The guide (MI355X_MICROARCH.md, DVFS item 7) reports ~1.12-1.15x the FLOP/s for
// the 16x16x32 shape in bf16 because the power-limited chip holds a higher clock on it; is that so for f16 and for a
// loop shaped like local224's (operands re-read from LDS by ds_read_b128 every step)?
//   hipcc -O3 --offload-arch=gfx950 -o mfma_shape_clock tools/probes/mfma_shape_clock.hip && ./mfma_shape_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LDS_BYTES = 48 * 1024;

// a wave owns a 128-row x 32-column tile: 32x32x16 -> 4 x 1 tiles, per 16-deep step 4 A + 1 B fragment reads, 4 MFMAs;
// 16x16x32 -> 8 x 2 tiles, per 32-deep step 8 A + 2 B reads, 16 MFMAs: equal flops per K and equal reads per K
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void loop(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ uint4 lds[LDS_BYTES / 16];
  for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) lds[i] = src[(blockIdx.x * 131 + i) % (1 << 16)];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float total = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int base = ((it * 8 + s) * 67 + wave * 13) & 1023;
        const uint4 fb = lds[(base + lane) & (LDS_BYTES / 16 - 1)];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const uint4 fa = lds[(base + 64 * (m + 1) + lane) & (LDS_BYTES / 16 - 1)];
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, fa), __builtin_bit_cast(h8, fb), acc[m], 0, 0, 0);
        }
      }
    }
    for (int i = 0; i < 4; ++i)
      for (int e = 0; e < 16; ++e) total += acc[i][e];
  } else {
    f32x4 acc[8][2];
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 2; ++j)
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int base = ((it * 4 + s) * 67 + wave * 13) & 1023;
        uint4 fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = lds[(base + 64 * j + lane) & (LDS_BYTES / 16 - 1)];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const uint4 fa = lds[(base + 64 * (m + 2) + lane) & (LDS_BYTES / 16 - 1)];
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, fa), __builtin_bit_cast(h8, fb[j]), acc[m][j], 0, 0, 0);
        }
      }
    }
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 2; ++j)
        for (int e = 0; e < 4; ++e) total += acc[i][j][e];
  }
  out[blockIdx.x * 256 + threadIdx.x] = total;
}

int main() {
  const int blocks = 2048, iters = 4000;
  uint4* src;
  float* out;
  hipMalloc(&src, (1 << 16) * 16);
  hipMalloc(&out, blocks * 256 * 4);
  // random fp16 values of moderate size (no NaN / Inf): exponent field 0x30..0x3f
  unsigned short* h = (unsigned short*)malloc((1 << 16) * 16);
  srand(1);
  for (int i = 0; i < (1 << 16) * 8; ++i) h[i] = (unsigned short)(((rand() & 1) << 15) | ((0x30 + (rand() & 7)) << 10) | (rand() & 1023));
  hipMemcpy(src, h, (1 << 16) * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    for (int shape : {32, 16}) {
      // per wave and iteration: 32 MFMAs x 32768 flop (32x32x16) = 64 x 16384 (16x16x32)
      const double flop = (double)blocks * 4 * iters * 32.0 * 32768.0;
      hipEventRecord(e0);
      if (shape == 32)
        loop<32><<<blocks, 256>>>(src, out, iters);
      else
        loop<16><<<blocks, 256>>>(src, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("rep %d  mfma %s: %8.2f ms  %8.1f TFLOP/s\n", rep, shape == 32 ? "32x32x16" : "16x16x32", ms, flop / ms / 1e9);
    }
  }
  return 0;
}
