// Probe: operand / result layout of v_mfma_f32_16x16x32_f16 on gfx950 (A[i][k] = i + 100 k ... decoded from the result).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// D = A (16 x 32) . B (32 x 16).  Hypothesis: lane l holds A[l % 16][8 * (l / 16) + 0..7], B[8 * (l / 16) + 0..7][l % 16];
// D register e of lane l = D[4 * (l / 16) + e][l % 16].
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  h8 a, b;
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * (l / 16) + j;
    a[j] = (_Float16)((l % 16 == 3 && k == 5) ? 1.0f : 0.0f);     // A = e_3 e_5^T
    b[j] = (_Float16)((k == 5) ? (float)(l % 16 + 1) : 0.0f);     // B[5][c] = c + 1
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = acc[e];
}

int main() {
  float* d;
  hipMalloc(&d, 64 * 4 * 4);
  probe<<<1, 64>>>(d);
  float h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // expected: D[3][c] = c + 1, everything else 0  ->  lanes with 4 * (l / 16) <= 3 < 4 * (l / 16) + 4, i.e. l / 16 == 0, e == 3
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 4; ++e) {
      const int row = 4 * (l / 16) + e, col = l % 16;
      const float want = row == 3 ? (float)(col + 1) : 0.f;
      if (h[l * 4 + e] != want) {
        if (bad < 8) printf("lane %d e %d: got %g want %g\n", l, e, h[l * 4 + e], want);
        ++bad;
      }
    }
  printf(bad ? "LAYOUT HYPOTHESIS WRONG (%d mismatches)\n" : "layout as assumed (%d mismatches)\n", bad);
  return 0;
}
