// Probe: does v_mfma_f32_32x32x16_f16 keep fp16 DENORMAL inputs (the lo halves of the f16x2 format are often
// denormal: lo ~ 2^-11 |x|), and does the VALU conversion produce them?   hipcc --offload-arch=gfx950 -o probe ...
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void probe(const float* in, float* out) {
  const int l = threadIdx.x;
  // A[r][k] = a for k == 0 (lanes with l>>5 == 0, element 0), else 0;  B[c][k] = b likewise  -> C[r][c] = a*b
  const float a = in[0], b = in[1];
  h8 av = {0, 0, 0, 0, 0, 0, 0, 0}, bv = av;
  if ((l >> 5) == 0) {
    av[0] = (_Float16)a;
    bv[0] = (_Float16)b;
  }
  f16v acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
  if (l == 0) {
    out[0] = acc[0];
    out[1] = (float)(_Float16)a;   // VALU round trip
    out[2] = (float)(_Float16)b;
  }
}

int main() {
  float *din, *dout;
  hipMalloc(&din, 8);
  hipMalloc(&dout, 12);
  const float cases[][2] = {{1.0f, 1.0f}, {3.0e-6f, 1.0f}, {3.0e-6f, 1024.0f}, {6.0e-8f, 4096.0f}, {2.0e-5f, 2.0e-5f},
                            {1.0e-7f, 1.0f}};
  for (auto& c : cases) {
    hipMemcpy(din, c, 8, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(din, dout);
    float h[3];
    hipMemcpy(h, dout, 12, hipMemcpyDeviceToHost);
    printf("a=%.4e b=%.4e  f16(a)=%.6e f16(b)=%.6e  mfma a*b=%.6e  expected=%.6e\n", c[0], c[1], h[1], h[2], h[0],
           (double)h[1] * (double)h[2]);
  }
  return 0;
}
