// Internal helpers shared by the gfx950 kernels of libavsum_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/avsum_hip.h"

#define AVS_WAVE 64

void avs_set_error(const char* fmt, ...);

#define AVS_REQUIRE(cond, code, ...)      \
  do {                                    \
    if (!(cond)) {                        \
      avs_set_error(__VA_ARGS__);         \
      return (code);                      \
    }                                     \
  } while (0)

// Launch-error check: kernel launches are asynchronous; this only catches
// configuration errors, which is what the ABI promises.
#define AVS_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      avs_set_error("%s: HIP launch failed: %s", name, hipGetErrorString(e__)); \
      return AVS_E_HIP;                                                     \
    }                                                                       \
  } while (0)

static inline bool avs_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int64_t avs_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef __bf16 avs_bf16;

__device__ __forceinline__ float avs_bf16_to_f32(unsigned short u) {
  return __uint_as_float(((unsigned)u) << 16);
}
__device__ __forceinline__ unsigned short avs_f32_to_bf16(float f) {
  avs_bf16 b = (avs_bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(unsigned short, b);
}

// two values -> packed bf16 (lo in bits 0..15): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned avs_pack_bf16x2(float lo, float hi) {
  return (unsigned)avs_f32_to_bf16(lo) | ((unsigned)avs_f32_to_bf16(hi) << 16);
}
// ReLU on a packed bf16 pair: as signed 16-bit integers the negative values (and -0) are exactly the negative
// integers, so max(x, 0) per half is relu, bit for bit what fmaxf(v, 0) before the rounding gives (the rounding is
// monotone and keeps the sign): one v_pk_max_i16 for two values
__device__ __forceinline__ unsigned avs_relu_bf16x2(unsigned v) {
  typedef short avs_s16x2 __attribute__((ext_vector_type(2)));
  const avs_s16x2 z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(avs_s16x2, v), z));
}

// ---- AVS_F16X2: a value as TWO fp16, x ~ hi + lo (hi = fp16(x), lo = fp16(x - hi): 22 significant bits, absolute
// floor 2^-25 from the fp16 denormals), stored in 4-byte slots like fp32: every aligned run of 8 slots (32 bytes)
// holds the 8 hi halves, then the 8 lo halves, of 8 consecutive elements.  Magnitudes saturate at 65504.
typedef _Float16 avs_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 avs_f16x2v __attribute__((ext_vector_type(2)));
#define AVS_F16_MAX 65504.0f

__device__ __forceinline__ unsigned avs_pack_f16x2(float a, float b) {
  const avs_f16x2v h = {(_Float16)a, (_Float16)b};   // v_cvt_f16_f32: RNE
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void avs_unpack_f16x2(unsigned u, float& a, float& b) {
  const avs_f16x2v h = __builtin_bit_cast(avs_f16x2v, u);
  a = (float)h[0];
  b = (float)h[1];
}
// 8 fp32 -> the 16 bytes of hi halves and the 16 bytes of lo halves
__device__ __forceinline__ void avs_f16x2_split8(const float (&v)[8], uint4& hi, uint4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = fminf(fmaxf(v[2 * j], -AVS_F16_MAX), AVS_F16_MAX);
    const float b = fminf(fmaxf(v[2 * j + 1], -AVS_F16_MAX), AVS_F16_MAX);
    h[j] = avs_pack_f16x2(a, b);
    float ha, hb;
    avs_unpack_f16x2(h[j], ha, hb);
#ifdef AVS_H2_DROP_LO
    // accuracy-study build only (make fp16emu): every value is stored as ONE fp16 - what a plain fp16-storage mode
    // (fp16 activations and weights, fp32 accumulation) computes, at the f16x2 kernels' speed-irrelevant cost
    l[j] = 0u;
#else
    l[j] = avs_pack_f16x2(a - ha, b - hb);
#endif
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ void avs_f16x2_join8(const uint4& hi, const uint4& lo, float (&v)[8]) {
  const unsigned h[4] = {hi.x, hi.y, hi.z, hi.w}, l[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float ha, hb, la, lb;
    avs_unpack_f16x2(h[j], ha, hb);
    avs_unpack_f16x2(l[j], la, lb);
    v[2 * j] = ha + la;
    v[2 * j + 1] = hb + lb;
  }
}
// ---- AVS_F16P8: fp16 hi + an 8-BIT remainder, 3 bytes per value (the inner block outputs of ResNet layers 1-2).
// x ~ hi + (u - 128) * step(hi), hi = fp16(x), step(hi) = ulp(max(|hi|, 2^-6)) / 256 = 2^(max(E, -6) - 18) with E the
// exponent of hi (a step of 2^-24 - the fp16 grid - below 2^-6), u = round((x - hi) / step) + 128 clamped to 1..255
// (u = 128 when hi = 0): 19-20 significant bits, the absolute floor of AVS_F16X2 (tools/h3_storage_study.py: the scores
// do not move).  Every remainder (u - 128) * step IS an fp16 number - exactly the lo half AVS_F16X2 holds for the same
// value - and is rebuilt with packed fp16 arithmetic: 3 VALU operations per value.  Every aligned run of 16 channels is
// 48 bytes: the 8 hi halves of channels 0-7, the 8 hi halves of channels 8-15, the 16 remainder bytes.
typedef unsigned short avs_u16x2 __attribute__((ext_vector_type(2)));
// the fp16 lo halves of two values from their hi halves (one dword) and their remainder bytes k, k + 1 of `rem`
__device__ __forceinline__ unsigned avs_f16p8_lo2(unsigned hi2, unsigned rem, int k) {
  // 2^max(E, -6) of each half (the exponent field alone; denormals and zero count as 2^-6), times 2^-18 (an fp16 denormal)
  const avs_u16x2 floor6 = {0x2400, 0x2400};
  const avs_u16x2 hp = __builtin_elementwise_max(__builtin_bit_cast(avs_u16x2, hi2 & 0x7C007C00u), floor6);
  const avs_f16x2v step = __builtin_bit_cast(avs_f16x2v, hp) * __builtin_bit_cast(avs_f16x2v, 0x00400040u);
  // 0x6400 | u = 1024 + u as fp16; minus 1152 = u - 128, exactly
  const unsigned q = __builtin_amdgcn_perm(rem, 0x64646464u, k == 0 ? 0x00050004u : 0x00070006u);
  const avs_f16x2v qh = __builtin_bit_cast(avs_f16x2v, q) + __builtin_bit_cast(avs_f16x2v, 0xE480E480u);
  return __builtin_bit_cast(unsigned, qh * step);
}
// 8 fp32 -> the 16 bytes of hi halves and the 8 remainder bytes
__device__ __forceinline__ void avs_f16p8_split8(const float (&v)[8], uint4& hi, uint2& rem) {
  unsigned h[4], r[2] = {0u, 0u};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = fminf(fmaxf(v[2 * j], -AVS_F16_MAX), AVS_F16_MAX);
    const float b = fminf(fmaxf(v[2 * j + 1], -AVS_F16_MAX), AVS_F16_MAX);
    h[j] = avs_pack_f16x2(a, b);
    float ha, hb;
    avs_unpack_f16x2(h[j], ha, hb);
    // frexp's exponent is E + 1 (0 for hi = 0, where the remainder rounds to zero whatever the step)
    const int ea = max(__builtin_amdgcn_frexp_expf(ha), -5), eb = max(__builtin_amdgcn_frexp_expf(hb), -5);
    // u in 1..255, i.e. |u - 128| <= 127 < half an ulp of hi in steps: fp16(decoded value) == hi always, so a stored
    // value has ONE representation and AVS_F16X2 holds it with the same hi and lo (a remainder within half a step of
    // +-half an ulp - a near-tie of the fp16 rounding - is stored one step short: 2^-18 relative at worst)
    const float qa = fmaxf(rintf(ldexpf(a - ha, 19 - ea)) + 128.f, 1.f);
    const float qb = fmaxf(rintf(ldexpf(b - hb, 19 - eb)) + 128.f, 1.f);
    // (v_cvt_pk_u8_f32 saturates at 255)
    r[j >> 1] = __builtin_amdgcn_cvt_pk_u8_f32(qa, 2 * (j & 1), r[j >> 1]);
    r[j >> 1] = __builtin_amdgcn_cvt_pk_u8_f32(qb, 2 * (j & 1) + 1, r[j >> 1]);
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  rem = make_uint2(r[0], r[1]);
}
// the fp16 lo halves (the matrix cores' second operand) of 8 values from their hi halves and remainder bytes
__device__ __forceinline__ uint4 avs_f16p8_lo8(const uint4& hi, const uint2& rem) {
  return make_uint4(avs_f16p8_lo2(hi.x, rem.x, 0), avs_f16p8_lo2(hi.y, rem.x, 2), avs_f16p8_lo2(hi.z, rem.y, 0),
                    avs_f16p8_lo2(hi.w, rem.y, 2));
}
__device__ __forceinline__ void avs_f16p8_join8(const uint4& hi, const uint2& rem, float (&v)[8]) {
  avs_f16x2_join8(hi, avs_f16p8_lo8(hi, rem), v);
}
// element type tag of the elementwise kernels: ONE slot (4 bytes); only whole runs of 8 are ever loaded or stored
struct avs_h2_tag { unsigned bits; };

template <typename T> struct avs_elem;
template <> struct avs_elem<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
struct avs_bf16_tag { unsigned short bits; };
template <> struct avs_elem<avs_bf16_tag> {
  static __device__ __forceinline__ float load(const avs_bf16_tag* p) { return avs_bf16_to_f32(p->bits); }
  static __device__ __forceinline__ void store(avs_bf16_tag* p, float v) { p->bits = avs_f32_to_bf16(v); }
};

__device__ __forceinline__ float avs_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float avs_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
