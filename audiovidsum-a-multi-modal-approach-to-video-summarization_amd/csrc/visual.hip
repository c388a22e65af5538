// Visual front-end kernels: all HBM-bound, one pass
// over their tensor, 16-byte accesses along the channel axis of NHWC.
#include "avs_internal.h"
#include <math.h>

// ---------------------------------------------------------------------------
// u8 HWC frame -> normalised, zero-padded, 4-channel NHWC image
// ---------------------------------------------------------------------------
struct NormParams {
  float denom;
  float mean[3], stdv[3];
  float aff_a[3], aff_b[3];
  int has_affine;
};

template <typename T>
__global__ __launch_bounds__(256) void frames_normalize_kernel(const uint8_t* __restrict__ src, int n, int h, int w,
                                                               NormParams np, T* __restrict__ out, int out_h,
                                                               int out_w, int pad_t, int pad_l) {
  const long long total = (long long)n * out_h * out_w;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % out_w);
    const long long t = i / out_w;
    const int oy = (int)(t % out_h);
    const long long img = t / out_h;
    const int sy = oy - pad_t, sx = ox - pad_l;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)sy < (unsigned)h && (unsigned)sx < (unsigned)w) {
      const uint8_t* s = src + ((img * h + sy) * (long long)w + sx) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float f = (float)s[c] / np.denom;
        f = (f - np.mean[c]) / np.stdv[c];
        if (np.has_affine) {
          f = f * np.aff_a[c];
          f = f + np.aff_b[c];
        }
        v[c] = f;
      }
    }
    T* o = out + i * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) avs_elem<T>::store(o + c, v[c]);
  }
}

// AVS_F16X2 image: a thread writes two pixels = one run of 8 slots (2 x (3 channels + a zero channel)).  The normalised
// value of a byte is one of 3 x 256: every workgroup evaluates frames_normalize_kernel's expression (two IEEE divisions per
// value) ONCE per (channel, byte) into an LDS table of fp16 hi | lo pairs - avs_f16x2_split8's arithmetic, bit for bit -
// and a pixel is three table reads.
__global__ __launch_bounds__(256) void frames_normalize_h2_kernel(const uint8_t* __restrict__ src, int n, int h, int w,
                                                                  NormParams np, uint4* __restrict__ out, int out_h,
                                                                  int out_w, int pad_t, int pad_l) {
  __shared__ unsigned lut[3][256];   // bits 0-15: the hi half, bits 16-31: the lo half
  for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) {
    const int c = i >> 8;
    float f = (float)(i & 255) / np.denom;
    f = (f - np.mean[c]) / np.stdv[c];
    if (np.has_affine) {
      f = f * np.aff_a[c];
      f = f + np.aff_b[c];
    }
    const float v8[8] = {f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint4 hi, lo;
    avs_f16x2_split8(v8, hi, lo);
    lut[c][i & 255] = (hi.x & 0xffffu) | (lo.x << 16);
  }
  __syncthreads();
  const int pw = out_w >> 1;
  const long long total = (long long)n * out_h * pw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % pw) * 2;
    const long long t = i / pw;
    const int oy = (int)(t % out_h);
    const long long img = t / out_h;
    const int sy = oy - pad_t;
    unsigned e[2][3] = {{0u, 0u, 0u}, {0u, 0u, 0u}};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int sx = ox + q - pad_l;
      if ((unsigned)sy < (unsigned)h && (unsigned)sx < (unsigned)w) {
        const uint8_t* s = src + ((img * h + sy) * (long long)w + sx) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) e[q][c] = lut[c][s[c]];
      }
    }
    // halves in slot order: pixel 0 (c0 c1 | c2 0), pixel 1 (c0 c1 | c2 0)
    out[2 * i] = make_uint4((e[0][0] & 0xffffu) | (e[0][1] << 16), e[0][2] & 0xffffu,
                            (e[1][0] & 0xffffu) | (e[1][1] << 16), e[1][2] & 0xffffu);
    out[2 * i + 1] = make_uint4((e[0][0] >> 16) | (e[0][1] & 0xffff0000u), e[0][2] >> 16,
                                (e[1][0] >> 16) | (e[1][1] & 0xffff0000u), e[1][2] >> 16);
  }
}

// fp32 <-> AVS_F16X2 (weights are converted once per parameter version; tests read activations back)
__global__ __launch_bounds__(256) void f16x2_pack_kernel(const float* __restrict__ src, uint4* __restrict__ dst,
                                                         long long runs) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < runs; i += (long long)gridDim.x * blockDim.x) {
    const float4 a = *reinterpret_cast<const float4*>(src + 8 * i), b = *reinterpret_cast<const float4*>(src + 8 * i + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint4 hi, lo;
    avs_f16x2_split8(v, hi, lo);
    dst[2 * i] = hi;
    dst[2 * i + 1] = lo;
  }
}
__global__ __launch_bounds__(256) void f16x2_unpack_kernel(const uint4* __restrict__ src, float* __restrict__ dst,
                                                           long long runs) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < runs; i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    avs_f16x2_join8(src[2 * i], src[2 * i + 1], v);
    *reinterpret_cast<float4*>(dst + 8 * i) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(dst + 8 * i + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

extern "C" int avs_f16x2_pack_f32(const float* d_src, void* d_dst, int64_t n, avs_stream_t stream) {
  AVS_REQUIRE(n >= 0 && n % 8 == 0, AVS_E_SHAPE, "avs_f16x2_pack_f32: n = %lld must be a multiple of 8", (long long)n);
  if (n == 0) return AVS_OK;
  AVS_REQUIRE(d_src && d_dst, AVS_E_ARG, "avs_f16x2_pack_f32: null pointer");
  AVS_REQUIRE(avs_aligned16(d_src) && (((uintptr_t)d_dst) & 31u) == 0, AVS_E_ALIGN,
              "avs_f16x2_pack_f32: src must be 16-byte, dst 32-byte aligned");
  long long gx = avs_cdiv(n / 8, 256);
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(f16x2_pack_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, d_src, (uint4*)d_dst, n / 8);
  AVS_CHECK_LAUNCH("avs_f16x2_pack_f32");
  return AVS_OK;
}

extern "C" int avs_f16x2_unpack_f32(const void* d_src, float* d_dst, int64_t n, avs_stream_t stream) {
  AVS_REQUIRE(n >= 0 && n % 8 == 0, AVS_E_SHAPE, "avs_f16x2_unpack_f32: n = %lld must be a multiple of 8", (long long)n);
  if (n == 0) return AVS_OK;
  AVS_REQUIRE(d_src && d_dst, AVS_E_ARG, "avs_f16x2_unpack_f32: null pointer");
  AVS_REQUIRE(avs_aligned16(d_dst) && (((uintptr_t)d_src) & 31u) == 0, AVS_E_ALIGN,
              "avs_f16x2_unpack_f32: dst must be 16-byte, src 32-byte aligned");
  long long gx = avs_cdiv(n / 8, 256);
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(f16x2_unpack_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_src, d_dst,
                     n / 8);
  AVS_CHECK_LAUNCH("avs_f16x2_unpack_f32");
  return AVS_OK;
}

// fp32 <-> AVS_F16P8 (fp16 hi + 8-bit remainder: 48 bytes per 16 values; tests and tools)
__global__ __launch_bounds__(256) void f16p8_pack_kernel(const float* __restrict__ src, char* __restrict__ dst, long long runs) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < runs; i += (long long)gridDim.x * blockDim.x) {
    const float4 a = *reinterpret_cast<const float4*>(src + 8 * i), b = *reinterpret_cast<const float4*>(src + 8 * i + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint4 hi;
    uint2 rem;
    avs_f16p8_split8(v, hi, rem);
    char* blk = dst + (i >> 1) * 48;
    *reinterpret_cast<uint4*>(blk + (i & 1) * 16) = hi;
    *reinterpret_cast<uint2*>(blk + 32 + (i & 1) * 8) = rem;
  }
}
__global__ __launch_bounds__(256) void f16p8_unpack_kernel(const char* __restrict__ src, float* __restrict__ dst, long long runs) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < runs; i += (long long)gridDim.x * blockDim.x) {
    const char* blk = src + (i >> 1) * 48;
    float v[8];
    avs_f16p8_join8(*reinterpret_cast<const uint4*>(blk + (i & 1) * 16), *reinterpret_cast<const uint2*>(blk + 32 + (i & 1) * 8), v);
    *reinterpret_cast<float4*>(dst + 8 * i) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(dst + 8 * i + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

extern "C" int avs_f16p8_pack_f32(const float* d_src, void* d_dst, int64_t n, avs_stream_t stream) {
  AVS_REQUIRE(n >= 0 && n % 16 == 0, AVS_E_SHAPE, "avs_f16p8_pack_f32: n = %lld must be a multiple of 16", (long long)n);
  if (n == 0) return AVS_OK;
  AVS_REQUIRE(d_src && d_dst, AVS_E_ARG, "avs_f16p8_pack_f32: null pointer");
  AVS_REQUIRE(avs_aligned16(d_src) && avs_aligned16(d_dst), AVS_E_ALIGN, "avs_f16p8_pack_f32: operands must be 16-byte aligned");
  long long gx = avs_cdiv(n / 8, 256);
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(f16p8_pack_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, d_src, (char*)d_dst, n / 8);
  AVS_CHECK_LAUNCH("avs_f16p8_pack_f32");
  return AVS_OK;
}

extern "C" int avs_f16p8_unpack_f32(const void* d_src, float* d_dst, int64_t n, avs_stream_t stream) {
  AVS_REQUIRE(n >= 0 && n % 16 == 0, AVS_E_SHAPE, "avs_f16p8_unpack_f32: n = %lld must be a multiple of 16", (long long)n);
  if (n == 0) return AVS_OK;
  AVS_REQUIRE(d_src && d_dst, AVS_E_ARG, "avs_f16p8_unpack_f32: null pointer");
  AVS_REQUIRE(avs_aligned16(d_src) && avs_aligned16(d_dst), AVS_E_ALIGN, "avs_f16p8_unpack_f32: operands must be 16-byte aligned");
  long long gx = avs_cdiv(n / 8, 256);
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(f16p8_unpack_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, (const char*)d_src, d_dst,
                     n / 8);
  AVS_CHECK_LAUNCH("avs_f16p8_unpack_f32");
  return AVS_OK;
}

extern "C" int avs_frames_normalize_u8(int dtype, const uint8_t* d_src, int n, int h, int w, float denom,
                                       const float* mean3, const float* std3, const float* affine6, void* d_out,
                                       int out_h, int out_w, int pad_t, int pad_l, avs_stream_t stream) {
  AVS_REQUIRE(dtype == AVS_F32 || dtype == AVS_BF16 || dtype == AVS_F16X2, AVS_E_ARG, "avs_frames_normalize_u8: bad dtype");
  AVS_REQUIRE(dtype != AVS_F16X2 || (out_w % 2 == 0 && (((uintptr_t)d_out) & 31u) == 0), AVS_E_ALIGN,
              "avs_frames_normalize_u8: AVS_F16X2 needs an even output width and a 32-byte aligned output");
  AVS_REQUIRE(n >= 0 && h > 0 && w > 0 && out_h >= h + pad_t && out_w >= w + pad_l && pad_t >= 0 && pad_l >= 0,
              AVS_E_SHAPE, "avs_frames_normalize_u8: bad extents");
  if (n == 0) return AVS_OK;
  AVS_REQUIRE(d_src && d_out && mean3 && std3, AVS_E_ARG, "avs_frames_normalize_u8: null pointer");
  AVS_REQUIRE(denom != 0.f, AVS_E_ARG, "avs_frames_normalize_u8: denom == 0");
  NormParams np;
  np.denom = denom;
  for (int c = 0; c < 3; ++c) {
    np.mean[c] = mean3[c];
    np.stdv[c] = std3[c];
    np.aff_a[c] = affine6 ? affine6[c] : 1.f;
    np.aff_b[c] = affine6 ? affine6[3 + c] : 0.f;
  }
  np.has_affine = affine6 != nullptr;
  const long long total = (long long)n * out_h * out_w;
  const int grid = (int)(avs_cdiv(total, 256) < 16384 ? avs_cdiv(total, 256) : 16384);
  if (dtype == AVS_F16X2)   // (a thread writes two pixels; few enough workgroups that the table build does not matter)
    hipLaunchKernelGGL(frames_normalize_h2_kernel, dim3(grid < 4096 ? grid : 4096), dim3(256), 0, (hipStream_t)stream, d_src, n, h, w, np,
                       (uint4*)d_out, out_h, out_w, pad_t, pad_l);
  else if (dtype == AVS_F32)
    hipLaunchKernelGGL(frames_normalize_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_src, n, h, w,
                       np, (float*)d_out, out_h, out_w, pad_t, pad_l);
  else
    hipLaunchKernelGGL(frames_normalize_kernel<avs_bf16_tag>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_src, n,
                       h, w, np, (avs_bf16_tag*)d_out, out_h, out_w, pad_t, pad_l);
  AVS_CHECK_LAUNCH("avs_frames_normalize_u8");
  return AVS_OK;
}

// ---------------------------------------------------------------------------
// cv2.resize(..., INTER_LINEAR) for uint8, 3 channels: 11-bit fixed-point
// coefficients, horizontal pass in int32, vertical pass with the >>4, >>16, +2, >>2
// rounding of OpenCV's VResizeLinear.  [3P-memory: OpenCV source absent here]
// ---------------------------------------------------------------------------
__device__ __forceinline__ void cv_linear_coef(int d, double scale, int smax, int& s0, int& s1, int& c0, int& c1) {
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) {
    f = 0.f;
    s = 0;
  }
  if (s >= smax - 1) {
    f = 0.f;
    s = smax - 1;
  }
  s0 = s;
  s1 = min(s + 1, smax - 1);
  c0 = __float2int_rn((1.f - f) * 2048.f);
  c1 = __float2int_rn(f * 2048.f);
}

__global__ __launch_bounds__(256) void resize_bilinear_kernel(const uint8_t* __restrict__ src, int n, int sh, int sw,
                                                              uint8_t* __restrict__ dst, int dh, int dw,
                                                              double scale_y, double scale_x) {
  const long long total = (long long)n * dh * dw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int dx = (int)(i % dw);
    const long long t = i / dw;
    const int dy = (int)(t % dh);
    const long long img = t / dh;
    int x0, x1, a0, a1, y0, y1, b0, b1;
    cv_linear_coef(dx, scale_x, sw, x0, x1, a0, a1);
    cv_linear_coef(dy, scale_y, sh, y0, y1, b0, b1);
    const uint8_t* r0 = src + (img * sh + y0) * (long long)sw * 3;
    const uint8_t* r1 = src + (img * sh + y1) * (long long)sw * 3;
    uint8_t* o = dst + i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int S0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
      const int S1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
      int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      o[c] = (uint8_t)v;
    }
  }
}

extern "C" int avs_resize_bilinear_u8(const uint8_t* d_src, int n, int sh, int sw, uint8_t* d_dst, int dh, int dw,
                                      avs_stream_t stream) {
  AVS_REQUIRE(n >= 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0, AVS_E_SHAPE, "avs_resize_bilinear_u8: bad extents");
  if (n == 0) return AVS_OK;
  AVS_REQUIRE(d_src && d_dst, AVS_E_ARG, "avs_resize_bilinear_u8: null pointer");
  const long long total = (long long)n * dh * dw;
  const int grid = (int)(avs_cdiv(total, 256) < 16384 ? avs_cdiv(total, 256) : 16384);
  // OpenCV: inv_scale = dsize/ssize (double); scale = 1/inv_scale.
  const double scale_x = 1.0 / ((double)dw / (double)sw);
  const double scale_y = 1.0 / ((double)dh / (double)sh);
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_src, n, sh, sw, d_dst, dh,
                     dw, scale_y, scale_x);
  AVS_CHECK_LAUNCH("avs_resize_bilinear_u8");
  return AVS_OK;
}

// ---------------------------------------------------------------------------
// Batch-statistics BatchNorm: per (group, channel) mean / biased variance over
// the group's rows, folded with gamma/beta into scale/shift.
// Thread = 4 (8) channels; a block is TC channel-threads x TR row-threads; Welford
// per row-thread + Chan's merge (see the kernel).
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <>
__device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
  const float4 f = *reinterpret_cast<const float4*>(p);
  v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
}
template <>
__device__ __forceinline__ void load4<avs_bf16_tag>(const avs_bf16_tag* p, float (&v)[4]) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(u.x << 16);
  v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16);
  v[3] = __uint_as_float(u.y & 0xffff0000u);
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <>
__device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void store4<avs_bf16_tag>(avs_bf16_tag* p, const float (&v)[4]) {
  uint2 u;
  u.x = (unsigned)avs_f32_to_bf16(v[0]) | ((unsigned)avs_f32_to_bf16(v[1]) << 16);
  u.y = (unsigned)avs_f32_to_bf16(v[2]) | ((unsigned)avs_f32_to_bf16(v[3]) << 16);
  *reinterpret_cast<uint2*>(p) = u;
}

template <typename T, int V>
__device__ __forceinline__ void loadv(const T* p, float (&v)[V]);

template <typename T, int V = 4>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, int c, long long ldx,
                                                       const int64_t* __restrict__ group_rows,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                       int tc) {
  // Welford's update per row-thread (running mean + sum of squares about it: nothing of the E[d^2] - E[d]^2 form, so a
  // constant frame - whose first row, a border pixel, is an outlier - loses nothing), then Chan's merge of the
  // row-threads in a fixed order.  Deterministic.
  __shared__ float red[2][256][V];
  __shared__ int cnt[256];
  const int g = blockIdx.x;
  const int tr = 256 / tc;
  const int ct = threadIdx.x % tc;  // channel-thread
  const int rt = threadIdx.x / tc;  // row-thread
  const int ch = (blockIdx.y * tc + ct) * V;
  const long long r0 = group_rows[g], r1 = group_rows[g + 1];
  const long long nrows = r1 - r0;
  float mean[V], m2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) mean[j] = m2[j] = 0.f;
  const bool active = ch < c && nrows > 0;
  int k = 0;
  if (active) {
    for (long long r = r0 + rt; r < r1; r += tr) {
      float v[V];
      loadv<T, V>(x + r * ldx + ch, v);
      ++k;
      const float inv_k = 1.f / (float)k;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float d = v[j] - mean[j];
        mean[j] = fmaf(d, inv_k, mean[j]);
        m2[j] = fmaf(d, v[j] - mean[j], m2[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) {
    red[0][threadIdx.x][j] = mean[j];
    red[1][threadIdx.x][j] = m2[j];
  }
  cnt[threadIdx.x] = k;
  __syncthreads();
  if (rt == 0 && active) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float n = 0.f, mu = 0.f, s2 = 0.f;
      for (int q = 0; q < tr; ++q) {
        const float nq = (float)cnt[q * tc + ct];
        if (nq == 0.f) continue;
        const float d = red[0][q * tc + ct][j] - mu, nn = n + nq;
        mu = mu + d * (nq / nn);
        s2 = s2 + red[1][q * tc + ct][j] + d * d * (n * nq / nn);
        n = nn;
      }
      const float var = fmaxf(s2 / n, 0.f);
      const float rstd = 1.f / sqrtf(var + eps);
      const float sc = rstd * gamma[ch + j];
      scale[(long long)g * c + ch + j] = sc;
      shift[(long long)g * c + ch + j] = beta[ch + j] - mu * sc;
    }
  }
}

// 16 bytes of channels per thread (4 fp32 / 8 bf16) whenever the channel count allows; the taps of
// neighbouring outputs overlap, so most loads are L2 hits and the kernel is bound by load issue.
template <typename T, int V>
__device__ __forceinline__ void loadv(const T* p, float (&v)[V]) {
  if constexpr (V == 4) {
    load4<T>(p, reinterpret_cast<float(&)[4]>(v));
  } else if constexpr (sizeof(T) == 4) {
    // AVS_F16X2: 8 slots = 16 bytes of hi halves + 16 bytes of lo halves
    static_assert(V == 8, "AVS_F16X2 moves whole runs of 8 slots");
    const uint4* q = reinterpret_cast<const uint4*>(p);
    avs_f16x2_join8(q[0], q[1], v);
  } else {
    static_assert(sizeof(T) == 2 && V == 8, "8-wide path is bf16 / f16x2 only");
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = __uint_as_float(w[j] << 16);
      v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
    }
  }
}
template <typename T, int V>
__device__ __forceinline__ void storev(T* p, const float (&v)[V]) {
  if constexpr (V == 4) {
    store4<T>(p, reinterpret_cast<const float(&)[4]>(v));
  } else if constexpr (sizeof(T) == 4) {
    uint4 hi, lo;
    avs_f16x2_split8(v, hi, lo);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = hi;
    q[1] = lo;
  } else {
    uint4 u;
    unsigned* w = reinterpret_cast<unsigned*>(&u);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      w[j] = (unsigned)avs_f32_to_bf16(v[2 * j]) | ((unsigned)avs_f32_to_bf16(v[2 * j + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = u;
  }
}

extern "C" int avs_bn_batch_stats(int dtype, const void* d_x, int64_t rows, int c, int64_t ldx,
                                  const int64_t* d_group_rows, int groups, const float* d_gamma, const float* d_beta,
                                  float eps, float* d_scale, float* d_shift, avs_stream_t stream) {
  AVS_REQUIRE(dtype == AVS_F32 || dtype == AVS_BF16 || dtype == AVS_F16X2, AVS_E_ARG, "avs_bn_batch_stats: bad dtype");
  const bool h2 = dtype == AVS_F16X2;
  AVS_REQUIRE(!h2 || (c % 8 == 0 && ldx % 8 == 0 && (((uintptr_t)d_x) & 31u) == 0), AVS_E_ALIGN,
              "avs_bn_batch_stats: AVS_F16X2 needs channels / strides in multiples of 8 slots, 32-byte aligned");
  AVS_REQUIRE(rows >= 0 && c > 0 && c % 4 == 0 && ldx >= c && ldx % 4 == 0 && groups >= 0, AVS_E_SHAPE,
              "avs_bn_batch_stats: rows=%lld c=%d ldx=%lld groups=%d", (long long)rows, c, (long long)ldx, groups);
  if (groups == 0) return AVS_OK;
  AVS_REQUIRE(d_x && d_group_rows && d_gamma && d_beta && d_scale && d_shift, AVS_E_ARG,
              "avs_bn_batch_stats: null pointer");
  AVS_REQUIRE(avs_aligned16(d_x), AVS_E_ALIGN, "avs_bn_batch_stats: x not 16-byte aligned");
  const int vw = h2 ? 8 : 4;
  int tc = 1;
  while (tc < 64 && tc * vw < c) tc <<= 1;  // power of two <= 64 channel-threads
  dim3 grid(groups, (unsigned)avs_cdiv(c, tc * vw));
  if (h2)
    hipLaunchKernelGGL((bn_stats_kernel<avs_h2_tag, 8>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const avs_h2_tag*)d_x, c, (long long)ldx, d_group_rows, d_gamma, d_beta, eps, d_scale,
                       d_shift, tc);
  else if (dtype == AVS_F32)
    hipLaunchKernelGGL(bn_stats_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)d_x, c,
                       (long long)ldx, d_group_rows, d_gamma, d_beta, eps, d_scale, d_shift, tc);
  else
    hipLaunchKernelGGL(bn_stats_kernel<avs_bf16_tag>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const avs_bf16_tag*)d_x, c, (long long)ldx, d_group_rows, d_gamma, d_beta, eps, d_scale,
                       d_shift, tc);
  AVS_CHECK_LAUNCH("avs_bn_batch_stats");
  return AVS_OK;
}


// V elements (4, or 8 = 16 bytes of bf16) per thread
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, int c, long long ldx,
                                                       const int64_t* __restrict__ group_rows, long long rows,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const T* __restrict__ res, long long ldr, int act,
                                                       T* __restrict__ y, long long ldy) {
  const int cv = c / V;
  long long r0 = 0, r1 = rows;
  long long g = 0;
  if (group_rows) {
    g = blockIdx.y;
    r0 = group_rows[g];
    r1 = group_rows[g + 1];
  }
  const long long total = (r1 - r0) * cv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = r0 + i / cv;
    const int ch = (int)(i % cv) * V;
    float v[V], sc[V], sf[V];
    loadv<T, V>(x + row * ldx + ch, v);
#pragma unroll
    for (int q = 0; q < V; q += 4) {
      load4<float>(scale + g * c + ch + q, reinterpret_cast<float(&)[4]>(sc[q]));
      load4<float>(shift + g * c + ch + q, reinterpret_cast<float(&)[4]>(sf[q]));
    }
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = v[j] * sc[j] + sf[j];
    if (res) {
      float rv[V];
      loadv<T, V>(res + row * ldr + ch, rv);
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] += rv[j];
    }
    if (act == AVS_ACT_RELU) {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    storev<T, V>(y + row * ldy + ch, v);
  }
}

extern "C" int avs_bn_apply(int dtype, const void* d_x, int64_t rows, int c, int64_t ldx, const int64_t* d_group_rows,
                            int groups, int64_t max_group_rows, const float* d_scale, const float* d_shift,
                            const void* d_residual, int64_t ldr, int act, void* d_y, int64_t ldy,
                            avs_stream_t stream) {
  AVS_REQUIRE(dtype == AVS_F32 || dtype == AVS_BF16 || dtype == AVS_F16X2, AVS_E_ARG, "avs_bn_apply: bad dtype");
  const bool h2 = dtype == AVS_F16X2;
  AVS_REQUIRE(!h2 || (c % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (!d_residual || ldr % 8 == 0) &&
                      (((uintptr_t)d_x | (uintptr_t)d_y | (uintptr_t)d_residual) & 31u) == 0),
              AVS_E_ALIGN, "avs_bn_apply: AVS_F16X2 needs channels / strides in multiples of 8 slots, 32-byte aligned");
  AVS_REQUIRE(rows >= 0 && c > 0 && c % 4 == 0 && ldx >= c && ldx % 4 == 0 && ldy >= c && ldy % 4 == 0 &&
                  (!d_residual || (ldr >= c && ldr % 4 == 0)),
              AVS_E_SHAPE, "avs_bn_apply: rows=%lld c=%d ldx=%lld ldy=%lld ldr=%lld", (long long)rows, c,
              (long long)ldx, (long long)ldy, (long long)ldr);
  AVS_REQUIRE((groups > 0) == (d_group_rows != nullptr), AVS_E_ARG, "avs_bn_apply: groups and d_group_rows disagree");
  if (rows == 0) return AVS_OK;
  AVS_REQUIRE(d_x && d_scale && d_shift && d_y, AVS_E_ARG, "avs_bn_apply: null pointer");
  const bool wide = dtype == AVS_BF16 && c % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (!d_residual || ldr % 8 == 0) &&
                    avs_aligned16(d_x) && avs_aligned16(d_y) && avs_aligned16(d_residual);
  const long long span = (groups > 0 ? max_group_rows : rows) * (c / ((wide || h2) ? 8 : 4));
  AVS_REQUIRE(span > 0, AVS_E_SHAPE, "avs_bn_apply: max_group_rows must be > 0");
  long long gx = avs_cdiv(span, 256);
  if (gx > 8192) gx = 8192;
  dim3 grid((unsigned)gx, groups > 0 ? groups : 1);
  AVS_REQUIRE(grid.y <= 65535, AVS_E_SHAPE, "avs_bn_apply: more than 65535 groups in one call");
  if (h2)
    hipLaunchKernelGGL((bn_apply_kernel<avs_h2_tag, 8>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const avs_h2_tag*)d_x, c, (long long)ldx, d_group_rows, (long long)rows, d_scale, d_shift,
                       (const avs_h2_tag*)d_residual, (long long)ldr, act, (avs_h2_tag*)d_y, (long long)ldy);
  else if (dtype == AVS_F32)
    hipLaunchKernelGGL((bn_apply_kernel<float, 4>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)d_x, c,
                       (long long)ldx, d_group_rows, (long long)rows, d_scale, d_shift, (const float*)d_residual,
                       (long long)ldr, act, (float*)d_y, (long long)ldy);
  else if (wide)
    hipLaunchKernelGGL((bn_apply_kernel<avs_bf16_tag, 8>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const avs_bf16_tag*)d_x, c, (long long)ldx, d_group_rows, (long long)rows, d_scale, d_shift,
                       (const avs_bf16_tag*)d_residual, (long long)ldr, act, (avs_bf16_tag*)d_y, (long long)ldy);
  else
    hipLaunchKernelGGL((bn_apply_kernel<avs_bf16_tag, 4>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const avs_bf16_tag*)d_x, c, (long long)ldx, d_group_rows, (long long)rows, d_scale, d_shift,
                       (const avs_bf16_tag*)d_residual, (long long)ldr, act, (avs_bf16_tag*)d_y, (long long)ldy);
  AVS_CHECK_LAUNCH("avs_bn_apply");
  return AVS_OK;
}

// ---------------------------------------------------------------------------
// Pooling on NHWC
// ---------------------------------------------------------------------------
// K3: the 3x3 window (every pooling of ResNet-50 and Inception-v3) with its nine loads issued together - coordinates
// clamped into the image, taps outside it dropped when the window is combined (the same order as the general loop: the
// same bits) - instead of nine dependent load -> combine rounds behind two `continue` branches.
template <typename T, int V, bool K3 = false>
__global__ __launch_bounds__(256) void pool2d_kernel(int mode, const T* __restrict__ x, int n, int h, int w, int c,
                                                     long long xps, int k, int s, int p, const float* __restrict__ bias,
                                                     int relu, T* __restrict__ y, int ho, int wo, long long yps) {
  const int cv = c / V;
  const long long total = (long long)n * ho * wo * cv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cv) * V;
    long long t = i / cv;
    const int ox = (int)(t % wo);
    t /= wo;
    const int oy = (int)(t % ho);
    const long long img = t / ho;
    float a[V];
#pragma unroll
    for (int j = 0; j < V; ++j) a[j] = mode == 0 ? -INFINITY : 0.f;
    if constexpr (K3) {
      float v[9][V];
      bool ok[9];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = oy * s - p + ky, ix = ox * s - p + kx;
          ok[ky * 3 + kx] = (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
          const int cy = iy < 0 ? 0 : (iy >= h ? h - 1 : iy), cx = ix < 0 ? 0 : (ix >= w ? w - 1 : ix);
          loadv<T, V>(x + ((img * h + cy) * (long long)w + cx) * xps + ch, v[ky * 3 + kx]);
        }
#pragma unroll
      for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int j = 0; j < V; ++j) a[j] = !ok[q] ? a[j] : (mode == 0 ? fmaxf(a[j], v[q][j]) : a[j] + v[q][j]);
    } else {
    for (int ky = 0; ky < k; ++ky) {
      const int iy = oy * s - p + ky;
      if ((unsigned)iy >= (unsigned)h) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int ix = ox * s - p + kx;
        if ((unsigned)ix >= (unsigned)w) continue;
        float v[V];
        loadv<T, V>(x + ((img * h + iy) * (long long)w + ix) * xps + ch, v);
#pragma unroll
        for (int j = 0; j < V; ++j) a[j] = mode == 0 ? fmaxf(a[j], v[j]) : a[j] + v[j];
      }
    }
    }
    if (mode == 1) {
      const float d = (float)(k * k);
#pragma unroll
      for (int j = 0; j < V; ++j) a[j] = a[j] / d;
    }
    if (bias != nullptr) {
#pragma unroll
      for (int j = 0; j < V; ++j) a[j] += bias[ch + j];
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < V; ++j) a[j] = fmaxf(a[j], 0.f);
    }
    storev<T, V>(y + ((img * ho + oy) * (long long)wo + ox) * yps + ch, a);
  }
}

extern "C" int avs_pool2d_nhwc(int dtype, int mode, const void* d_x, int n, int h, int w, int c, int64_t x_px_stride,
                               int k, int s, int p, const float* d_bias, int act, void* d_y, int ho, int wo,
                               int64_t y_px_stride, avs_stream_t stream) {
  AVS_REQUIRE(dtype == AVS_F32 || dtype == AVS_BF16 || dtype == AVS_F16X2, AVS_E_ARG, "avs_pool2d_nhwc: bad dtype");
  const bool h2 = dtype == AVS_F16X2;
  AVS_REQUIRE(!h2 || (c % 8 == 0 && x_px_stride % 8 == 0 && y_px_stride % 8 == 0 &&
                      (((uintptr_t)d_x | (uintptr_t)d_y) & 31u) == 0),
              AVS_E_ALIGN, "avs_pool2d_nhwc: AVS_F16X2 needs channels / strides in multiples of 8 slots, 32-byte aligned");
  AVS_REQUIRE(act == AVS_ACT_NONE || act == AVS_ACT_RELU, AVS_E_ARG, "avs_pool2d_nhwc: bad activation %d", act);
  const int relu = act == AVS_ACT_RELU;
  AVS_REQUIRE(mode == 0 || mode == 1, AVS_E_ARG, "avs_pool2d_nhwc: bad mode");
  AVS_REQUIRE(n >= 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && k > 0 && s > 0 && p >= 0 && p < k && ho > 0 &&
                  wo > 0 && x_px_stride >= c && x_px_stride % 4 == 0 && y_px_stride >= c && y_px_stride % 4 == 0,
              AVS_E_SHAPE, "avs_pool2d_nhwc: bad extents");
  AVS_REQUIRE((ho - 1) * s - p < h && (wo - 1) * s - p < w, AVS_E_SHAPE, "avs_pool2d_nhwc: output extent too large");
  if (n == 0) return AVS_OK;
  AVS_REQUIRE(d_x && d_y, AVS_E_ARG, "avs_pool2d_nhwc: null pointer");
  const bool wide = dtype == AVS_BF16 && c % 8 == 0 && x_px_stride % 8 == 0 && y_px_stride % 8 == 0 &&
                    avs_aligned16(d_x) && avs_aligned16(d_y);
  const long long total = (long long)n * ho * wo * (c / ((wide || h2) ? 8 : 4));
  long long gx = avs_cdiv(total, 256);
  if (gx > 65536) gx = 65536;
  if (h2 && k == 3)
    hipLaunchKernelGGL((pool2d_kernel<avs_h2_tag, 8, true>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, mode,
                       (const avs_h2_tag*)d_x, n, h, w, c, (long long)x_px_stride, k, s, p, d_bias, relu,
                       (avs_h2_tag*)d_y, ho, wo, (long long)y_px_stride);
  else if (h2)
    hipLaunchKernelGGL((pool2d_kernel<avs_h2_tag, 8>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, mode,
                       (const avs_h2_tag*)d_x, n, h, w, c, (long long)x_px_stride, k, s, p, d_bias, relu,
                       (avs_h2_tag*)d_y, ho, wo, (long long)y_px_stride);
  else if (dtype == AVS_F32)
    hipLaunchKernelGGL((pool2d_kernel<float, 4>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, mode,
                       (const float*)d_x, n, h, w, c, (long long)x_px_stride, k, s, p, d_bias, relu, (float*)d_y, ho,
                       wo, (long long)y_px_stride);
  else if (wide && k == 3)
    hipLaunchKernelGGL((pool2d_kernel<avs_bf16_tag, 8, true>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, mode,
                       (const avs_bf16_tag*)d_x, n, h, w, c, (long long)x_px_stride, k, s, p, d_bias, relu,
                       (avs_bf16_tag*)d_y, ho, wo, (long long)y_px_stride);
  else if (wide)
    hipLaunchKernelGGL((pool2d_kernel<avs_bf16_tag, 8>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, mode,
                       (const avs_bf16_tag*)d_x, n, h, w, c, (long long)x_px_stride, k, s, p, d_bias, relu,
                       (avs_bf16_tag*)d_y, ho, wo, (long long)y_px_stride);
  else
    hipLaunchKernelGGL((pool2d_kernel<avs_bf16_tag, 4>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, mode,
                       (const avs_bf16_tag*)d_x, n, h, w, c, (long long)x_px_stride, k, s, p, d_bias, relu,
                       (avs_bf16_tag*)d_y, ho, wo, (long long)y_px_stride);
  AVS_CHECK_LAUNCH("avs_pool2d_nhwc");
  return AVS_OK;
}

// BatchNorm apply + ReLU + max pooling in one pass (the ResNet stem: bn1 -> relu -> maxpool 3x3/2,
// features/extractors.py:29): reads the RAW convolution once and writes only the pooled map, instead of
// avs_bn_apply (read + write of the full-resolution map) followed by avs_pool2d_nhwc (another read).
// max(relu(.)) == relu(max(.)) and rounding to the activation dtype is monotone, so the result is bit-identical
// to the two-kernel sequence.  Group of an image = the range of d_group_rows (input rows) its first row is in.
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_maxpool_kernel(const T* __restrict__ x, int n, int h, int w, int c,
                                                         long long xps, const int64_t* __restrict__ group_rows,
                                                         int groups, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int relu, int k, int s,
                                                         int p, T* __restrict__ y, int ho, int wo, long long yps) {
  const int cv = c / V;
  const long long total = (long long)n * ho * wo * cv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cv) * V;
    long long t = i / cv;
    const int ox = (int)(t % wo);
    t /= wo;
    const int oy = (int)(t % ho);
    const long long img = t / ho;
    long long g = 0;
    if (group_rows) {  // largest g with group_rows[g] <= first row of the image
      const long long row = img * h * (long long)w;
      int lo = 0, hi = groups - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (group_rows[mid] <= row)
          lo = mid;
        else
          hi = mid - 1;
      }
      g = lo;
    }
    float sc[V], sf[V], a[V];
    if constexpr (V == 8) {
      load4<float>(scale + g * c + ch, reinterpret_cast<float(&)[4]>(sc[0]));
      load4<float>(scale + g * c + ch + 4, reinterpret_cast<float(&)[4]>(sc[4]));
      load4<float>(shift + g * c + ch, reinterpret_cast<float(&)[4]>(sf[0]));
      load4<float>(shift + g * c + ch + 4, reinterpret_cast<float(&)[4]>(sf[4]));
    } else {
      load4<float>(scale + g * c + ch, reinterpret_cast<float(&)[4]>(sc[0]));
      load4<float>(shift + g * c + ch, reinterpret_cast<float(&)[4]>(sf[0]));
    }
#pragma unroll
    for (int j = 0; j < V; ++j) a[j] = -INFINITY;
    for (int ky = 0; ky < k; ++ky) {
      const int iy = oy * s - p + ky;
      if ((unsigned)iy >= (unsigned)h) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int ix = ox * s - p + kx;
        if ((unsigned)ix >= (unsigned)w) continue;
        float v[V];
        loadv<T, V>(x + ((img * h + iy) * (long long)w + ix) * xps + ch, v);
#pragma unroll
        for (int j = 0; j < V; ++j) a[j] = fmaxf(a[j], v[j] * sc[j] + sf[j]);  // avs_bn_apply's expression
      }
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < V; ++j) a[j] = fmaxf(a[j], 0.f);
    }
    storev<T, V>(y + ((img * ho + oy) * (long long)wo + ox) * yps + ch, a);
  }
}

extern "C" int avs_bn_maxpool_nhwc(int dtype, const void* d_x, int n, int h, int w, int c, int64_t x_px_stride,
                                   const int64_t* d_group_rows, int groups, const float* d_scale,
                                   const float* d_shift, int relu, int k, int s, int p, void* d_y, int ho, int wo,
                                   int64_t y_px_stride, avs_stream_t stream) {
  AVS_REQUIRE(dtype == AVS_F32 || dtype == AVS_BF16 || dtype == AVS_F16X2, AVS_E_ARG, "avs_bn_maxpool_nhwc: bad dtype");
  const bool h2 = dtype == AVS_F16X2;
  AVS_REQUIRE(!h2 || (c % 8 == 0 && x_px_stride % 8 == 0 && y_px_stride % 8 == 0 &&
                      (((uintptr_t)d_x | (uintptr_t)d_y) & 31u) == 0),
              AVS_E_ALIGN, "avs_bn_maxpool_nhwc: AVS_F16X2 needs channels / strides in multiples of 8 slots, 32-byte aligned");
  AVS_REQUIRE(n >= 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && k > 0 && s > 0 && p >= 0 && p < k && ho > 0 &&
                  wo > 0 && x_px_stride >= c && x_px_stride % 4 == 0 && y_px_stride >= c && y_px_stride % 4 == 0,
              AVS_E_SHAPE, "avs_bn_maxpool_nhwc: bad extents");
  AVS_REQUIRE((ho - 1) * s - p < h && (wo - 1) * s - p < w, AVS_E_SHAPE, "avs_bn_maxpool_nhwc: output extent too large");
  AVS_REQUIRE((groups > 0) == (d_group_rows != nullptr), AVS_E_ARG,
              "avs_bn_maxpool_nhwc: groups and d_group_rows disagree");
  if (n == 0) return AVS_OK;
  AVS_REQUIRE(d_x && d_y && d_scale && d_shift, AVS_E_ARG, "avs_bn_maxpool_nhwc: null pointer");
  AVS_REQUIRE(avs_aligned16(d_scale) && avs_aligned16(d_shift), AVS_E_ALIGN,
              "avs_bn_maxpool_nhwc: scale / shift must be 16-byte aligned");
  const bool wide = dtype == AVS_BF16 && c % 8 == 0 && x_px_stride % 8 == 0 && y_px_stride % 8 == 0 &&
                    avs_aligned16(d_x) && avs_aligned16(d_y);
  const long long total = (long long)n * ho * wo * (c / ((wide || h2) ? 8 : 4));
  long long gx = avs_cdiv(total, 256);
  if (gx > 65536) gx = 65536;
  if (h2)
    hipLaunchKernelGGL((bn_maxpool_kernel<avs_h2_tag, 8>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream,
                       (const avs_h2_tag*)d_x, n, h, w, c, (long long)x_px_stride, d_group_rows, groups, d_scale,
                       d_shift, relu, k, s, p, (avs_h2_tag*)d_y, ho, wo, (long long)y_px_stride);
  else if (dtype == AVS_F32)
    hipLaunchKernelGGL((bn_maxpool_kernel<float, 4>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream,
                       (const float*)d_x, n, h, w, c, (long long)x_px_stride, d_group_rows, groups, d_scale, d_shift,
                       relu, k, s, p, (float*)d_y, ho, wo, (long long)y_px_stride);
  else if (wide)
    hipLaunchKernelGGL((bn_maxpool_kernel<avs_bf16_tag, 8>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream,
                       (const avs_bf16_tag*)d_x, n, h, w, c, (long long)x_px_stride, d_group_rows, groups, d_scale,
                       d_shift, relu, k, s, p, (avs_bf16_tag*)d_y, ho, wo, (long long)y_px_stride);
  else
    hipLaunchKernelGGL((bn_maxpool_kernel<avs_bf16_tag, 4>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream,
                       (const avs_bf16_tag*)d_x, n, h, w, c, (long long)x_px_stride, d_group_rows, groups, d_scale,
                       d_shift, relu, k, s, p, (avs_bf16_tag*)d_y, ho, wo, (long long)y_px_stride);
  AVS_CHECK_LAUNCH("avs_bn_maxpool_nhwc");
  return AVS_OK;
}

template <typename T, int V = 4>
__global__ __launch_bounds__(256) void global_avgpool_kernel(const T* __restrict__ x, int n, int hw, int c,
                                                             float* __restrict__ y, long long ldy) {
  const int cv = c / V;
  const long long total = (long long)n * cv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cv) * V;
    const long long img = i / cv;
    float a[V];
#pragma unroll
    for (int j = 0; j < V; ++j) a[j] = 0.f;
    const T* base = x + img * hw * (long long)c + ch;
    for (int r = 0; r < hw; ++r) {
      float v[V];
      loadv<T, V>(base + (long long)r * c, v);
#pragma unroll
      for (int j = 0; j < V; ++j) a[j] += v[j];
    }
    const float d = (float)hw;
#pragma unroll
    for (int j = 0; j < V; ++j) a[j] = a[j] / d;
#pragma unroll
    for (int q = 0; q < V; q += 4) store4<float>(y + img * ldy + ch + q, reinterpret_cast<const float(&)[4]>(a[q]));
  }
}

extern "C" int avs_global_avgpool_nhwc(int dtype, const void* d_x, int n, int hw, int c, float* d_y, int64_t ldy,
                                       avs_stream_t stream) {
  AVS_REQUIRE(dtype == AVS_F32 || dtype == AVS_BF16 || dtype == AVS_F16X2, AVS_E_ARG, "avs_global_avgpool_nhwc: bad dtype");
  const bool h2 = dtype == AVS_F16X2;
  AVS_REQUIRE(!h2 || (c % 8 == 0 && (((uintptr_t)d_x) & 31u) == 0), AVS_E_ALIGN,
              "avs_global_avgpool_nhwc: AVS_F16X2 needs channels in multiples of 8 slots, 32-byte aligned");
  AVS_REQUIRE(n >= 0 && hw > 0 && c > 0 && c % 4 == 0 && ldy >= c && ldy % 4 == 0, AVS_E_SHAPE,
              "avs_global_avgpool_nhwc: bad extents");
  if (n == 0) return AVS_OK;
  AVS_REQUIRE(d_x && d_y, AVS_E_ARG, "avs_global_avgpool_nhwc: null pointer");
  AVS_REQUIRE(avs_aligned16(d_y), AVS_E_ALIGN, "avs_global_avgpool_nhwc: y not 16-byte aligned");
  const long long total = (long long)n * (c / (h2 ? 8 : 4));
  long long gx = avs_cdiv(total, 256);
  if (gx > 16384) gx = 16384;
  if (h2)
    hipLaunchKernelGGL((global_avgpool_kernel<avs_h2_tag, 8>), dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream,
                       (const avs_h2_tag*)d_x, n, hw, c, d_y, (long long)ldy);
  else if (dtype == AVS_F32)
    hipLaunchKernelGGL(global_avgpool_kernel<float>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream,
                       (const float*)d_x, n, hw, c, d_y, (long long)ldy);
  else
    hipLaunchKernelGGL(global_avgpool_kernel<avs_bf16_tag>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream,
                       (const avs_bf16_tag*)d_x, n, hw, c, d_y, (long long)ldy);
  AVS_CHECK_LAUNCH("avs_global_avgpool_nhwc");
  return AVS_OK;
}

// out[s,:] = mean over rows seg[s]..seg[s+1] in row order (numpy mean(axis=0))
__global__ __launch_bounds__(256) void segment_mean_kernel(const float* __restrict__ x, long long ldx, int d,
                                                           const int64_t* __restrict__ seg, int nseg,
                                                           float* __restrict__ out, long long ldo) {
  const long long total = (long long)nseg * d;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % d);
    const long long s = i / d;
    const long long r0 = seg[s], r1 = seg[s + 1];
    float a = 0.f;
    for (long long r = r0; r < r1; ++r) a += x[r * ldx + ch];
    out[s * ldo + ch] = r1 > r0 ? a / (float)(r1 - r0) : 0.f;
  }
}

extern "C" int avs_segment_mean_f32(const float* d_x, int64_t ldx, int d, const int64_t* d_seg, int nseg, float* d_out,
                                    int64_t ldo, avs_stream_t stream) {
  AVS_REQUIRE(d > 0 && ldx >= d && ldo >= d && nseg >= 0, AVS_E_SHAPE, "avs_segment_mean_f32: bad extents");
  if (nseg == 0) return AVS_OK;
  AVS_REQUIRE(d_x && d_seg && d_out, AVS_E_ARG, "avs_segment_mean_f32: null pointer");
  long long gx = avs_cdiv((long long)nseg * d, 256);
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(segment_mean_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, d_x, (long long)ldx,
                     d, d_seg, nseg, d_out, (long long)ldo);
  AVS_CHECK_LAUNCH("avs_segment_mean_f32");
  return AVS_OK;
}


// ---------------------------------------------------------------------------
// Shot-boundary scan (features/extractors.py:388-393 calls PySceneDetect's ContentDetector):
// per frame, the sum over pixels of |H - H_prev|, |S - S_prev|, |V - V_prev| with OpenCV's 8-bit BGR->HSV
// (H in [0,180), 12-bit fixed-point division tables) [3P-memory: cv2 / scenedetect sources absent here].
// Integer arithmetic end to end (block reduction by wave shuffle, one integer atomic per block), so the sums are
// exact and order-independent.  `step` = PySceneDetect's downscale stride (frame[::step, ::step]).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int cv_round_div(int num_shifted, double den) { return (int)rint((double)num_shifted / den); }

__device__ __forceinline__ void bgr2hsv_u8(int b, int g, int r, int& h, int& s, int& v) {
  v = max(b, max(g, r));
  const int vmin = min(b, min(g, r));
  const int diff = v - vmin;
  const int sdiv = v > 0 ? cv_round_div(255 << 12, 1.0 * v) : 0;
  const int hdiv = diff > 0 ? cv_round_div(180 << 12, 6.0 * diff) : 0;
  s = (diff * sdiv + (1 << 11)) >> 12;
  int hh;
  if (v == r)
    hh = g - b;
  else if (v == g)
    hh = b - r + 2 * diff;
  else
    hh = r - g + 4 * diff;
  hh = (hh * hdiv + (1 << 11)) >> 12;
  if (hh < 0) hh += 180;
  h = hh;
}

__global__ __launch_bounds__(256) void hsv_frame_diff_kernel(const uint8_t* __restrict__ frames, int h, int w,
                                                             int step, int ph, int pw, unsigned* __restrict__ sums) {
  __shared__ unsigned red[4][3];
  const long long f = blockIdx.y + 1;  // frame f against frame f-1
  const uint8_t* cur = frames + f * h * (long long)w * 3;
  const uint8_t* prv = cur - (long long)h * w * 3;
  unsigned a0 = 0, a1 = 0, a2 = 0;
  const int npx = ph * pw;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    const int y = (i / pw) * step, x = (i % pw) * step;
    const long long o = ((long long)y * w + x) * 3;
    int h1, s1, v1, h0, s0, v0;
    bgr2hsv_u8(cur[o], cur[o + 1], cur[o + 2], h1, s1, v1);
    bgr2hsv_u8(prv[o], prv[o + 1], prv[o + 2], h0, s0, v0);
    a0 += (unsigned)abs(h1 - h0);
    a1 += (unsigned)abs(s1 - s0);
    a2 += (unsigned)abs(v1 - v0);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a0 += __shfl_xor(a0, o, 64);
    a1 += __shfl_xor(a1, o, 64);
    a2 += __shfl_xor(a2, o, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave][0] = a0;
    red[wave][1] = a1;
    red[wave][2] = a2;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    atomicAdd(sums + f * 3 + threadIdx.x,
              red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

extern "C" int avs_hsv_frame_diff_u8(const uint8_t* d_frames, int n, int h, int w, int step, uint32_t* d_sums,
                                     avs_stream_t stream) {
  AVS_REQUIRE(n >= 0 && h > 0 && w > 0 && step > 0, AVS_E_SHAPE, "avs_hsv_frame_diff_u8: bad extents");
  if (n == 0) return AVS_OK;
  AVS_REQUIRE(d_frames && d_sums, AVS_E_ARG, "avs_hsv_frame_diff_u8: null pointer");
  const int ph = (h + step - 1) / step, pw = (w + step - 1) / step;  // len(range(0, h, step))
  AVS_REQUIRE((long long)ph * pw * 255 < (1ll << 32), AVS_E_SHAPE, "avs_hsv_frame_diff_u8: frame too large for u32 sums");
  hipError_t e = hipMemsetAsync(d_sums, 0, sizeof(uint32_t) * 3 * (size_t)n, (hipStream_t)stream);
  if (e != hipSuccess) {
    avs_set_error("avs_hsv_frame_diff_u8: memset failed: %s", hipGetErrorString(e));
    return AVS_E_HIP;
  }
  if (n == 1) return AVS_OK;
  AVS_REQUIRE(n - 1 <= 65535, AVS_E_SHAPE, "avs_hsv_frame_diff_u8: at most 65536 frames per call");
  int bx = (int)avs_cdiv((long long)ph * pw, 256 * 8);
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(hsv_frame_diff_kernel, dim3(bx, n - 1), dim3(256), 0, (hipStream_t)stream, d_frames, h, w, step, ph,
                     pw, d_sums);
  AVS_CHECK_LAUNCH("avs_hsv_frame_diff_u8");
  return AVS_OK;
}

// ---------------------------------------------------------------------------
// Upload of frames by a PULL kernel: a few workgroups read pinned (device-mapped) host memory over PCIe with 16-byte
// loads and store to HBM.  Unlike a copy-engine / blit transfer its footprint on the chip is the caller's choice
// (`workgroups` CUs' worth of one 256-thread block each, 4 loads of 16 bytes in flight per lane), so the upload of the
// next pass can run beside the current pass's kernels at a known, small cost (pipeline.py: the PCIe-inclusive path).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pull_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long n16,
                                                        const uint8_t* __restrict__ tail_src, uint8_t* __restrict__ tail_dst,
                                                        int tail) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a;
    dst[i + stride] = b;
    dst[i + 2 * stride] = c;
    dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) tail_dst[threadIdx.x] = tail_src[threadIdx.x];
}

extern "C" int avs_pull_copy_u8(const uint8_t* h_src_mapped, uint8_t* d_dst, int64_t bytes, int workgroups,
                                avs_stream_t stream) {
  AVS_REQUIRE(bytes >= 0 && workgroups > 0 && workgroups <= 1024, AVS_E_SHAPE, "avs_pull_copy_u8: bytes=%lld workgroups=%d",
              (long long)bytes, workgroups);
  if (bytes == 0) return AVS_OK;
  AVS_REQUIRE(h_src_mapped && d_dst, AVS_E_ARG, "avs_pull_copy_u8: null pointer");
  AVS_REQUIRE(avs_aligned16(h_src_mapped) && avs_aligned16(d_dst), AVS_E_ALIGN, "avs_pull_copy_u8: 16-byte aligned buffers");
  const long long n16 = bytes / 16;
  hipLaunchKernelGGL(pull_copy_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint4*>(h_src_mapped), reinterpret_cast<uint4*>(d_dst), n16, h_src_mapped + n16 * 16,
                     d_dst + n16 * 16, (int)(bytes - n16 * 16));
  AVS_CHECK_LAUNCH("avs_pull_copy_u8");
  return AVS_OK;
}
