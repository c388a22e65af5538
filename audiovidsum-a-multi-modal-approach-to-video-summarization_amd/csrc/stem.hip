// The ResNet-50 stem of the throughput path in two launches: uint8 frames -> conv1 7x7/2 -> bn1 (batch statistics per
// micro-batch group) -> ReLU -> maxpool 3x3/2 (features/extractors.py:29,65,126-140: children()[0:4] of the trunk on
// the (x - mean)/std input).
//
// Unfused this stage is four kernels and 4.6 MB of HBM traffic per frame: normalise (150 KB in, 427 KB out), convolution
// (427 KB in, 1.6 MB raw out), statistics fold, normalise + pool (1.6 MB in, 401 KB out) - the raw 112x112x64 map goes
// out and comes back only because the pooling has to wait for the group's statistics.  It does not have to: with
// y' = relu(scale * y + shift), scale = gamma / sigma, the map y -> y' is monotone (non-decreasing for scale >= 0,
// non-increasing for scale < 0; multiply, add, ReLU and the final rounding are all monotone in floating point), so
//     maxpool(y') = relu(scale * (scale >= 0 ? MAXpool(y) : MINpool(y)) + shift)        exactly, bit for bit.
// And which of the two it is does not wait for the statistics either: sigma > 0, so sign(scale) = sign(gamma), a
// parameter.  stem_fused_kernel therefore pools the RAW (bf16-rounded) convolution output on chip - max for the channels
// with gamma >= 0, min for the others - and writes ONE 56x56x64 map (401 KB per frame) plus the tile's partial sums
// for the statistics; the affine + ReLU is applied afterwards in place (stem_finish_kernel) or by the consumers while
// they stage it (apply = 0: conv1x1_bn_kernel<XF> / bn_gram_affine_kernel<XF>).  0.55 - 1.35 MB of traffic per frame,
// and the normalisation of the input rides in the staging (the padded 4-channel image never exists in HBM).
//
// One workgroup (5 waves) = one frame x 32 of the 64 output channels, walking the frame's 49 tiles; a tile = 8 x 8 pooled outputs = 17 x 17 convolution
// outputs (16 x 16 owned + the halo row / column the pooling windows reach into, recomputed: 13 % extra matrix work).
//   * input patch 39 x 39 pixels: uint8 loads, (x / denom - mean) / std in fp32 (frames_normalize_kernel's arithmetic),
//     bf16, 4 channels (the 4th zero) -> LDS [39][40] pixels of 8 bytes; zero outside the image (the conv's padding);
//   * implicit GEMM from the LDS image, M = 320 rows (289 used) x N = 64 x K = 7 kernel rows x 8 pixels x 4 channels
//     (224, weights zero in the padding: the layout of the unfused stem): the A fragment of (kernel row, half) for an
//     output pixel is the 16 bytes of two neighbouring patch pixels - one aligned ds_read_b128, no im2col copy;
//     weights [64][224] resident in LDS for the whole row of tiles; v_mfma_f32_32x32x16_bf16;
//   * epilogue: column sums of the OWNED outputs (fp32 accumulators) -> per-tile partials (fixed order, no atomics);
//     bf16 raw tile -> LDS; 3x3/2 max (min where gamma < 0) over it -> 16-byte stores.
#include "avs_internal.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int IMG = 224, CONV = 112, POOL = 56, COUT = 64;
constexpr int TILES = 7;                 // tiles per side: 7 x 8 pooled = 56
constexpr int TW = 17;                   // convolution outputs per tile side (16 owned + 1 halo)
constexpr int MROWS = TW * TW;           // 289 used rows of the 320-row GEMM tile
constexpr int PH = 39, PW = 40;          // patch rows / pixels (pixel = 4 x bf16 = 8 bytes)
constexpr int PATCH_BYTES = PH * PW * 8;
constexpr int KDIM = 224;                // 7 kernel rows x 8 pixels x 4 channels
constexpr int CW = 32;                   // output channels per workgroup: two workgroups share a row of tiles, so that four
                                         // of them (20 waves) fit a CU's LDS - with all 64 channels it is two (measured:
                                         // 54 % of the wave-cycles waiting at the five barriers of a tile)
constexpr int W_PITCH = KDIM * 2 + 16;   // 464 bytes: 16 consecutive rows hit 16 different 16-byte slots
constexpr int W_BYTES = CW * W_PITCH;
constexpr int RT_PITCH = CW * 2 + 16;    // raw tile row: 32 channels + 16 bytes of padding
constexpr int RT_BYTES = MROWS * RT_PITCH;
constexpr int MAIN_BYTES = RT_BYTES > PATCH_BYTES ? RT_BYTES : PATCH_BYTES;
constexpr int WAVES = 5;
}  // namespace

struct StemParams {
  const uint8_t* frames;
  const char* w;
  char* osel;          // [n,56,56,64] bf16: per channel the window's max (gamma >= 0) or min (gamma < 0) of the raw output
  const float* gamma;
  float* part;
  long long ldw;
  float denom, mean[3], stdv[3];
#ifdef AVS_STUDY
  int debug;   // study build only (AVS_STEM_DEBUG): 1 no pooling, 2 no raw tile, 4 no staging, 8 no matrix work, 16 no statistics
#endif
};
#ifdef AVS_STUDY
#define STEM_DBG(p, bit) ((p).debug & (bit))
#else
#define STEM_DBG(p, bit) 0
#endif

__global__ __launch_bounds__(WAVES * 64, 4) void stem_fused_kernel(StemParams p) {
  __shared__ __attribute__((aligned(16))) char wimg[W_BYTES];
  __shared__ __attribute__((aligned(16))) char mainb[MAIN_BYTES];
  __shared__ float sred[WAVES][2][CW];
  __shared__ unsigned short lut[3][256];   // bf16((v / denom - mean_c) / std_c) for every byte value: no division per pixel
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const int c0 = (blockIdx.x & 1) * CW;            // this workgroup's channels c0 .. c0 + 31
  const long long img = blockIdx.x >> 1;           // ... of one frame: all 49 tiles (weights, LUT and masks set up once)
  const uint8_t* __restrict__ src = p.frames + img * (long long)(IMG * IMG * 3);

  // weights -> LDS once per workgroup: 32 rows x 28 chunks of 16 bytes
  for (int i = t; i < CW * (KDIM / 8); i += WAVES * 64) {
    const int n = i / (KDIM / 8), ch = i - n * (KDIM / 8);
    *reinterpret_cast<uint4*>(wimg + n * W_PITCH + ch * 16) =
        *reinterpret_cast<const uint4*>(p.w + ((long long)(c0 + n) * p.ldw + ch * 8) * 2);
  }
  for (int i = t; i < 3 * 256; i += WAVES * 64) {
    const int c = i >> 8, v = i & 255;
    const float f = (float)v / p.denom;            // frames_normalize_kernel's expression, evaluated once per value
    lut[c][v] = avs_f32_to_bf16((f - p.mean[c]) / p.stdv[c]);
  }
  // per-lane constants: A base offsets of this lane's two 32-row blocks, ownership masks of its 2 x 16 accumulator rows
  int abase[2];
  unsigned own[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int m = wave * 64 + mt * 32 + lr;
    if (m >= MROWS) m = MROWS - 1;   // rows 289 .. 319 compute garbage from a valid address and are never stored
    const int ly = m / TW, lx = m - ly * TW;
    abase[mt] = ((2 * ly) * PW + 2 * lx) * 8 + lh * 16;
    unsigned mk = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int me = wave * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      const int ey = me / TW, ex = me - ey * TW;
      if (me < MROWS && ey >= 1 && ex >= 1) mk |= 1u << e;   // the halo row / column belongs to the neighbouring tile
    }
    own[mt] = mk;
  }
  const int bbase = lr * W_PITCH + lh * 16;
  // pooling: a thread's items all have the same 8 channels (320 % 4 == 0).  On the order-preserving keys negation is
  // the complement, so the channels with gamma < 0 run the same packed MAX on complemented keys (= min of the values).
  s16x2 flip[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = c0 + (t & 3) * 8 + 2 * j;
    flip[j] = (s16x2){(short)(p.gamma[ch] < 0.f ? -1 : 0), (short)(p.gamma[ch + 1] < 0.f ? -1 : 0)};
  }

  // The patch pixels of a tile are fetched into REGISTERS one tile ahead (15 byte loads per thread, issued before the
  // matrix work of the previous tile and landing during it and its epilogue), then normalised through the LUT and
  // written to LDS: one exposed memory latency per workgroup instead of five per tile.
  constexpr int PPT = (PH * PW + WAVES * 64 - 1) / (WAVES * 64);   // patch pixels per thread (5)
  // (the bytes stay in separate registers until they are used: packing them at load time would make the loads' wait
  //  part of the prefetch and nothing would be in flight during the matrix work)
  unsigned pb0[PPT], pb1[PPT], pb2[PPT];
  unsigned inside = 0u;   // bit j: pixel j of this thread lies inside the image
  int ppy[PPT], ppx[PPT], poff[PPT];   // this thread's patch pixels: row, column, byte offset from the patch origin
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int i = t + j * (WAVES * 64);
    ppy[j] = i < PH * PW ? i / PW : (1 << 20);    // beyond the patch: fails every range test below
    ppx[j] = i - (i / PW) * PW;
    poff[j] = (ppy[j] * IMG + ppx[j]) * 3;
  }
  auto gload = [&](int tile) {
    const int ty = tile / TILES, tx = tile - ty * TILES;
    const int iy0 = 32 * ty - 5, ix0 = 32 * tx - 5;   // image rows 32 ty - 5 .., columns 32 tx - 5 ..
    // rows / columns of the patch that lie inside the image (tile-uniform); column 39 is layout padding
    const int ylo = iy0 < 0 ? -iy0 : 0, yhi = IMG - iy0 < PH ? IMG - iy0 : PH;
    const int xlo = ix0 < 0 ? -ix0 : 0, xhi = IMG - ix0 < PH ? IMG - ix0 : PH;
    const uint8_t* org = src + (iy0 * IMG + ix0) * 3;
    inside = 0u;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      if (ppy[j] >= ylo && ppy[j] < yhi && ppx[j] >= xlo && ppx[j] < xhi) {
        const uint8_t* s = org + poff[j];
        pb0[j] = s[0];
        pb1[j] = s[1];
        pb2[j] = s[2];
        inside |= 1u << j;
      }
    }
  };
  auto pstore = [&]() {
    if (STEM_DBG(p, 4)) return;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int i = t + j * (WAVES * 64);
      if (i >= PH * PW) break;
      unsigned lo = 0u, hi = 0u;
      if ((inside >> j) & 1u) {
        lo = (unsigned)lut[0][pb0[j]] | ((unsigned)lut[1][pb1[j]] << 16);
        hi = (unsigned)lut[2][pb2[j]];
      }
      *reinterpret_cast<uint2*>(mainb + i * 8) = make_uint2(lo, hi);
    }
  };
  gload(0);
  __syncthreads();   // the LUT and the weights are published
  for (int tile = 0; tile < TILES * TILES; ++tile) {
    const int ty = tile / TILES, tx = tile - ty * TILES;
    pstore();
    __syncthreads();
    if (tile + 1 < TILES * TILES && !STEM_DBG(p, 4)) gload(tile + 1);

    // ---- implicit GEMM: 14 steps of 16 reduction elements = (kernel row, half of its 8 pixels)
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    if (!STEM_DBG(p, 8))
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      const int aoff = (s >> 1) * (PW * 8) + (s & 1) * 32;
      uint4 fa[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) fa[mt] = *reinterpret_cast<const uint4*>(mainb + abase[mt] + aoff);
      const uint4 fb = *reinterpret_cast<const uint4*>(wimg + bbase + s * 32);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[mt]),
                                                          __builtin_bit_cast(bf16x8, fb), acc[mt], 0, 0, 0);
    }
    // ---- statistics of the owned outputs (fp32 accumulators): lane = channel, fixed order across the waves
    if (!STEM_DBG(p, 16)) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = ((own[mt] >> e) & 1u) ? acc[mt][e] : 0.f;
          s1 += v;
          s2 = fmaf(v, v, s2);
        }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lh == 0) {
        sred[wave][0][lr] = s1;
        sred[wave][1][lr] = s2;
      }
    }
    __syncthreads();   // every wave is done reading the patch: the raw tile may overwrite it; sred is complete
    if (t < 2 * CW) {
      const int which = t / CW, ch = t - which * CW;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) s += sred[w][which][ch];
      p.part[((img * (TILES * TILES) + ty * TILES + tx) * 2 + which) * COUT + c0 + ch] = s;
    }
    // ---- raw tile, rounded to bf16 as the unfused path stores it: [289][32 + pad]
    if (!STEM_DBG(p, 2))
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = wave * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m < MROWS) *reinterpret_cast<unsigned short*>(mainb + m * RT_PITCH + lr * 2) = avs_f32_to_bf16(acc[mt][e]);
      }
    __syncthreads();
    // ---- 3x3 / 2 max (min) over the raw tile: item = (pooled pixel of the 8 x 8, 8 of the 32 channels).
    // bf16 values compare like their bit patterns after the map key(x) = x ^ ((x >> 15) & 0x7fff) (sign-magnitude ->
    // two's complement order), which is its own inverse: the window runs on PACKED signed 16-bit max (two channels
    // per instruction), the keys are mapped back at the end.  Exact, no float conversion.
    for (int it = t; it < 64 * 4 && !STEM_DBG(p, 1); it += WAVES * 64) {
      const int cg = it & 3, pp = it >> 2;
      const int pyl = pp >> 3, pxl = pp & 7;
      s16x2 kmax[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) kmax[j] = (s16x2){(short)-32768, (short)-32768};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int ly = 2 * pyl + dy;
        if (ty == 0 && ly == 0) continue;   // convolution row -1: outside the map (maxpool pads with -inf)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int lx = 2 * pxl + dx;
          if (tx == 0 && lx == 0) continue;
          const uint4 v = *reinterpret_cast<const uint4*>(mainb + (ly * TW + lx) * RT_PITCH + cg * 16);
          const unsigned vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const s16x2 x = __builtin_bit_cast(s16x2, vv[j]);
            const s16x2 k = x ^ ((x >> 15) & (s16x2){(short)0x7fff, (short)0x7fff}) ^ flip[j];
            kmax[j] = __builtin_elementwise_max(kmax[j], k);
          }
        }
      }
      unsigned mx[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const s16x2 m7 = (s16x2){(short)0x7fff, (short)0x7fff};
        const s16x2 k = kmax[j] ^ flip[j];
        mx[j] = __builtin_bit_cast(unsigned, k ^ ((k >> 15) & m7));
      }
      const long long o = (((img * POOL + 8 * ty + pyl) * POOL + 8 * tx + pxl) * COUT + c0 + cg * 8) * 2;
      *reinterpret_cast<uint4*>(p.osel + o) = make_uint4(mx[0], mx[1], mx[2], mx[3]);
    }
    __syncthreads();   // the raw tile has been read: the next patch may overwrite it
  }
}

// One thread per (group, channel): the group's tile partials added in frame / tile order, folded into the affine
// scale = gamma / sqrt(var + eps), shift = beta - mean * scale (biased variance; bn_fold_kernel's arithmetic).
__global__ __launch_bounds__(256) void stem_fold_kernel(const float* __restrict__ part, int groups, int fpg,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, float* __restrict__ scale, float* __restrict__ shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= groups * COUT) return;
  const int g = i / COUT, ch = i - g * COUT;
  const long long t0 = (long long)g * fpg * (TILES * TILES), t1 = t0 + (long long)fpg * (TILES * TILES);
  float s1 = 0.f, s2 = 0.f;
  for (long long tl = t0; tl < t1; ++tl) {
    s1 += part[(tl * 2) * COUT + ch];
    s2 += part[(tl * 2 + 1) * COUT + ch];
  }
  const float inv_n = 1.f / (float)((long long)fpg * CONV * CONV);
  const float mean = s1 * inv_n;
  const float var = fmaxf(s2 * inv_n - mean * mean, 0.f);
  const float sc = gamma[ch] / sqrtf(var + eps);
  scale[i] = sc;
  shift[i] = beta[ch] - mean * sc;
}

// y = relu(scale * y + shift) IN PLACE on the selected-extreme map, in avs_bn_apply's arithmetic (multiply, add, ReLU,
// round): 8 channels per thread.
__global__ __launch_bounds__(256) void stem_finish_kernel(long long items, int fpg, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int relu, char* y) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < items;
       i += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(i & 7);
    const long long px = i >> 3;
    const long long g = (px / (POOL * POOL)) / fpg;
    const uint4 a = *reinterpret_cast<const uint4*>(y + i * 16);
    const unsigned av[4] = {a.x, a.y, a.z, a.w};
    const float4 s0 = *reinterpret_cast<const float4*>(scale + g * COUT + cg * 8);
    const float4 s1 = *reinterpret_cast<const float4*>(scale + g * COUT + cg * 8 + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(shift + g * COUT + cg * 8);
    const float4 h1 = *reinterpret_cast<const float4*>(shift + g * COUT + cg * 8 + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    unsigned out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = __uint_as_float(av[j] << 16);
      const float x1 = __uint_as_float(av[j] & 0xffff0000u);
      float y0 = x0 * sc[2 * j] + sh[2 * j];
      float y1 = x1 * sc[2 * j + 1] + sh[2 * j + 1];
      if (relu) {
        y0 = fmaxf(y0, 0.f);
        y1 = fmaxf(y1, 0.f);
      }
      out[j] = (unsigned)avs_f32_to_bf16(y0) | ((unsigned)avs_f32_to_bf16(y1) << 16);
    }
    *reinterpret_cast<uint4*>(y + i * 16) = make_uint4(out[0], out[1], out[2], out[3]);
  }
}

extern "C" int64_t avs_stem_workspace_bytes(int n) {
  if (n < 0) return AVS_E_SHAPE;
  // per-tile partial sums fp32 [n * 49, 2, 64] (the pooled map goes straight to the output)
  return (int64_t)n * TILES * TILES * 2 * COUT * 4;
}

extern "C" int avs_stem_conv_bn_pool_bf16(const uint8_t* d_frames, int n, float denom, const float* mean3,
                                          const float* std3, const void* d_w, int64_t ldw, int frames_per_group,
                                          const float* d_gamma, const float* d_beta, float eps, int apply, int relu,
                                          void* d_y, float* d_scale, float* d_shift, void* d_ws, int64_t ws_bytes,
                                          avs_stream_t stream) {
  const char* who = "avs_stem_conv_bn_pool_bf16";
  AVS_REQUIRE(n >= 0 && frames_per_group > 0, AVS_E_SHAPE, "%s: n=%d frames_per_group=%d", who, n, frames_per_group);
  if (n == 0) return AVS_OK;
  AVS_REQUIRE(n % frames_per_group == 0, AVS_E_UNSUPPORTED, "%s: needs equal-sized groups (%d frames in groups of %d)",
              who, n, frames_per_group);
  AVS_REQUIRE(d_frames && mean3 && std3 && d_w && d_gamma && d_beta && d_y && d_scale && d_shift && d_ws, AVS_E_ARG,
              "%s: null pointer", who);
  AVS_REQUIRE(denom != 0.f, AVS_E_ARG, "%s: denom == 0", who);
  AVS_REQUIRE(ldw >= KDIM && ldw % 8 == 0, AVS_E_SHAPE, "%s: weight rows are 7 x 8 x 4 = 224 elements, stride %lld", who,
              (long long)ldw);
  AVS_REQUIRE(avs_aligned16(d_w) && avs_aligned16(d_y) && avs_aligned16(d_ws) && avs_aligned16(d_scale) &&
                  avs_aligned16(d_shift),
              AVS_E_ALIGN, "%s: operands must be 16-byte aligned", who);
  const int64_t need = avs_stem_workspace_bytes(n);
  AVS_REQUIRE(ws_bytes >= need, AVS_E_WORKSPACE, "%s: workspace %lld < %lld bytes", who, (long long)ws_bytes,
              (long long)need);
  AVS_REQUIRE((long long)n * TILES * 2 < (1ll << 31), AVS_E_SHAPE, "%s: too many frames", who);
  StemParams p{};
  p.frames = d_frames;
  p.w = (const char*)d_w;
  p.ldw = ldw;
  p.osel = (char*)d_y;
  p.gamma = d_gamma;
  p.part = reinterpret_cast<float*>(d_ws);
  p.denom = denom;
#ifdef AVS_STUDY
  p.debug = getenv("AVS_STEM_DEBUG") ? atoi(getenv("AVS_STEM_DEBUG")) : 0;
#endif
  for (int c = 0; c < 3; ++c) {
    p.mean[c] = mean3[c];
    p.stdv[c] = std3[c];
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(stem_fused_kernel, dim3((unsigned)(n * 2)), dim3(WAVES * 64), 0, st, p);
  const int groups = n / frames_per_group;
  hipLaunchKernelGGL(stem_fold_kernel, dim3((unsigned)avs_cdiv((int64_t)groups * COUT, 256)), dim3(256), 0, st, p.part,
                     groups, frames_per_group, d_gamma, d_beta, eps, d_scale, d_shift);
  if (apply) {
    const long long items = (long long)n * POOL * POOL * (COUT / 8);
    long long gx = avs_cdiv(items, 256);
    if (gx > 65536) gx = 65536;
    hipLaunchKernelGGL(stem_finish_kernel, dim3((unsigned)gx), dim3(256), 0, st, items, frames_per_group, d_scale,
                       d_shift, relu, (char*)d_y);
  }
  AVS_CHECK_LAUNCH(who);
  return AVS_OK;
}
