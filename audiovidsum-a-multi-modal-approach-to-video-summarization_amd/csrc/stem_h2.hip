// The ResNet-50 stem of the parity-grade (AVS_F16X2) path in two launches: uint8 frames -> conv1 7x7/2 -> batch statistics
// of bn1 per micro-batch group -> 3x3/2 pooling of the RAW output (features/extractors.py:29,65,126-140: children()[0:4]
// of the trunk on the (x - mean)/std input, no /255).  bn1 + ReLU are applied by the first consumer of the pooled map
// (avs_bn_gram_affine_f16x2's input affine: it reads the map anyway and stores the finished activation in place).
//
// Unfused, this stage was frames_normalize (150 KB in, 853 KB out per frame) -> a K-padded 7x7 contraction (853 KB in,
// 3.2 MB raw out) -> bn_maxpool (3.2 MB in, 803 KB out): 7.6 % of the step and ~11 MB of HBM traffic per frame for a
// stage whose algorithmic I/O is 150 KB in + 803 KB out.  Three things make the one-kernel form cheap in this format:
//
//  (1) THE IMAGE NEEDS NO lo HALF.  x = (v / denom - mean_c) / std_c is affine in the byte v, so
//          conv(x)[o] = sum_taps (w[o,tap,c] / (denom std_c)) v[tap,c]  -  sum_{taps inside the image} w[o,tap,c] mean_c / std_c
//      The bytes 0..255 are EXACT fp16 numbers: the A operand is the raw image (one fp16 per value, nothing to split),
//      the B operand the rescaled weights w' = w / (denom std_c) as fp16 hi | lo (22 bits, prepared once per parameter
//      version) - two MFMAs per product instead of three, and no rounding of the input at all (the unfused path rounded
//      x to 22 bits).  The second sum counts only the taps INSIDE the image (the convolution pads the NORMALISED input with
//      zeros): it rides in the pixel's fourth channel, which the 4-channel layout pads anyway - 1.0 for a pixel inside the
//      image, 0.0 outside - against the weight -sum_c w[o,tap,c] mean_c / std_c: no border cases anywhere in the kernel.
//  (2) POOLING DOES NOT WAIT FOR THE STATISTICS (the argument of stem.hip, format-independent): y -> relu(scale y +
//      shift) -> split is monotone and sign(scale) = sign(gamma), so the window's max (gamma >= 0) or min (gamma < 0) of
//      the RAW fp32 outputs is pooled on chip and only that leaves: ONE 56x56x64 map.
//  (3) CENTRED STATISTICS WITHOUT A SECOND ROUND: sums of (y - p) and (y - p)^2 about a per-channel pivot p = the frame's
//      own output at an interior pixel of its first tile (for a constant frame every interior output EQUALS p: exact zeros;
//      in general |p - mean| is a few standard deviations: one or two bits).  A lane keeps its two sums in registers for
//      the whole frame (49 tile sums of 32 values each), the frame's (mean, M2) leave once, frames of a group are merged by Chan's update in
//      frame order: deterministic, nothing of the E[y^2] - E[y]^2 form about the origin.
//
// One workgroup (4 waves) = one frame x 32 of the 64 output channels, walking the frame's 56 tiles of 8 x 7 pooled outputs =
// 17 x 15 convolution outputs (16 x 14 owned + the pooling halo row / column, recomputed) = 255 GEMM rows: eight 32-row
// blocks, two per wave - the four SIMDs carry the same matrix work.  Patch 39 x 36 pixels of 4 x fp16 in LDS (one unaligned
// dword per pixel fetched into registers a tile ahead), A fragment = the 16 bytes of two neighbouring patch pixels (K = 7
// kernel rows x 8 pixels x 4 channels = 224, zero weights in the padding), weights resident in LDS as f16x2 rows.
// THE PHASES OVERLAP INSIDE EVERY WAVE: the patch and the raw tile have their own LDS regions, so the pooling of tile i - 1
// (LDS reads + VALU, then the stores of its 56 pooled pixels) is written between the matrix steps of tile i - the matrix
// pipe and the vector / LDS pipes are separate, and a wave's MFMAs run on while it issues the pooling.  Two barriers per
// tile: [patch i and raw i - 1 ready] -> matrix work i || pooling i - 1 -> [both read] -> statistics + raw tile i.
// (The first version ran staging, matrix, statistics, raw tile and pooling one after the other between three barriers, five
// waves on four SIMDs: 24.8 ms per 11 286 frames, the matrix pipe 19 % busy.)
#include "avs_internal.h"
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int IMG = 224, CONV = 112, POOL = 56, COUT = 64;
constexpr int TY = 7, TX = 8;            // tiles per frame: 7 rows of 8
constexpr int TH = 17, TWD = 15;         // convolution outputs per tile: 17 rows x 15 columns (16 x 14 owned + the halo)
constexpr int PPY = 8, PPX = 7;          // pooled outputs per tile
constexpr int MROWS = TH * TWD;          // 255 used rows of the 256-row GEMM tile
constexpr int PH = 39, PW = 36;          // patch rows / pixels (pixel = 4 x fp16 = 8 bytes)
constexpr int PATCH_BYTES = PH * PW * 8;
constexpr int KDIM = 224;                // 7 kernel rows x 8 pixels x 4 channels
constexpr int CW = 32;                   // output channels per workgroup
constexpr int W_PITCH = KDIM * 4 + 16;   // an f16x2 weight row (hi8 | lo8 per 32 bytes) + 16: 16 rows -> 16 different slots
constexpr int W_BYTES = CW * W_PITCH;
constexpr int RT_PITCH = 144;            // raw tile row: 32 fp32 + 16 bytes
constexpr int RT_BYTES = 256 * RT_PITCH;
constexpr int WAVES = 4, THREADS = WAVES * 64;
constexpr int SRED_BYTES = WAVES * 2 * CW * 4, PIV_BYTES = CW * 4;
constexpr int SMEM_BYTES = W_BYTES + PATCH_BYTES + RT_BYTES + SRED_BYTES + PIV_BYTES;
constexpr int PIVOT_ROW = 9 * TWD + 7;   // tile row of the pivot: convolution output (8, 6) of tile (0, 0), an interior pixel
static_assert(SMEM_BYTES <= 80 * 1024, "two workgroups per CU");
}  // namespace

struct StemH2Params {
  const uint8_t* frames;
  const char* w;       // f16x2 [64][ldw slots], 7 x 8 x 4 layout: w / (denom std_c) | channel 3: -sum_c w mean_c / std_c
  const float* gamma;
  char* osel;          // f16x2 [n,56,56,64]: per channel the window's max (gamma >= 0) or min (gamma < 0) of the raw output
  float* part;         // [n][2][64]: the frame's mean and its sum of squares about that mean, per channel
  long long ldw;
};

struct __attribute__((packed, aligned(1))) avs_u32_unaligned { unsigned v; };

__global__ __launch_bounds__(THREADS, 2) void stem_h2_kernel(StemH2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wimg = smem;
  char* patch = smem + W_BYTES;
  char* rawt = patch + PATCH_BYTES;
  float* sred = reinterpret_cast<float*>(rawt + RT_BYTES);                      // [WAVES][2][CW]
  float* piv_s = sred + WAVES * 2 * CW;                                         // [CW]
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int lr = lane & 31, lh = lane >> 5;
  const int c0 = (blockIdx.x & 1) * CW;            // this workgroup's channels c0 .. c0 + 31
  const long long img = blockIdx.x >> 1;           // ... of one frame
  const uint8_t* __restrict__ src = p.frames + img * (long long)(IMG * IMG * 3);

  // weights -> LDS once per workgroup: 32 rows x 56 chunks of 16 bytes
  for (int i = t; i < CW * (KDIM / 4); i += THREADS) {
    const int n = i / (KDIM / 4), ch = i - n * (KDIM / 4);
    *reinterpret_cast<uint4*>(wimg + n * W_PITCH + ch * 16) =
        *reinterpret_cast<const uint4*>(p.w + ((long long)(c0 + n) * p.ldw) * 4 + ch * 16);
  }
  // per-lane constants: A base offsets of this lane's two 32-row blocks; per accumulator element (bit 16 mt + e) whether
  // it is owned (not the halo row / column, not the tile's padding row 255)
  int abase[2];
  unsigned own = 0;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int m = wave * 64 + mt * 32 + lr;
    if (m >= MROWS) m = MROWS - 1;   // row 255 computes garbage from a valid address; it is stored to the raw tile's spare row
    const int ly = m / TWD, lx = m - ly * TWD;
    abase[mt] = ((2 * ly) * PW + 2 * lx) * 8 + lh * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int me = wave * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      const int ey = me / TWD, ex = me - ey * TWD;
      if (me < MROWS && ey >= 1 && ex >= 1) own |= 1u << (16 * mt + e);
    }
  }
  const int bbase = lr * W_PITCH + lh * 32;
  // pooling: an item = (pooled pixel, 8 channels), threads 0 .. 223; the channels with gamma < 0 pool -y
  // (threads 224 .. 255 run item 0's arithmetic too and store nothing: no divergent block between the matrix instructions)
  const bool pool_on = t < PPY * PPX * 4;
  const int p_cg = t & 3, p_pp = pool_on ? (t >> 2) : 0;
  const int p_pyl = p_pp / PPX, p_pxl = p_pp - p_pyl * PPX;
  float sg[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) sg[j] = p.gamma[c0 + p_cg * 8 + j] < 0.f ? -1.f : 1.f;
  const char* praw = rawt + ((2 * p_pyl) * TWD + 2 * p_pxl) * RT_PITCH + p_cg * 32;

  // the patch pixels of a tile are fetched into REGISTERS one tile ahead: ONE unaligned dword per pixel (its 3 bytes and a
  // neighbour's), converted and written to LDS at the top of their tile
  constexpr int PPT = (PH * PW + THREADS - 1) / THREADS;   // patch pixels per thread (6)
  unsigned pbytes[PPT];
  unsigned inside = 0u;
  int pyx[PPT];   // this thread's patch pixels: row << 8 | column
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int i = t + j * THREADS;
    const int py = i < PH * PW ? i / PW : 255;    // beyond the patch: fails every range test below
    pyx[j] = (py << 8) | (i - (i / PW) * PW);
  }
  auto gload = [&](int tile) {
    const int ty = tile / TX, tx = tile - ty * TX;
    const int iy0 = 32 * ty - 5, ix0 = 28 * tx - 5;
    const int ylo = iy0 < 0 ? -iy0 : 0, yhi = IMG - iy0 < PH ? IMG - iy0 : PH;
    const int xlo = ix0 < 0 ? -ix0 : 0, xhi = IMG - ix0 < PW ? IMG - ix0 : PW;
    inside = 0u;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int py = pyx[j] >> 8, px = pyx[j] & 255;
      if (py >= ylo && py < yhi && px >= xlo && px < xhi) {
        const int o = ((iy0 + py) * IMG + ix0 + px) * 3;          // the pixel's first byte inside the frame
        // bytes o - 1 .. o + 2 (one byte of the pixel before, then B G R), except for the tensor's very first pixel
        const bool prev = o > 0 || img > 0;
        const unsigned d = reinterpret_cast<const avs_u32_unaligned*>(src + o - (prev ? 1 : 0))->v;
        pbytes[j] = prev ? d >> 8 : d;
        inside |= 1u << j;
      }
    }
  };
  auto pstore = [&]() {
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int i = t + j * THREADS;
      if (i >= PH * PW) break;
      unsigned lo = 0u, hi = 0u;
      if ((inside >> j) & 1u) {   // bytes are exact in fp16 (round-toward-zero conversion of an integer <= 255)
        const float b0 = (float)(pbytes[j] & 255u), b1 = (float)((pbytes[j] >> 8) & 255u), b2 = (float)((pbytes[j] >> 16) & 255u);
        lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(b0, b1));
        hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(b2, 1.f));   // channel 3: "inside the image"
      }
      *reinterpret_cast<uint2*>(patch + i * 8) = make_uint2(lo, hi);
    }
  };
  // the pooled pixel of tile `tile` from the raw tile in LDS: taps read between the matrix steps (tap = 0 .. 8), then finish
  float pm[8];
  auto pool_tap = [&](int tile, int tap) {
    const int ty = tile / TX, tx = tile - ty * TX;
    const int dy = tap / 3, dx = tap - dy * 3;
    // convolution row / column -1 lies outside the map (maxpool pads with -inf): that tap's products are replaced by -inf
    // (a select, not a branch: the pooling sits between matrix instructions)
    const bool out = (dy == 0 && ty == 0 && p_pyl == 0) || (dx == 0 && tx == 0 && p_pxl == 0);
    const char* row = praw + (dy * TWD + dx) * RT_PITCH;
    const float4 va = *reinterpret_cast<const float4*>(row);
    const float4 vb = *reinterpret_cast<const float4*>(row + 16);
    const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = v[j] * sg[j];
      if (dy == 0 || dx == 0) x = out ? -INFINITY : x;
      pm[j] = tap == 0 ? x : fmaxf(pm[j], x);
    }
  };
  auto pool_finish = [&](int tile) {
    const int ty = tile / TX, tx = tile - ty * TX;
    float v8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v8[j] = pm[j] * sg[j];
    uint4 hi, lo;
    avs_f16x2_split8(v8, hi, lo);
    const long long o = (((img * POOL + PPY * ty + p_pyl) * POOL + PPX * tx + p_pxl) * COUT + c0 + p_cg * 8) * 4;
    if (pool_on) {
      *reinterpret_cast<uint4*>(p.osel + o) = hi;
      *reinterpret_cast<uint4*>(p.osel + o + 16) = lo;
    }
  };

  gload(0);
  float pivot = 0.f, s1 = 0.f, s2 = 0.f;
  constexpr int NT = TY * TX;
  // one tile: POOL = the previous tile's pooling rides between the matrix steps (every tile but the first: the first is
  // peeled so that the loop body is ONE basic block the scheduler can interleave freely)
  auto body = [&](auto pool_c, int tile) {
    constexpr bool POOLING = decltype(pool_c)::value;
    pstore();
    __syncthreads();   // patch(tile) and raw(tile - 1) are complete (and, the first time, the weights)
    if (tile + 1 < NT) gload(tile + 1);

    // ---- implicit GEMM: 14 steps of 16 reduction elements = (kernel row, half of its 8 pixels), hi(w) then lo(w)
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      const int aoff = (s >> 1) * (PW * 8) + (s & 1) * 32;
      uint4 fa[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) fa[mt] = *reinterpret_cast<const uint4*>(patch + abase[mt] + aoff);
      const uint4 fbh = *reinterpret_cast<const uint4*>(wimg + bbase + s * 64);
      const uint4 fbl = *reinterpret_cast<const uint4*>(wimg + bbase + s * 64 + 16);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[mt]), __builtin_bit_cast(f16x8, fbh),
                                                         acc[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[mt]), __builtin_bit_cast(f16x8, fbl),
                                                         acc[mt], 0, 0, 0);
      if constexpr (POOLING) {
        if (s < 9) pool_tap(tile - 1, s);
        if (s == 9) pool_finish(tile - 1);
      }
    }
    if (tile == 0 && wave == PIVOT_ROW / 64 && lh == ((PIVOT_ROW % 32) >> 2 & 1)) {
      constexpr int PMT = (PIVOT_ROW % 64) / 32, PR = PIVOT_ROW % 32;
      constexpr int PE = (PR & 3) + 4 * (PR >> 3);
      piv_s[lr] = acc[PMT][PE];
    }
    __syncthreads();   // every wave is done reading patch(tile) and raw(tile - 1); the pivot is published
    if (tile == 0) pivot = piv_s[lr];
    // ---- statistics of the owned outputs about the pivot (the tile's 32 values summed first, then added to the frame's
    // sums: chains of 32 + 56 terms - on a smooth frame a long chain's rounding errors are correlated and cost 5e-5 of the
    // variance); raw tile (fp32) -> LDS [256][32 + pad]
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = acc[mt][e];
        const float d = ((own >> (16 * mt + e)) & 1u) ? v - pivot : 0.f;
        t1 += d;
        t2 = fmaf(d, d, t2);
        const int m = wave * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        *reinterpret_cast<float*>(rawt + m * RT_PITCH + lr * 4) = v;
      }
    s1 += t1;
    s2 += t2;
  };
  body(std::false_type{}, 0);
  for (int tile = 1; tile < NT; ++tile) body(std::true_type{}, tile);
  // ---- the last tile's pooling
  __syncthreads();
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) pool_tap(NT - 1, tap);
  pool_finish(NT - 1);
  // ---- the frame's statistics: lanes -> waves (fixed order) -> (mean, M2) of this workgroup's 32 channels
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  if (lh == 0) {
    sred[(wave * 2 + 0) * CW + lr] = s1;
    sred[(wave * 2 + 1) * CW + lr] = s2;
  }
  __syncthreads();
  if (t < CW) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      a += sred[(w * 2 + 0) * CW + t];
      b += sred[(w * 2 + 1) * CW + t];
    }
    const float inv_n = 1.f / (float)(CONV * CONV);
    const float dm = a * inv_n;                    // mean - pivot
    p.part[(img * 2 + 0) * COUT + c0 + t] = piv_s[t] + dm;
    p.part[(img * 2 + 1) * COUT + c0 + t] = fmaxf(b - a * dm, 0.f);   // sum (y - mean)^2 = sum d^2 - (sum d)^2 / n
  }
}

// One thread per (group, channel): the frames of a group merged by Chan's update in frame order, folded into the affine
// scale = gamma / sqrt(var + eps), shift = beta - mean * scale (biased variance).
__global__ __launch_bounds__(256) void stem_h2_fold_kernel(const float* __restrict__ part, int groups, int fpg,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, float* __restrict__ scale, float* __restrict__ shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= groups * COUT) return;
  const int g = i / COUT, ch = i - g * COUT;
  const float nf = (float)(CONV * CONV);
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int f = 0; f < fpg; ++f) {
    const long long fr = (long long)g * fpg + f;
    const float mf = part[(fr * 2 + 0) * COUT + ch], qf = part[(fr * 2 + 1) * COUT + ch];
    const float tot = n + nf, delta = mf - mean;
    mean += delta * (nf / tot);
    m2 += qf + delta * delta * (n * nf / tot);
    n = tot;
  }
  const float var = m2 / n;
  const float sc = gamma[ch] / sqrtf(var + eps);
  scale[i] = sc;
  shift[i] = beta[ch] - mean * sc;
}

extern "C" int64_t avs_stem_f16x2_workspace_bytes(int n) {
  if (n < 0) return AVS_E_SHAPE;
  return (int64_t)n * 2 * COUT * 4;   // per frame: mean and centred sum of squares, fp32 [n][2][64]
}

extern "C" int avs_stem_conv_pool_f16x2(const uint8_t* d_frames, int n, const void* d_w, int64_t ldw,
                                        int frames_per_group, const float* d_gamma, const float* d_beta, float eps,
                                        void* d_y, float* d_scale, float* d_shift, void* d_ws, int64_t ws_bytes,
                                        avs_stream_t stream) {
  const char* who = "avs_stem_conv_pool_f16x2";
  AVS_REQUIRE(n >= 0 && frames_per_group > 0, AVS_E_SHAPE, "%s: n=%d frames_per_group=%d", who, n, frames_per_group);
  if (n == 0) return AVS_OK;
  AVS_REQUIRE(n % frames_per_group == 0, AVS_E_UNSUPPORTED, "%s: needs equal-sized groups (%d frames in groups of %d)",
              who, n, frames_per_group);
  AVS_REQUIRE(d_frames && d_w && d_gamma && d_beta && d_y && d_scale && d_shift && d_ws, AVS_E_ARG,
              "%s: null pointer", who);
  AVS_REQUIRE(ldw >= KDIM && ldw % 8 == 0, AVS_E_SHAPE, "%s: weight rows are 7 x 8 x 4 = 224 slots, stride %lld", who,
              (long long)ldw);
  AVS_REQUIRE((((uintptr_t)d_w) & 31u) == 0 && (((uintptr_t)d_y) & 31u) == 0 && avs_aligned16(d_ws) &&
                  avs_aligned16(d_scale) && avs_aligned16(d_shift),
              AVS_E_ALIGN, "%s: AVS_F16X2 operands must be 32-byte aligned, the others 16-byte", who);
  const int64_t need = avs_stem_f16x2_workspace_bytes(n);
  AVS_REQUIRE(ws_bytes >= need, AVS_E_WORKSPACE, "%s: workspace %lld < %lld bytes", who, (long long)ws_bytes,
              (long long)need);
  AVS_REQUIRE((long long)n * 2 < (1ll << 31), AVS_E_SHAPE, "%s: too many frames", who);
  StemH2Params p{};
  p.frames = d_frames;
  p.w = (const char*)d_w;
  p.ldw = ldw;
  p.gamma = d_gamma;
  p.osel = (char*)d_y;
  p.part = reinterpret_cast<float*>(d_ws);
  hipStream_t st = (hipStream_t)stream;
  AVS_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(stem_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SMEM_BYTES) == hipSuccess,
              AVS_E_HIP, "%s: cannot reserve %d bytes of LDS", who, SMEM_BYTES);
  hipLaunchKernelGGL(stem_h2_kernel, dim3((unsigned)(n * 2)), dim3(THREADS), SMEM_BYTES, st, p);
  const int groups = n / frames_per_group;
  hipLaunchKernelGGL(stem_h2_fold_kernel, dim3((unsigned)avs_cdiv((int64_t)groups * COUT, 256)), dim3(256), 0, st, p.part,
                     groups, frames_per_group, d_gamma, d_beta, eps, d_scale, d_shift);
  AVS_CHECK_LAUNCH(who);
  return AVS_OK;
}
