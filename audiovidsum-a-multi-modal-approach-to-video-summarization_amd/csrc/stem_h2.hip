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
// One workgroup (5 waves) = one frame x 32 of the 64 output channels, walking the frame's 49 tiles of 8 x 8 pooled outputs =
// 17 x 17 convolution outputs (16 x 16 owned + the pooling halo, recomputed).  Patch 39 x 39 pixels of 4 x fp16 in LDS (bytes
// fetched into registers one tile ahead), A fragment = the 16 bytes of two neighbouring patch pixels (K = 7 kernel rows x
// 8 pixels x 4 channels = 224, zero weights in the padding), weights resident in LDS as f16x2 rows, v_mfma_f32_32x32x16_f16.
#include "avs_internal.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int IMG = 224, CONV = 112, POOL = 56, COUT = 64;
constexpr int TILES = 7;                 // tiles per side: 7 x 8 pooled = 56
constexpr int TW = 17;                   // convolution outputs per tile side (16 owned + 1 halo)
constexpr int MROWS = TW * TW;           // 289 used rows of the 320-row GEMM tile
constexpr int PH = 39, PW = 40;          // patch rows / pixels (pixel = 4 x fp16 = 8 bytes)
constexpr int PATCH_BYTES = PH * PW * 8;
constexpr int KDIM = 224;                // 7 kernel rows x 8 pixels x 4 channels
constexpr int CW = 32;                   // output channels per workgroup
constexpr int W_PITCH = KDIM * 4 + 16;   // an f16x2 weight row (hi8 | lo8 per 32 bytes) + 16: 16 rows -> 16 different slots
constexpr int W_BYTES = CW * W_PITCH;
constexpr int RT_PITCH = 160;            // raw tile row: 32 fp32 + 32 bytes (40 dwords: the pooling's b128 reads are conflict-free)
constexpr int RT_BYTES = MROWS * RT_PITCH;
constexpr int MAIN_BYTES = RT_BYTES > PATCH_BYTES ? RT_BYTES : PATCH_BYTES;
constexpr int WAVES = 5, THREADS = WAVES * 64;
constexpr int SRED_BYTES = WAVES * 2 * CW * 4, PIV_BYTES = CW * 4;
constexpr int SMEM_BYTES = W_BYTES + MAIN_BYTES + SRED_BYTES + PIV_BYTES;
constexpr int PIVOT_ROW = 9 * TW + 9;    // tile row of the pivot: convolution output (8, 8) of tile (0, 0), an interior pixel
}  // namespace

struct StemH2Params {
  const uint8_t* frames;
  const char* w;       // f16x2 [64][ldw slots], 7 x 8 x 4 layout: w / (denom std_c) | channel 3: -sum_c w mean_c / std_c
  const float* gamma;
  char* osel;          // f16x2 [n,56,56,64]: per channel the window's max (gamma >= 0) or min (gamma < 0) of the raw output
  float* part;         // [n][2][64]: the frame's mean and its sum of squares about that mean, per channel
  long long ldw;
};

// (two workgroups per CU = 10 waves: three waves on some SIMDs -> at most 168 registers)
__global__ __launch_bounds__(THREADS, 3) void stem_h2_kernel(StemH2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wimg = smem;
  char* mainb = smem + W_BYTES;
  float* sred = reinterpret_cast<float*>(smem + W_BYTES + MAIN_BYTES);          // [WAVES][2][CW]
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
  // it is owned (not the halo row / column)
  int abase[2];
  unsigned own = 0;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int m = wave * 64 + mt * 32 + lr;
    if (m >= MROWS) m = MROWS - 1;   // rows 289 .. 319 compute garbage from a valid address and are never stored
    const int ly = m / TW, lx = m - ly * TW;
    abase[mt] = ((2 * ly) * PW + 2 * lx) * 8 + lh * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int me = wave * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      const int ey = me / TW, ex = me - ey * TW;
      const unsigned bit = 1u << (16 * mt + e);
      if (me < MROWS && ey >= 1 && ex >= 1) own |= bit;
    }
  }
  const int bbase = lr * W_PITCH + lh * 32;
  // pooling: an item = (pooled pixel, 8 channels); the channels with gamma < 0 pool -y (max of -y = -min of y)
  const int p_cg = t & 3, p_pp = (t >> 2) & 63;    // threads 0 .. 255
  const int p_pyl = p_pp >> 3, p_pxl = p_pp & 7;
  // the item's two 16-byte chunks are read in an order that alternates with the pooled column (bank spread, below):
  // chunk a = channels ja .. ja + 3 of the item's 8, chunk b the other four
  const int p_ja = (p_pxl & 1) * 4, p_jb = 4 - p_ja;
  float sga[4], sgb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sga[j] = p.gamma[c0 + p_cg * 8 + p_ja + j] < 0.f ? -1.f : 1.f;
    sgb[j] = p.gamma[c0 + p_cg * 8 + p_jb + j] < 0.f ? -1.f : 1.f;
  }

  // the patch pixels of a tile are fetched into REGISTERS one tile ahead (byte loads issued before the matrix work of the
  // previous tile), converted and written to LDS at the top of their tile
  constexpr int PPT = (PH * PW + THREADS - 1) / THREADS;   // patch pixels per thread (5)
  unsigned pb0[PPT], pb1[PPT], pb2[PPT];
  unsigned inside = 0u;
  int pyx[PPT];   // this thread's patch pixels: row << 8 | column
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int i = t + j * THREADS;
    const int py = i < PH * PW ? i / PW : 255;    // beyond the patch: fails every range test below
    pyx[j] = (py << 8) | (i - (i / PW) * PW);
  }
  auto gload = [&](int tile) {
    const int ty = tile / TILES, tx = tile - ty * TILES;
    const int iy0 = 32 * ty - 5, ix0 = 32 * tx - 5;
    const int ylo = iy0 < 0 ? -iy0 : 0, yhi = IMG - iy0 < PH ? IMG - iy0 : PH;
    const int xlo = ix0 < 0 ? -ix0 : 0, xhi = IMG - ix0 < PH ? IMG - ix0 : PH;   // (column 39 is layout padding)
    const uint8_t* org = src + (iy0 * IMG + ix0) * 3;
    inside = 0u;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int py = pyx[j] >> 8, px = pyx[j] & 255;
      if (py >= ylo && py < yhi && px >= xlo && px < xhi) {
        const uint8_t* s = org + (py * IMG + px) * 3;
        pb0[j] = s[0];
        pb1[j] = s[1];
        pb2[j] = s[2];
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
        lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz((float)pb0[j], (float)pb1[j]));
        hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz((float)pb2[j], 1.f));   // channel 3: "inside the image"
      }
      *reinterpret_cast<uint2*>(mainb + i * 8) = make_uint2(lo, hi);
    }
  };
  gload(0);
  __syncthreads();   // the weights are published
  float pivot = 0.f, s1 = 0.f, s2 = 0.f;
  for (int tile = 0; tile < TILES * TILES; ++tile) {
    const int ty = tile / TILES, tx = tile - ty * TILES;
    pstore();
    __syncthreads();
    if (tile + 1 < TILES * TILES) gload(tile + 1);

    // ---- implicit GEMM: 14 steps of 16 reduction elements = (kernel row, half of its 8 pixels); hi(w) then lo(w)
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
      for (int mt = 0; mt < 2; ++mt) fa[mt] = *reinterpret_cast<const uint4*>(mainb + abase[mt] + aoff);
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
    }
    if (tile == 0 && wave == PIVOT_ROW / 64 && lh == ((PIVOT_ROW % 32) >> 2 & 1)) {
      // (row 162 = wave 2, block 1, row 2 of the block: element e with (e & 3) + 8 (e >> 2) + 4 lh == 2 -> e = 2, lh = 0)
      constexpr int PMT = (PIVOT_ROW % 64) / 32, PR = PIVOT_ROW % 32;
      constexpr int PE = (PR & 3) + 4 * (PR >> 3);
      piv_s[lr] = acc[PMT][PE];
    }
    __syncthreads();   // every wave is done reading the patch: the raw tile may overwrite it; the pivot is published
    if (tile == 0) pivot = piv_s[lr];
    // ---- statistics of the owned outputs about the pivot (the tile's 32 values summed first, then added to the frame's
    // sums: chains of 32 + 49 terms instead of 1568 - on a smooth frame the long chain's rounding errors are correlated
    // and cost 5e-5 of the variance); raw tile (fp32) -> LDS [289][32 + pad]
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
        if (m < MROWS) *reinterpret_cast<float*>(mainb + m * RT_PITCH + lr * 4) = v;
      }
    s1 += t1;
    s2 += t2;
    __syncthreads();
    // ---- 3x3 / 2 max (min) over the raw tile: item = (pooled pixel of the 8 x 8, 8 of the 32 channels) -> one 32-byte run.
    // The item's two 16-byte chunks are read in an order that alternates with the pooled column: with the 160-byte row
    // pitch the 16 lanes that share an LDS cycle then touch 16 different 16-byte slots.
    if (t < 256) {
      const int pyl = p_pyl, pxl = p_pxl;
      const int ca = (2 * p_cg) * 16 + p_ja * 4, cb = (2 * p_cg) * 16 + p_jb * 4;
      float ma[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, mb[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int ly = 2 * pyl + dy;
        if (ty == 0 && ly == 0) continue;   // convolution row -1: outside the map (maxpool pads with -inf)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int lx = 2 * pxl + dx;
          if (tx == 0 && lx == 0) continue;
          const char* row = mainb + (ly * TW + lx) * RT_PITCH;
          const float4 va = *reinterpret_cast<const float4*>(row + ca);
          const float4 vb = *reinterpret_cast<const float4*>(row + cb);
          ma[0] = fmaxf(ma[0], va.x * sga[0]);
          ma[1] = fmaxf(ma[1], va.y * sga[1]);
          ma[2] = fmaxf(ma[2], va.z * sga[2]);
          ma[3] = fmaxf(ma[3], va.w * sga[3]);
          mb[0] = fmaxf(mb[0], vb.x * sgb[0]);
          mb[1] = fmaxf(mb[1], vb.y * sgb[1]);
          mb[2] = fmaxf(mb[2], vb.z * sgb[2]);
          mb[3] = fmaxf(mb[3], vb.w * sgb[3]);
        }
      }
      const bool swap = (pxl & 1) != 0;   // chunk a holds the item's channels 4 .. 7
      float v8[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xa = ma[j] * sga[j], xb = mb[j] * sgb[j];
        v8[j] = swap ? xb : xa;
        v8[4 + j] = swap ? xa : xb;
      }
      uint4 hi, lo;
      avs_f16x2_split8(v8, hi, lo);
      const long long o = (((img * POOL + 8 * ty + pyl) * POOL + 8 * tx + pxl) * COUT + c0 + p_cg * 8) * 4;
      *reinterpret_cast<uint4*>(p.osel + o) = hi;
      *reinterpret_cast<uint4*>(p.osel + o + 16) = lo;
    }
    __syncthreads();   // the raw tile has been read: the next patch may overwrite it
  }
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
