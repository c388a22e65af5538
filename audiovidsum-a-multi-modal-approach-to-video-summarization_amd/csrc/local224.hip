// AVS_F16X2 convolution + whole BatchNorm (+ residual, + ReLU) for groups of 193..224 rows - the per-frame 14x14 maps of
// ResNet-50's layer 3 - on a tile that FITS the group (features/extractors.py:65: self.resnet(resnet_batch), train-mode
// BatchNorm per micro-batch group).
//
// The general kernel (igemm.hip, EPI_BNLOCAL) puts such a group into a 256-row tile whose four waves own 64 rows each: a
// quarter of the matrix work and of the operand DMA is spent on rows that do not exist, and four waves cannot share seven
// 32-row blocks evenly by ROWS.  Here the tile is 224 rows x 128 columns and the waves split the COLUMNS: every wave owns
// all seven 32-row blocks of its 32 columns (7 accumulator blocks = 112 registers).
//   * 7/8 of the matrix instructions and 352 instead of 384 staged rows per reduction step;
//   * a column's statistics are sums over ONE wave's registers: two shuffle-free rounds (sum -> mean -> centred squares)
//     plus one cross-half exchange each - no LDS tables, no barriers between the rounds;
//   * fragment reads per step and wave: 14 (A, shared by the four waves) + 2 (B) ds_read_b128 for 21 MFMAs - 0.76 per
//     MFMA, inside what the LDS sustains next to the DMA writes (2 per MFMA gap are free).
// Operand staging, the scalar tap walk, the LDS image (64-byte rows, chunk q of row r in slot q ^ ((r >> 2) & 3)), the
// three-buffer pipeline with hand-counted waits and the order of the three fp16 MFMAs per product are those of
// igemm_kernel<..., PIPE, FASTK, SPLIT = 2>: the convolution itself is bit-identical to the 256-row form, the statistics
// are summed in a different (fixed) order.  Waves 0 and 1 issue six DMA instructions per step, waves 2 and 3 five (352 rows
// = 5.5 passes of the 256 threads); each wave waits on its own count.
#include "avs_internal.h"
#include "igemm_params.h"
#include <type_traits>
#include <utility>

namespace {
constexpr int L_ROWS = 224;                 // A rows of a tile
constexpr int L_MT = L_ROWS / 32;           // 32-row blocks per wave
constexpr int L_BN = 128;                   // columns of a tile (32 per wave)
constexpr int L_CPRR = 4;                   // 16-byte slots per 64-byte LDS row
constexpr int L_STEP = 16;                  // reduction elements (slots) per step
constexpr int L_BUF = (L_ROWS + L_BN) * L_CPRR;   // uint4 slots per operand buffer
constexpr int L_P = 32 + 8;                 // words per staged output row
constexpr unsigned L_OOB = 0x80000000u;

template <int OFF>
__device__ __forceinline__ uint4 lds_read_b128_at(unsigned byte_addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
  return v;
}
template <int... I>
__device__ __forceinline__ void read_blocks(uint4 (&f)[L_MT], unsigned addr, std::integer_sequence<int, I...>) {
  ((f[I] = lds_read_b128_at<I * 32 * 64>(addr)), ...);
}
}  // namespace

template <bool SPATIAL>
__global__ __launch_bounds__(256, 2) void igemm_h2_local224_kernel(IgemmParams p) {
  __shared__ uint4 lds[3 * L_BUF];

  // XCD-aware, bijective block remap (as igemm_kernel): blocks that share an XCD take consecutive tiles
  const unsigned nwg = gridDim.x, orig = blockIdx.x;
  const unsigned q = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
  const unsigned wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
  int tn = wg % p.tiles_n;
  int tm = wg / p.tiles_n;
  if (p.cluster > 1) {
    // clustered form: the tiles of a group wait for each other, so they take CONSECUTIVE block ids (dispatched together);
    // XCD placement (round-robin over block ids: speed only, never correctness) keeps a row tile's column tiles together
    const unsigned c = (unsigned)p.cluster, tnn = (unsigned)p.tiles_n;
    const unsigned tiles_m = nwg / tnn;
    const unsigned full = (8u % c == 0u) ? (tiles_m / 8u) * 8u : 0u;   // row tiles covered by whole super-units of 8
    if (orig < full * tnn) {
      // super-unit = 8 row tiles x all column tiles, block = column tile * 8 + row tile: a row tile's column tiles all sit on
      // XCD (row tile % 8) and share its L2 copy of the A rows; the tiles of a cluster (c | 8) are c consecutive blocks
      const unsigned su = orig / (8u * tnn), l = orig - su * (8u * tnn);
      tn = (int)(l >> 3);
      tm = (int)(su * 8u + (l & 7u));
    } else {
      // (cluster sizes that do not divide 8, and the last row tiles of a launch) unit = cluster x column tiles
      const unsigned o2 = orig - full * tnn, unit = c * tnn;
      const unsigned u = o2 / unit, l = o2 - u * unit;
      tn = (int)(l / c);
      tm = (int)(full + u * c + l % c);
    }
  }
  const int m0 = tm * p.tile_rows;
  const int n0 = tn * L_BN;
  const int used = m0 + p.tile_rows <= p.M ? p.tile_rows : p.M - m0;   // rows of this tile that exist (one group)

  const char* __restrict__ x = p.x;
  const char* __restrict__ w = p.w;
  char* __restrict__ y = p.y;

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const bool extra = wave < 2;                      // this wave stages rows 192 .. 223 too
  const int c = t & (L_CPRR - 1);
  const int rb = t >> 2;                            // 0 .. 63
  const int cq = c ^ ((rb >> 2) & (L_CPRR - 1));    // the k-chunk this thread fetches into slot c

  // ---- staging state: per-row tap masks and 32-bit buffer offsets (vector), the tap walk (scalar) ----
  unsigned amask[4], aoff[4], boff[2];
  int f_tap = 0, f_ci0 = 0, f_kw = 0;
  int f_koff = 0;   // bytes: the current tap + channel block relative to a row's first tap
  int f_kb = 0;     // bytes: the current k step inside a B row
  __amdgpu_buffer_rsrc_t a_rsrc, b_rsrc;
  {
    const int taps = p.K / p.cin;
    long long a_origin;
    int n_first = 0;
    if (p.lin_stride >= 0) {
      a_origin = (long long)m0 * p.lin_stride;
    } else {
      n_first = m0 / p.HoWo;
      a_origin = (long long)n_first * p.x_img_stride - (long long)p.ph * p.x_row_stride - (long long)p.pw * p.x_px_stride;
    }
    a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x) + a_origin * 4, 0, (int)L_OOB, 0x00020000);
    b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(w) + (p.w_kstep ? (long long)n0 * 64 : (long long)n0 * p.ldb * 4), 0, (int)L_OOB, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned mk = 0, off = L_OOB;
      const int r = rb + 64 * i;
      if (r < used) {
        const int m = m0 + r;
        if (p.lin_stride >= 0) {
          mk = ~0u;
          off = (unsigned)((long long)r * p.lin_stride * 4) + cq * 16;
        } else {
          const int n = m / p.HoWo;
          const int rem = m - n * p.HoWo;
          const int ho = rem / p.Wo;
          const int wo = rem - ho * p.Wo;
          const int hi0 = ho * p.sh - p.ph, wi0 = wo * p.sw - p.pw;
          if constexpr (SPATIAL) {
            for (int tp = 0; tp < taps; ++tp) {
              const int th = tp / p.KW, tw = tp - th * p.KW;
              if ((unsigned)(hi0 + th) < (unsigned)p.H && (unsigned)(wi0 + tw) < (unsigned)p.W) mk |= 1u << tp;
            }
          } else {
            mk = ~0u;
          }
          off = (unsigned)(((long long)(n - n_first) * p.x_img_stride + (long long)(hi0 + p.ph) * p.x_row_stride +
                            (long long)(wi0 + p.pw) * p.x_px_stride) * 4) + cq * 16;
        }
      }
      amask[i] = mk;
      aoff[i] = off;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      boff[i] = p.w_kstep ? (unsigned)((rb + 64 * i) * 64) + (unsigned)(cq >> 2) * (unsigned)p.N * 64u + (cq & 3) * 16
                          : (unsigned)((long long)(rb + 64 * i) * p.ldb * 4) + cq * 16;
  }

  auto stage = [&](int buf) {
    uint4* abuf = lds + buf * L_BUF + wave * 64;        // wave-uniform; lane l lands at +l
    uint4* bbuf = abuf + L_ROWS * L_CPRR;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      unsigned off = aoff[i];
      if constexpr (SPATIAL) off = ((amask[i] >> f_tap) & 1u) ? off : L_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(abuf + 256 * i), 16,
                                               (int)off, f_koff, 0, 0);
    }
    if (extra) {
      unsigned off = aoff[3];
      if constexpr (SPATIAL) off = ((amask[3] >> f_tap) & 1u) ? off : L_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(abuf + 256 * 3), 16,
                                               (int)off, f_koff, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (__attribute__((address_space(3))) void*)(bbuf + 256 * i), 16,
                                               (int)boff[i], f_kb, 0, 0);
    f_kb += p.w_kstep ? p.N * 64 : L_STEP * 4;
    // next step (scalar): the same tap's next channel block, or the next tap
    f_ci0 += L_STEP;
    f_koff += L_STEP * 4;
    if (f_ci0 == p.cin) {
      f_ci0 = 0;
      ++f_tap;
      f_koff += (int)((p.x_px_stride - p.cin) * 4);
      if (++f_kw == p.KW) {
        f_kw = 0;
        f_koff += (int)((p.x_row_stride - (long long)p.KW * p.x_px_stride) * 4);
      }
    }
  };

  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[L_MT];
#pragma unroll
  for (int i = 0; i < L_MT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // per-lane fragment byte offsets inside a buffer: a row's chunks alternate hi8 | lo8; this lane half feeds elements
  // 8 * lh .. + 7 of the step: hi = chunk 2 * lh, lo = chunk 2 * lh + 1.  Block mt of A is mt * 32 rows further: an
  // immediate offset of the read (the swizzle term only depends on the row's low bits).
  const unsigned lds_base = (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)lds);
  unsigned fa_off[2], fb_off[2];
#pragma unroll
  for (int hl = 0; hl < 2; ++hl) {
    const int chunk = 2 * lh + hl;
    fa_off[hl] = (unsigned)(lr * L_CPRR + (chunk ^ ((lr >> 2) & (L_CPRR - 1)))) * 16u;
    const int brow = wave * 32 + lr;
    fb_off[hl] = (unsigned)((L_ROWS + brow) * L_CPRR + (chunk ^ ((brow >> 2) & (L_CPRR - 1)))) * 16u;
  }

  // The residual rows of the first three blocks are fetched BEFORE the reduction (48 registers through the main loop),
  // the next three when it ends: no memory latency is left in front of the first stores, and the rest land under them.
  const int col = n0 + wave * 32 + lr;
  const float inv_n = 1.f / (float)p.tile_rows;   // (a tile holds exactly one group, or one member of a cluster)
  const int lim = used - 4 * lh;   // block-local row offsets below this one exist
  // the residual runs of a block: staged row rl0 (+ 16) of the block, columns 8 * grp .. + 7 of the wave's 32
  const int rl0 = lane >> 2, grp = lane & 3;
  const int col8 = n0 + wave * 32 + grp * 8;
  constexpr int DEPTH = 6;   // blocks of residual rows in flight
  uint4 rhi[DEPTH][2], rlo[DEPTH][2];
  auto res_fetch = [&](auto blk) {
    constexpr int mt = decltype(blk)::value;
    const long long row0 = (long long)m0 + mt * 32 + rl0;
    const char* base = p.residual + (row0 * p.ldr + col8) * 4;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      // rows that do not exist read the residual's first run (never used): no divergent branch around the loads
      const bool ok = mt * 32 + rl0 + 16 * it < used;
      const uint4* rp = reinterpret_cast<const uint4*>(ok ? base + (long long)it * 16 * p.ldr * 4 : p.residual);
      rhi[mt % DEPTH][it] = rp[0];
      rlo[mt % DEPTH][it] = rp[1];
    }
  };
  if (p.residual) {
    res_fetch(std::integral_constant<int, 0>{});
    res_fetch(std::integral_constant<int, 1>{});
    res_fetch(std::integral_constant<int, 2>{});
  }

  const int steps = p.K / L_STEP;
  stage(0);
  if (steps > 1) stage(1);
  int cur = 0;  // buffer holding step s
  for (int s = 0; s < steps; ++s) {
    if (s + 1 < steps) {
      if (extra)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const unsigned bbase = lds_base + (unsigned)cur * (L_BUF * 16u);
    uint4 fah[L_MT], fal[L_MT], fbh, fbl;
    fbh = avs_lds_read_b128(bbase + fb_off[0]);
    read_blocks(fah, bbase + fa_off[0], std::make_integer_sequence<int, L_MT>{});
    fbl = avs_lds_read_b128(bbase + fb_off[1]);
    read_blocks(fal, bbase + fa_off[1], std::make_integer_sequence<int, L_MT>{});
    // the DMA of step s+2 is issued while the fragment reads are in flight
    if (s + 2 < steps) stage(cur == 0 ? 2 : cur - 1);  // (s+2) % 3 == (cur + 2) % 3
    // hi*hi starts as soon as the hi fragments have landed (LDS reads return in order); every fragment passes through an
    // empty asm after the wait that covers it, so no use of it can be scheduled ahead of that wait
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(1 + L_MT) : "memory");
    avs_pin(fbh);
#pragma unroll
    for (int mt = 0; mt < L_MT; ++mt) avs_pin(fah[mt]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < L_MT; ++mt)
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(avs_f16x8, fah[mt]),
                                                       __builtin_bit_cast(avs_f16x8, fbh), acc[mt], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    avs_pin(fbl);
#pragma unroll
    for (int mt = 0; mt < L_MT; ++mt) avs_pin(fal[mt]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < L_MT; ++mt) {
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(avs_f16x8, fal[mt]),
                                                       __builtin_bit_cast(avs_f16x8, fbh), acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(avs_f16x8, fah[mt]),
                                                       __builtin_bit_cast(avs_f16x8, fbl), acc[mt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    cur = cur == 2 ? 0 : cur + 1;
  }

  // ---- epilogue: the group's BatchNorm from this wave's own registers, then 32 rows at a time through LDS ----
  // A lane holds column lr of its wave's 32, rows (e & 3) + 8 * (e >> 2) + 4 * lh of every block; rows >= used are
  // exact zeros (their operand rows were never fetched).
  if (p.residual) {
    res_fetch(std::integral_constant<int, 3>{});
    res_fetch(std::integral_constant<int, 4>{});
    res_fetch(std::integral_constant<int, 5>{});
  }
  float s1 = 0.f;
#pragma unroll
  for (int mt = 0; mt < L_MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) s1 += acc[mt][e];
  s1 += __shfl_xor(s1, 32, 64);
  const float mean = s1 * inv_n;
  float s2 = 0.f;
#pragma unroll
  for (int mt = 0; mt < L_MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int roff = mt * 32 + (e & 3) + 8 * (e >> 2);
      const float d = acc[mt][e] - mean;
      s2 = roff < lim ? fmaf(d, d, s2) : s2;
    }
  s2 += __shfl_xor(s2, 32, 64);
  float g_mean = mean, g_var = s2 * inv_n;
  if (p.cluster > 1) {
    // ---- one exchange with the other tiles of the group (same columns): lane (lr, lh) publishes the tile's mean (lh = 0) or
    // centred sum of squares (lh = 1) of column lr as an 8-byte {value, epoch} granule - ONE agent-scope store, untorn, so
    // the tag IS the flag: no fence, no separate flag word - then reads the same granule of every partner until its tag is
    // this launch's epoch.  Every wave publishes before it polls (no circular wait); partners are neighbours in dispatch
    // order; the wait is bounded (p.xerr counts the waves that gave up: 0 after a healthy launch).
    const int cj = tm % p.cluster, t0 = tm - cj;
    unsigned long long* slot = p.xchg + (((long long)tm * p.tiles_n + tn) * 4 + wave) * 64 + lane;
    const float mine = lh ? s2 : mean;
    __hip_atomic_store(slot, ((unsigned long long)p.epoch << 32) | (unsigned long long)__float_as_uint(mine), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    const float nt = (float)p.tile_rows;
    float n = 0.f, mu = 0.f, m2 = 0.f;
    const unsigned long long* const gbase = p.xchg + (((long long)t0 * p.tiles_n + tn) * 4 + wave) * 64 + lane;
    const long long gstride = (long long)p.tiles_n * 4 * 64;      // granules between consecutive row tiles
    for (int j0 = 0; j0 < p.cluster; j0 += 4) {
      // four partners' granules in flight at once (one latency for the usual cluster of four), then each is verified and -
      // only if its tag is not this launch's yet - polled on
      unsigned long long g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u < p.cluster ? j0 + u : cj;           // (past the cluster / own tile: a valid address, unused)
        g[u] = __hip_atomic_load(gbase + j * gstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u;
        if (j >= p.cluster) break;
        float mj = mean, qj = s2;
        if (j != cj) {
          bool ok = (unsigned)(g[u] >> 32) == p.epoch;
          for (int spin = 0; spin < (1 << 20) && __builtin_amdgcn_read_exec() != __builtin_amdgcn_ballot_w64(ok); ++spin) {
            __builtin_amdgcn_s_sleep(8);
            g[u] = __hip_atomic_load(gbase + j * gstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (unsigned)(g[u] >> 32) == p.epoch;
          }
          if (__builtin_amdgcn_read_exec() != __builtin_amdgcn_ballot_w64(ok) && lane == 0) atomicAdd(p.xerr, 1u);
          const float v = __uint_as_float((unsigned)g[u]);
          const float o = __shfl_xor(v, 32, 64);
          mj = lh ? o : v;
          qj = lh ? v : o;
        }
        // Chan's update in tile order: the same operands in the same order on every tile of the group -> the same bits
        const float tot = n + nt, delta = mj - mu;
        mu += delta * (nt / tot);
        m2 += qj + delta * delta * (n * nt / tot);
        n = tot;
      }
    }
    g_mean = mu;
    g_var = m2 / n;
  }
  const float sc = p.gamma[col] / sqrtf(g_var + p.eps);
  const float sf = p.beta[col] - g_mean * sc;
#pragma unroll
  for (int mt = 0; mt < L_MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mt][e] = fmaf(acc[mt][e], sc, sf);
  const bool relu = p.act == AVS_ACT_RELU;

  __syncthreads();   // every wave has read its last fragments: the staging regions alias the operand buffers
  float* const wreg = reinterpret_cast<float*>(lds) + wave * (32 * L_P);
  auto emit = [&](auto blk) {
    constexpr int mt = decltype(blk)::value;
#pragma unroll
    for (int e = 0; e < 16; ++e) wreg[((e & 3) + 8 * (e >> 2) + 4 * lh) * L_P + lr] = acc[mt][e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const long long row0 = (long long)m0 + mt * 32 + rl0;
    char* const ybase = y + (row0 * p.ldc + col8) * 4;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const float* src = wreg + (rl0 + 16 * it) * L_P + grp * 8;
      const float4 f0 = *reinterpret_cast<const float4*>(src);
      const float4 f1 = *reinterpret_cast<const float4*>(src + 4);
      if (!(mt * 32 + rl0 + 16 * it < used)) continue;
      float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
      if (p.residual) {
        float rv[8];
        avs_pin(rhi[mt % DEPTH][it]);
        avs_pin(rlo[mt % DEPTH][it]);
        avs_f16x2_join8(rhi[mt % DEPTH][it], rlo[mt % DEPTH][it], rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rv[j];
      }
      if (relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      uint4 hi, lo;
      avs_f16x2_split8(v, hi, lo);
      uint4* dst = reinterpret_cast<uint4*>(ybase + (long long)it * 16 * p.ldc * 4);
      dst[0] = hi;
      dst[1] = lo;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if constexpr (mt + DEPTH < L_MT) {
      if (p.residual) res_fetch(std::integral_constant<int, mt + DEPTH>{});
    }
  };
  emit(std::integral_constant<int, 0>{});
  emit(std::integral_constant<int, 1>{});
  emit(std::integral_constant<int, 2>{});
  emit(std::integral_constant<int, 3>{});
  emit(std::integral_constant<int, 4>{});
  emit(std::integral_constant<int, 5>{});
  emit(std::integral_constant<int, 6>{});
}

// The shapes the 224-row form takes: AVS_F16X2, one group of 193 .. 224 rows per tile, cout in multiples of 128, a
// reduction walked by the scalar tap walk in 64-byte steps inside the 2 GiB buffer window, the caller not asking for
// another tile (avs_conv_desc.variant: AVS_TILE_AUTO or AVS_TILE_224).
bool igemm_h2_local224_ok(const IgemmParams& p, int dtype) {
  const int tile_mode = p.variant & 3;
  if (dtype != AVS_F16X2 || !(tile_mode == AVS_TILE_AUTO || tile_mode == AVS_TILE_224)) return false;
  if (p.variant & AVS_STAGING_GENERIC) return false;
  const int cl = p.cluster > 1 ? p.cluster : 1;
  if (p.tile_rows * cl != p.rows_per_group || p.tile_rows <= 192 || p.tile_rows > L_ROWS) return false;
  if (p.N % L_BN != 0 || p.M % p.rows_per_group != 0) return false;
  if (p.cin % L_STEP != 0 || p.K % L_STEP != 0 || p.K % p.cin != 0 || p.K / p.cin > 32) return false;
  const long long rows = 256;
  long long extent;
  if (p.lin_stride >= 0)
    extent = rows * p.lin_stride + p.K;
  else
    extent = (rows / p.HoWo + 2) * p.x_img_stride + (long long)(p.K / (p.cin * p.KW) + p.ph) * p.x_row_stride +
             (long long)(p.KW + p.pw) * p.x_px_stride + p.cin;
  return extent * 4 < (1ll << 31) && (long long)L_BN * p.ldb * 4 + (long long)p.K * 4 < (1ll << 31) &&
         p.x_img_stride >= 0 && p.x_row_stride >= 0 && p.x_px_stride >= 0;
}

void igemm_h2_local224_launch(const IgemmParams& p, bool spatial, dim3 grid, hipStream_t stream) {
  if (spatial)
    hipLaunchKernelGGL(igemm_h2_local224_kernel<true>, grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(igemm_h2_local224_kernel<false>, grid, dim3(256), 0, stream, p);
}
