// Implicit-GEMM convolution / batched NT GEMM for gfx950 (CDNA4).
//
// One kernel serves every dense contraction of the hot path (SURVEY K3, K6,
// K16-K20): the reduction index k runs over (kh, kw, ci) of an NHWC activation
// tensor, so a 1x1/stride-1 convolution, an nn.Linear and a strided-batched
// GEMM are the same code with different address parameters.
//
// Tile: 128 x BN outputs per 256-thread workgroup (4 waves as 2 x 2); 128 BYTES
// of reduction per LDS row and step (32 fp32 / 64 bf16), so both dtypes share
// the loads, the LDS image and the swizzle.
//
// Staging is LDS-DMA: every thread issues global_load_lds_dwordx4 (16 bytes,
// per-lane SOURCE address, wave-linear LDS destination) for 4 A chunks and
// BN/32 B chunks per step into the buffer that is not being read; nothing
// passes through VGPRs and there is no ds_write.  Taps that fall into the
// zero padding, rows past M/N and the K tail read a 16-byte zero word instead.
// An LDS row is eight 16-byte slots; chunk q of row r lives in slot
// q ^ ((r >> 1) & 7): the DMA image is linear, the permutation is applied to
// the SOURCE chunk each thread fetches and again when fragments are read, so
// the ds_read_b128 fragment reads of 16 different rows are conflict-free.
//
//   bf16: v_mfma_f32_32x32x16_bf16 — lane (r = l & 31, h = l >> 5) feeds the
//         8 bf16 of chunk 2*ks + h, ks = 0..3.
//   fp32: v_mfma_f32_32x32x2_f32 x 4 per chunk — the 4 floats of a chunk are
//         four K=2 steps; the k order inside a tile is permuted identically
//         for A and B, which only reorders an exact-fmaf sum.
//
// bf16 results leave through LDS (the tile is re-read row-major and stored as
// 16-byte runs); fp32 results are stored from the accumulators (lanes 0..31 of
// one register are 32 consecutive channels = 128 bytes).
// With SPATIAL = false (1x1 kernels without padding, linears, GEMMs) the
// per-tap bounds tests compile away.
#include "avs_internal.h"
#include "igemm_params.h"
#include <type_traits>

__device__ __attribute__((aligned(16))) unsigned int avs_zero16[4];

// Ablation switches exist in the study build only (make study -> libavsum_hip_study.so); the shipped kernels carry none.
#ifdef AVS_STUDY
#define AVS_DEBUG_BIT(p, bit) ((p).debug & (bit))
#else
#define AVS_DEBUG_BIT(p, bit) 0
#endif

#define AVS_GLDS16(src, dst)                                                                        \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),            \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// AVS_F32_SPLIT: fp32 operands, arithmetic on the bf16 matrix cores.  x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi),
// |r| <= 2^-17 |x|; a*b ~ ah*bh + ah*bl + al*bh (each product exact in the fp32 accumulator), i.e. three
// v_mfma_f32_32x32x16_bf16 per 16 reduction elements instead of eight v_mfma_f32_32x32x2_f32: 5.3x the matrix rate
// at a relative error of ~2^-15 per product (bf16 alone: 2^-8).  The split is 3 VALU operations per operand element,
// amortised over the tile's other dimension.
__device__ __forceinline__ void avs_split_bf16(const float4& p0, const float4& p1, bf16x8& hi, bf16x8& lo) {
  const float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
  unsigned h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned short h0 = avs_f32_to_bf16(v[2 * j]), h1 = avs_f32_to_bf16(v[2 * j + 1]);
    const unsigned short l0 = avs_f32_to_bf16(v[2 * j] - avs_bf16_to_f32(h0));
    const unsigned short l1 = avs_f32_to_bf16(v[2 * j + 1] - avs_bf16_to_f32(h1));
    h[j] = (unsigned)h0 | ((unsigned)h1 << 16);
    l[j] = (unsigned)l0 | ((unsigned)l1 << 16);
  }
  hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

// ROWB = bytes of reduction per LDS row and step: 128 (fewer barriers per MAC; 64 KB of LDS, 2 workgroups per
// CU) for the long reductions, 64 (32 KB, 3 workgroups per CU: more DMA in flight) for the short ones, whose
// cost is the latency of their few loads and their stores rather than the matrix work.
// EPI: the epilogue is what a tile of a short reduction costs, so its common forms are compiled separately:
//   EPI_PLAIN  alpha == 1, no bias, no activation, no statistics (ResNet convolutions, projections)
//   EPI_STATS  EPI_PLAIN + fused BatchNorm batch statistics: per-tile partial column sums in the tile's own slots
//              (no atomics), folded in tile order by bn_fold_kernel: deterministic
//   EPI_ANY    everything decided at run time (bias per column / row, ReLU, alpha)
//   EPI_BRELU  alpha == 1, bias per column, ReLU (the folded-BatchNorm convolutions of Inception-v3; Linear+ReLU)
//   EPI_BNLOCAL bf16, 256-row tiles only: the whole BatchNorm in the epilogue WITHOUT any traffic between workgroups.
//              A tile holds floor(256 / rows_per_group) WHOLE groups (tile pitch = that many rows, the rest of the
//              256 rows idle), so a group's statistics are sums over this tile's accumulators alone: per-wave masked
//              column sums -> LDS across the four row-waves -> scale/shift table -> normalise (+ residual, + ReLU)
//              -> store.  No atomics, no waiting, deterministic.  Taken when groups are at most 256 rows and fill
//              at least 3/4 of the tile (per-frame 14x14 and 7x7 maps: 196 = 77 %, 5 x 49 = 96 %).
// (A form in which tiles of larger groups exchanged statistics through float atomics and waited for each other inside
//  one launch was built in round 1 and removed: results were not reproducible run to run and it only won for groups of
//  two or three tiles, which the tile-local form covers.)
//   EPI_AFFINE AVS_F16X2, 1x1 convolutions: a folded BatchNorm affine GIVEN per group of rows (computed beforehand from
//              the input's Gram matrix, avs_bn_gram_affine_f16x2) + residual + ReLU: the one streaming pass of the
//              expanding 1x1 layers whose groups are too large for a tile.  256-row tiles; 128-row tiles (three
//              workgroups per CU) for wide outputs of reductions up to 128 channels

// PIPE: three operand buffers, the DMA of step s+2 is issued in step s; fragment reads are inline-asm
// ds_read_b128 and the waits are hand-counted (s_waitcnt vmcnt(N) + raw s_barrier), because hipcc orders every
// LDS read it can see behind ALL outstanding LDS-DMA (vmcnt(0)), which caps a plain-HIP loop at one step of
// prefetch.  Order per step: wait for this step's DMA -> barrier -> issue step s+2 -> read fragments -> MFMA.
// WR: rows of waves.  2 = the 128 x BN tile (2 x 2 waves of 64 x BN/2); 4 = a 256 x BN tile (4 x 1 waves of 64 x BN):
// twice the matrix work per barrier and 0.75 (BN = 128) instead of 1 fragment read per MFMA, for layers with many
// rows whose cost is the loop itself (the N = 64 layers run 4 MFMAs per wave between barriers at WR = 2).
// FASTK: cin is a multiple of one reduction step (and there are at most 32 taps), so a step never straddles a tap:
// the tap walk (kh, kw, ci) is the same for every thread - kept in scalar registers and advanced by additions -
// a row's padding test is one bit of a per-row tap mask built once, and a DMA source is base + scalar offset.
// (Measured on the 3x3 layers: the general staging code issues ~180 vector + scalar instructions per 16 MFMAs and
// the loop ran at 45 % matrix-core occupancy for that reason alone - with no staging at all it reaches 1.5 PFLOP/s.)
// SPLIT: 0 = operands as stored; 1 = AVS_F32_SPLIT (fp32 operands split into bf16 hi + lo in the loop);
// 2 = AVS_F16X2 (operands ARE stored as fp16 hi + lo runs: the 16-byte chunks of a step alternate hi8 | lo8, so the
// fragments need no arithmetic at all - three v_mfma_f32_32x32x16_f16 per 16 reduction elements - and the epilogue
// splits each output once, where it is produced, instead of every consumer splitting it per tile and step)
// TAP9 (AVS_F16X2, 3x3 / stride 1 / pad 1 on a dense NHWC input, 256-row tiles): for output pixel (y, x) and tap (dy, dx)
// the input pixel is row m + (dy - 1) W + (dx - 1) of the SAME flat pixel array the outputs are rows of.  So a 16-channel
// block of the tile's pixels (+ W + 1 rows of halo on either side: 384 buffer rows) is fetched ONCE - two buffers, a
// block ahead - and the nine taps of the block are nine reduction steps that read their A fragments from it at shifted
// rows; a tap that leaves the frame reads a row of zeros (per-lane tap masks; this also covers tiles that straddle
// frames).  Only the weights of a (block, tap) are fetched per step (a ring of three).  L2 -> LDS traffic of the A
// operand / 9: for the 64-column layers, which sit at the L2 -> LDS intake ceiling (~21 B / clk / CU) with the matrix
// cores 37 % busy.  The reduction runs block-major / tap-minor: another (fixed) summation order than the tap-major walk.
// AP8 (AVS_F16X2 convolution + statistics, 1x1 on dense rows, 256-row tiles: conv1 of the ResNet bottlenecks in layers
// 1-2): the input is an AVS_F16P8 tensor (fp16 hi + 8-bit remainder, 48 bytes per 16 channels).  The LDS-DMA staging is
// the AVS_F16X2 one with other source offsets: of the four 16-byte slots of an LDS row, slot "hi 0-7" takes the block's
// first 16 bytes, slot "hi 8-15" its second, slot "lo 0-7" the 16 remainder bytes and slot "lo 8-15" nothing (an
// out-of-range lane), and a step advances the source by 48 instead of 64 bytes: 3 instead of 4 bytes per input value from
// HBM, the same LDS image for the hi fragments.  A lane reads its 8 remainder bytes with one ds_read_b64 and rebuilds the
// fp16 lo fragment in registers (5 VALU operations per value, behind the hi*hi MFMAs of the step).  (A first version
// fetched the A fragments straight into registers - a wave owns its 64 rows for all columns on these tiles, nothing is
// shared - with buffer_load_dwordx4 + dwordx2 per lane: 19 - 25 % SLOWER than the AVS_F16X2 path, 16-byte pieces of 32
// rows per instruction; removed.)
template <int ES, int BN, bool ACC64, bool SPATIAL, int ROWB, int EPI, bool PIPE, int WR = 2, bool FASTK = false,
          int SPLIT = 0, bool TAP9 = false, bool AP8 = false>
__global__ __launch_bounds__(256, (ROWB == 64 && !ACC64 && WR == 2) ? ((BN == 64 && ES == 2) ? 4 : 3) : 2) void igemm_kernel(
    IgemmParams p) {
  static_assert(!AP8 || (SPLIT == 2 && ES == 4 && PIPE && ROWB == 64 && FASTK && !SPATIAL && !TAP9 && EPI == EPI_STATS),
                "AVS_F16P8 input: the AVS_F16X2 1x1 convolution + statistics on the pipelined tiles");
  static_assert(!TAP9 || (((SPLIT == 2 && ES == 4) || (SPLIT == 0 && ES == 2)) && PIPE && WR == 4 && ROWB == 64 && FASTK &&
                          SPATIAL && (EPI == EPI_STATS || EPI == EPI_BNLOCAL || (EPI == EPI_BRELU && SPLIT == 2))),
                "the shifted-row form: AVS_F16X2 / bf16 convolution + statistics / tile-local BatchNorm (3x3), AVS_F16X2 bias + "
                "ReLU (any stride-1 'same' filter) on the pipelined 256-row tiles");
  static_assert(SPLIT == 0 || (ES == 4 && !ACC64), "the split arithmetic is for 4-byte operands");
  static_assert(WR == 2 || (WR == 4 && !ACC64 && (ES == 2 || SPLIT != 0)),
                "256-row tiles are built for the bf16 variants and the fp32-split arithmetic");
  static_assert(!PIPE || (ROWB == 64 && !ACC64), "the 3-buffer pipeline is built for the 64-byte-row variants");
  static_assert(!ACC64 || (ES == 4 && BN == 64), "fp64 slice accumulation: fp32 operands, narrow tile only");
  static_assert(ROWB == 64 || ROWB == 128, "row bytes");
  constexpr int CE = 16 / ES;        // elements per 16-byte chunk
  constexpr int BKE = ROWB / ES;     // elements per LDS row (one reduction step)
  constexpr int CPRR = ROWB / 16;    // 16-byte slots per LDS row
  constexpr int RPP = 256 / CPRR;    // rows staged per pass of the 256 threads
  constexpr int SH = ROWB == 128 ? 1 : 2;  // rows sharing one 256-byte bank row = 1 << SH
  constexpr int KS = ROWB / 32;      // MFMA sub-steps per row (two chunks each)
  constexpr int WC = 4 / WR;         // columns of waves
  constexpr int WCOLS = BN / WC;     // output columns per wave
  constexpr int A_ROWS = 64 * WR;
  constexpr int NA = A_ROWS / RPP;   // A rows staged per thread
  constexpr int NB = (BN + RPP - 1) / RPP;   // B rows staged per thread (BN = 96: two passes, the second one half empty)
  constexpr int NT = WCOLS / 32;     // 32-wide column tiles per wave
  constexpr int BUF = (A_ROWS + NB * RPP) * CPRR;  // uint4 slots per buffer
  // BN = 96: the column tile of the bias + ReLU convolutions whose cout is 96, 160, 192, 288 ... (Inception-v3): a 128-wide
  // tile spends 25 - 37 % of such a layer's matrix work on columns that do not exist
  static_assert(BN == 64 || BN == 128 || (BN == 96 && SPLIT == 2 && WR == 4 && PIPE && ROWB == 64 && EPI == EPI_BRELU),
                "96-column tiles: the AVS_F16X2 bias + ReLU form on the pipelined 256-row tiles");
  static_assert(WCOLS % 32 == 0, "a wave's columns are whole 32-wide MFMA tiles");
  constexpr int CT_PITCH = BN * 2 + 16;      // bf16 epilogue staging tile: row pitch in bytes (16 bytes of padding)
  constexpr int CT_SLOTS = ES == 2 ? (A_ROWS * CT_PITCH) / 16 : 0;
  constexpr int NBUF = PIPE ? 3 : 2;
  // behind the staging tile: EPI_BNLOCAL's scale/shift table; EPI_STATS' per-wave column sums [WR][2 groups][2][BN]
  constexpr int TAB_SLOTS = EPI == EPI_BNLOCAL ? (BNLOCAL_MAX_GROUPS * 2 * BN * 4) / 16
                            : EPI == EPI_STATS ? (WR * 2 * 2 * BN * 4) / 16
                                               : 0;
  // AVS_F16X2 epilogue: every wave stages 32 of its rows at a time as fp32, row-major (pitch WCOLS + 8 words), and
  // reads them back as runs of 8 columns; the statistics tables alias those regions (a barrier separates the uses)
  constexpr int H2_P = WCOLS + 8;                               // words per staged row
  constexpr int H2_SLOTS = SPLIT == 2 ? (4 * 32 * H2_P * 4) / 16 : 0;
  // TAP9: two A buffers of 384 rows (+ 4 rows of zeros) and a ring of three weight tiles
  constexpr int T9_AROWS = 384;
  constexpr int T9_ABUF = (T9_AROWS + 4) * CPRR;
  constexpr int T9_BBUF = NB * RPP * CPRR;
  constexpr int OPERAND_SLOTS = TAP9 ? 2 * T9_ABUF + 3 * T9_BBUF : NBUF * BUF;
  constexpr int LDS_BASE = OPERAND_SLOTS > CT_SLOTS + TAB_SLOTS ? OPERAND_SLOTS : CT_SLOTS + TAB_SLOTS;
  constexpr int LDS_SLOTS = LDS_BASE > H2_SLOTS ? LDS_BASE : H2_SLOTS;

  __shared__ uint4 lds[LDS_SLOTS];

  // XCD-aware, bijective block remap: blocks that share an XCD (orig % 8)
  // take consecutive tiles, so neighbouring column tiles reuse A rows in L2.
  const unsigned nwg = gridDim.x, orig = blockIdx.x;
  const unsigned q = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
  const unsigned wg = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
  const int tn = wg % p.tiles_n;
  const int tm = wg / p.tiles_n;
  const int m0 = EPI == EPI_BNLOCAL ? tm * p.tile_rows : tm * A_ROWS;
  const int n0 = tn * BN;

  const long long z = blockIdx.z;
  const char* __restrict__ x = p.x + z * p.sA * ES;
  const char* __restrict__ w = p.w + z * p.sB * ES;
  char* __restrict__ y = p.y + z * p.sC * ES;
  const float* __restrict__ bias = p.bias ? p.bias + z * p.sBias : nullptr;
  const char* zsrc = reinterpret_cast<const char*>(avs_zero16);

  const int t = threadIdx.x;
  // the wave index is uniform: say so, and every address built from it (LDS-DMA destinations, tile quadrants)
  // stays in scalar registers
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int c = t & (CPRR - 1);
  const int rb = t / CPRR;                          // 0..RPP-1
  const int cq = c ^ ((rb >> SH) & (CPRR - 1));     // the k-chunk this thread fetches into slot c

  // ---- per-thread row bases (rows rb + 32*i of the A tile, and of the B tile) ----
  const char* a_base[NA];
  int hi0[NA], wi0[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int m = m0 + rb + RPP * i;
    if constexpr (EPI == EPI_BNLOCAL) {
      if (rb + RPP * i >= p.tile_rows) m = p.M;  // the tile's idle rows
    }
    if (m < p.M && p.lin_stride >= 0) {
      hi0[i] = 0;
      wi0[i] = 0;
      a_base[i] = x + (long long)m * p.lin_stride * ES;
    } else if (m < p.M) {
      const int n = m / p.HoWo;
      const int rem = m - n * p.HoWo;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      hi0[i] = ho * p.sh - p.ph;
      wi0[i] = wo * p.sw - p.pw;
      a_base[i] = x + ((long long)n * p.x_img_stride + (long long)hi0[i] * p.x_row_stride +
                       (long long)wi0[i] * p.x_px_stride) * ES;
    } else {
      hi0[i] = -(1 << 29);
      wi0[i] = 0;
      a_base[i] = nullptr;  // never dereferenced: marks the row invalid
    }
  }
  const char* b_base[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int n = n0 + rb + RPP * i;
    b_base[i] = (n < p.N && rb + RPP * i < BN) ? w + (long long)n * p.ldb * ES : nullptr;
  }

  if (AVS_DEBUG_BIT(p, 2)) {
#pragma unroll
    for (int i = 0; i < NA; ++i) a_base[i] = nullptr;
#pragma unroll
    for (int i = 0; i < NB; ++i) b_base[i] = nullptr;
  }
  int kc = cq * CE;
  int kk = kc / p.cin;
  int ci = kc - kk * p.cin;
  int kh = kk / p.KW;
  int kw = kk - kh * p.KW;

  // ---- FASTK state: per-row tap masks and 32-bit buffer offsets (vector), the tap walk (scalar) ----
  // Operands are fetched with buffer_load_dwordx4 ... lds: address = resource base (SGPRs) + per-lane offset (one
  // VGPR, fixed for the whole reduction) + scalar offset (the tap walk / the k step).  A lane whose tap falls into
  // the padding (or whose row does not exist) presents an offset beyond num_records and the hardware returns
  // zeros, so staging costs no vector arithmetic per step beyond that select.  The A base is moved back by the
  // padding so that every row's own offset is non-negative; the launcher guarantees a tile's rows and the tap walk
  // stay inside the 2 GiB window.
  constexpr unsigned BUF_OOB = 0x80000000u;
  unsigned amask[FASTK ? NA : 1];
  unsigned aoff[FASTK ? NA : 1], boff[FASTK ? NB : 1];
  int f_tap = 0, f_ci0 = 0, f_kw = 0;
  int f_koff = 0;   // bytes: the current tap + channel block relative to a row's first tap
  int f_kb = 0;     // bytes: the current k step inside a B row
  __amdgpu_buffer_rsrc_t a_rsrc, b_rsrc;
  if constexpr (FASTK) {
    const int taps = p.K / p.cin;
    // uniform tile origin: the first image the tile touches (or its first row on the linear-row path)
    long long a_origin;
    int n_first = 0;
    if (p.lin_stride >= 0) {
      // study build, bit 6 (1x1 layers, 64-byte steps): the INPUT is read reduction-step major too, x[k / 32][M][32] - what
      // a channel-blocked activation layout would give the A operand (whole cache lines per DMA instruction)
      a_origin = AVS_DEBUG_BIT(p, 64) ? (long long)m0 * BKE : (long long)m0 * p.lin_stride;
    } else {
      n_first = m0 / p.HoWo;
      a_origin = (long long)n_first * p.x_img_stride - (long long)p.ph * p.x_row_stride - (long long)p.pw * p.x_px_stride;
    }
    a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x) + a_origin * (AP8 ? 3 : ES), 0, (int)BUF_OOB, 0x00020000);
    b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(w) + (p.w_kstep ? (long long)n0 * 64 : (long long)n0 * p.ldb * ES), 0, (int)BUF_OOB, 0x00020000);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      unsigned mk = 0, off = BUF_OOB;
      int m = m0 + rb + RPP * i;
      if constexpr (EPI == EPI_BNLOCAL) {
        if (rb + RPP * i >= p.tile_rows) m = p.M;
      }
      if (a_base[i] != nullptr && m < p.M) {
        if (AP8) {
          // AVS_F16P8 rows: slot "hi 0-7" <- bytes 0-15 of the 48-byte block, "lo 0-7" <- the remainder bytes (32-47),
          // "hi 8-15" <- bytes 16-31, "lo 8-15" <- nothing
          mk = ~0u;
          off = cq == 3 ? BUF_OOB : (unsigned)((long long)(m - m0) * p.lin_stride * 3) + (cq == 0 ? 0u : cq == 1 ? 32u : 16u);
        } else if (p.lin_stride >= 0) {
          mk = ~0u;
          off = AVS_DEBUG_BIT(p, 64) ? (unsigned)((m - m0) * ROWB) + cq * 16
                                     : (unsigned)((long long)(m - m0) * p.lin_stride * ES) + cq * 16;
        } else {
          const int n = m / p.HoWo;
          if constexpr (SPATIAL) {
            for (int tp = 0; tp < taps; ++tp) {
              const int th = tp / p.KW, tw = tp - th * p.KW;
              if ((unsigned)(hi0[i] + th) < (unsigned)p.H && (unsigned)(wi0[i] + tw) < (unsigned)p.W) mk |= 1u << tp;
            }
          } else {
            mk = ~0u;
          }
          off = (unsigned)(((long long)(n - n_first) * p.x_img_stride + (long long)(hi0[i] + p.ph) * p.x_row_stride +
                            (long long)(wi0[i] + p.pw) * p.x_px_stride) * ES) + cq * 16;
        }
      }
      amask[i] = mk;
      aoff[i] = off;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      boff[i] = b_base[i] == nullptr ? BUF_OOB
                : p.w_kstep        ? (unsigned)((rb + RPP * i) * 64) + (unsigned)(cq >> 2) * (unsigned)p.N * 64u + (cq & 3) * 16
                                   : (unsigned)((long long)(rb + RPP * i) * p.ldb * ES) + cq * 16;
  }

  auto stage = [&](int buf) {
    if (AVS_DEBUG_BIT(p, 8)) return;  // study build: no staging at all (the loop runs on whatever the LDS holds)
    uint4* abuf = lds + buf * BUF + wave * 64;        // wave-uniform; lane l lands at +l
    uint4* bbuf = abuf + A_ROWS * CPRR;
    if constexpr (FASTK) {
      if (!AVS_DEBUG_BIT(p, 16)) {  // study build: bit 4 skips the A operand, bit 5 the B operand
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          unsigned off = aoff[i];
          if constexpr (SPATIAL) off = ((amask[i] >> f_tap) & 1u) ? off : BUF_OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(abuf + 256 * i), 16,
                                                   (int)off, f_koff, 0, 0);
        }
      }
      if (!AVS_DEBUG_BIT(p, 32)) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (__attribute__((address_space(3))) void*)(bbuf + 256 * i), 16,
                                                   (int)boff[i], f_kb, 0, 0);
      }
      f_kb += p.w_kstep ? p.N * ROWB : BKE * ES;   // (ROWB / 64 pieces of N x 64 bytes per step)
      // next step (scalar): the same tap's next channel block, or the next tap
      f_ci0 += BKE;
      f_koff += AP8 ? 48 : (AVS_DEBUG_BIT(p, 64) ? p.M * ROWB : BKE * ES);
      if (f_ci0 == p.cin) {
        f_ci0 = 0;
        ++f_tap;
        f_koff += (int)((p.x_px_stride - p.cin) * ES);
        if (++f_kw == p.KW) {
          f_kw = 0;
          f_koff += (int)((p.x_row_stride - (long long)p.KW * p.x_px_stride) * ES);
        }
      }
      return;
    }
    const bool kval = kc < p.K;
    const long long koff = ((long long)kh * p.x_row_stride + (long long)kw * p.x_px_stride + ci) * ES;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      bool ok = kval && a_base[i] != nullptr;
      if constexpr (SPATIAL)
        ok = ok && (unsigned)(hi0[i] + kh) < (unsigned)p.H && (unsigned)(wi0[i] + kw) < (unsigned)p.W;
      // branch-free select: exactly one DMA instruction per chunk (the counted waits rely on it)
      const unsigned long long am = ok ? ~0ull : 0ull;
      const char* src = reinterpret_cast<const char*>(
          (reinterpret_cast<unsigned long long>(a_base[i] + koff) & am) |
          (reinterpret_cast<unsigned long long>(zsrc) & ~am));
      AVS_GLDS16(src, abuf + 256 * i);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const unsigned long long bm = (kval && b_base[i] != nullptr) ? ~0ull : 0ull;
      // AVS_W_KSTEP32: element (n, k) lives at ((k / S) * N + n) * S + k % S, S = the elements of a 64-byte step
      constexpr int KSE = 64 / ES;
      const char* bsrc = p.w_kstep ? w + (((long long)(kc / KSE) * p.N + (n0 + rb + RPP * i)) * KSE + (kc % KSE)) * ES
                                   : b_base[i] + (long long)kc * ES;
      const char* src = reinterpret_cast<const char*>((reinterpret_cast<unsigned long long>(bsrc) & bm) |
                                                      (reinterpret_cast<unsigned long long>(zsrc) & ~bm));
      AVS_GLDS16(src, bbuf + 256 * i);
    }
    kc += BKE;
    ci += BKE;
    while (ci >= p.cin) {
      ci -= p.cin;
      if (++kw == p.KW) {
        kw = 0;
        ++kh;
      }
    }
  };

  const int wr = wave / WC, wc = wave % WC;
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ACC64: every reduction step (32 fp32 elements) is an exact-fmaf MFMA chain in fp32; the steps are summed
  // in fp64, so the rounding error scales with the size of a STEP sum, not of the whole running sum.
  double acc64[ACC64 ? 2 : 1][ACC64 ? NT : 1][ACC64 ? 16 : 1];
  if constexpr (ACC64) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc64[i][j][e] = 0.0;
  }

  const int steps = (p.K + BKE - 1) / BKE;
  if constexpr (TAP9) {
    constexpr int T9_NA = T9_AROWS / RPP;   // DMA instructions per wave and A block (6)
    // TGEN (bias + ReLU: Inception-v3's 1x7 / 7x1 / 3x3 / 1x3 / 3x1 stride-1 "same" layers): any KH x KW with the filter's
    // reach ph W + pw <= 64 rows, the input possibly a channel slice of a wider tensor (pixel stride > cin).  The 3x3 forms
    // of the BatchNorm epilogues keep their constants (the same code as before).
    constexpr bool TGEN = EPI == EPI_BRELU;
    const int ntap = TGEN ? p.K / p.cin : 9;
    const int tkw = TGEN ? p.KW : 3;
    const int tph = TGEN ? p.ph : 1, tpw = TGEN ? p.pw : 1;
    const long long pxs = TGEN ? p.x_px_stride : (long long)p.cin;   // elements between consecutive pixels
    const unsigned lds_base = (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)lds);
    const int w1 = TGEN ? tph * p.W + tpw : p.W + 1;
    // buffer row a holds flat input row m0 - w1 + a; the buffer window starts at the first row fetched
    const long long mbase = m0 > w1 ? m0 - w1 : 0;
    const __amdgpu_buffer_rsrc_t a9 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x) + mbase * pxs * ES, 0,
                                                                         (int)BUF_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t b9 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(w) + (p.w_kstep ? (long long)n0 * 64 : (long long)n0 * p.ldb * ES), 0, (int)BUF_OOB, 0x00020000);
    unsigned aoff9[T9_NA], boff9[NB];
#pragma unroll
    for (int i = 0; i < T9_NA; ++i) {
      const long long m = (long long)m0 - w1 + rb + RPP * i;
      aoff9[i] = (m >= 0 && m < p.M) ? (unsigned)((m - mbase) * pxs * ES) + cq * 16 : BUF_OOB;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      boff9[i] = (n0 + rb + RPP * i >= p.N || rb + RPP * i >= BN) ? BUF_OOB
                 : p.w_kstep                                      ? (unsigned)((rb + RPP * i) * 64) + (cq & 3) * 16
                                                                  : (unsigned)((long long)(rb + RPP * i) * p.ldb * ES) + cq * 16;
    // bit tp of vm[mt]: tap tp of this lane's row of block mt lies inside its frame
    unsigned vm[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      unsigned mk = 0;
      const int m = m0 + wr * 64 + mt * 32 + lr;
      bool live = m < p.M;
      if constexpr (EPI == EPI_BNLOCAL) live = live && wr * 64 + mt * 32 + lr < p.tile_rows;   // the tile's idle rows
      if (live) {
        const int pix = m % p.HoWo;
        const int yy = pix / p.W, xx = pix - yy * p.W;
        if constexpr (TGEN) {
          for (int tp = 0; tp < ntap; ++tp)
            if ((unsigned)(yy + tp / tkw - tph) < (unsigned)p.H && (unsigned)(xx + tp % tkw - tpw) < (unsigned)p.W) mk |= 1u << tp;
        } else {
#pragma unroll
          for (int tp = 0; tp < 9; ++tp)
            if ((unsigned)(yy + tp / 3 - 1) < (unsigned)p.H && (unsigned)(xx + tp % 3 - 1) < (unsigned)p.W) mk |= 1u << tp;
        }
      }
      vm[mt] = mk;
    }
    if (t < 32) lds[(t >> 4) * T9_ABUF + T9_AROWS * CPRR + (t & 15)] = make_uint4(0u, 0u, 0u, 0u);   // the rows of zeros
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned fb9[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
        const int brow = nt * 32 + lr;
        // sub-step hl: AVS_F16X2 = the hi / lo chunk of this lane half's 8 elements; bf16 = its chunk of 16-element half hl
        const int chunk = SPLIT == 2 ? 2 * lh + hl : 2 * hl + lh;
        fb9[nt][hl] = (unsigned)(brow * CPRR + (chunk ^ ((brow >> SH) & (CPRR - 1)))) * 16u;
      }
    const int nblk = p.cin / BKE;   // 16-channel blocks; step s = (block s / 9, tap s % 9)
    auto stage_a = [&](int blk, int buf) {
      uint4* abuf = lds + buf * T9_ABUF + wave * 64;
#pragma unroll
      for (int i = 0; i < T9_NA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a9, (__attribute__((address_space(3))) void*)(abuf + 256 * i), 16,
                                                 (int)aoff9[i], blk * ROWB, 0, 0);
    };
    int sb_blk = 0, sb_tap = 0, sb_slot = 0;   // the step the next stage_b call fetches, and its ring slot
    auto stage_b = [&]() {
      uint4* bbuf = lds + 2 * T9_ABUF + sb_slot * T9_BBUF + wave * 64;
      const int kstep = sb_tap * nblk + sb_blk;   // reduction elements tap * cin + 16 * block .. + 15 of a filter row
      const int kb = p.w_kstep ? kstep * (p.N * ROWB) : kstep * ROWB;
#pragma unroll
      for (int i = 0; i < NB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b9, (__attribute__((address_space(3))) void*)(bbuf + 256 * i), 16,
                                                 (int)boff9[i], kb, 0, 0);
      sb_slot = sb_slot == 2 ? 0 : sb_slot + 1;
      if (++sb_tap == ntap) {
        sb_tap = 0;
        ++sb_blk;
      }
    };
    stage_a(0, 0);
    stage_b();
    if (steps > 1) stage_b();
    const int abase = wr * 64 + lr + w1;   // buffer row of this lane's row of block 0 under the centre tap
    int blk = 0, tap = 0, tdx = 0, slot = 0, sig = -w1;   // sig: the tap's row shift (dy - 1) W + (dx - 1)
    for (int s = 0; s < steps; ++s) {
      // loads return in order: this step's weights (issued two steps ago, AFTER any A block of that step) cover the A
      // block; younger than them: the next step's weights and - when the step before was a block's first tap - an A block
      if (s + 1 < steps) {
        if (tap == 1 && blk + 1 < nblk)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB + T9_NA) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      const unsigned abuf = lds_base + (unsigned)(blk & 1) * (T9_ABUF * 16u);
      const unsigned bbase = lds_base + (unsigned)(2 * T9_ABUF + slot * T9_BBUF) * 16u;
      uint4 fa[2][2], fb[2][NT];
      unsigned aaddr[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int a = ((vm[mt] >> tap) & 1u) ? abase + mt * 32 + sig : T9_AROWS;   // outside the frame: the zero row
        aaddr[mt] = abuf + (unsigned)a * ROWB + (unsigned)((SPLIT == 2 ? 2 * lh : lh) ^ ((a >> SH) & (CPRR - 1))) * 16u;
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) fa[0][mt] = avs_lds_read_b128(aaddr[mt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[0][nt] = avs_lds_read_b128(bbase + fb9[nt][0]);
#pragma unroll
      // (the second sub-step's chunk index differs in one bit: the lo chunk beside the hi one / the other 32-byte half)
      for (int mt = 0; mt < 2; ++mt) fa[1][mt] = avs_lds_read_b128(aaddr[mt] ^ (SPLIT == 2 ? 16u : 32u));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[1][nt] = avs_lds_read_b128(bbase + fb9[nt][1]);
      // the next A block goes out at a block's first tap (its buffer was last read in the step before: every wave is
      // past this step's barrier), then the weights of step s + 2
      if (tap == 0 && blk + 1 < nblk) stage_a(blk + 1, (blk + 1) & 1);
      if (s + 2 < steps) stage_b();
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 + NT) : "memory");
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) avs_pin(fa[0][mt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) avs_pin(fb[0][nt]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          if constexpr (SPLIT == 2)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(avs_f16x8, fa[0][mt]),
                                                                 __builtin_bit_cast(avs_f16x8, fb[0][nt]), acc[mt][nt], 0, 0, 0);
          else
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[0][mt]),
                                                                  __builtin_bit_cast(bf16x8, fb[0][nt]), acc[mt][nt], 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) avs_pin(fa[1][mt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) avs_pin(fb[1][nt]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (SPLIT == 2) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(avs_f16x8, fa[1][mt]),
                                                                 __builtin_bit_cast(avs_f16x8, fb[0][nt]), acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(avs_f16x8, fa[0][mt]),
                                                                 __builtin_bit_cast(avs_f16x8, fb[1][nt]), acc[mt][nt], 0, 0, 0);
          } else {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[1][mt]),
                                                                  __builtin_bit_cast(bf16x8, fb[1][nt]), acc[mt][nt], 0, 0, 0);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
      slot = slot == 2 ? 0 : slot + 1;
      ++sig;   // next tap: one pixel to the right, or the next row's first
      if (++tdx == tkw) {
        tdx = 0;
        sig += p.W - tkw;
      }
      if (++tap == ntap) {
        tap = 0;
        ++blk;
        sig = -w1;
      }
    }
    __syncthreads();   // the epilogue's tables and staging regions alias the operand buffers
  } else if constexpr (PIPE) {
    constexpr int NDMA = NA + NB;  // DMA instructions one stage() issues per wave
    const unsigned lds_base = (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)lds);
    // per-lane fragment byte offsets inside a buffer (ks = 0; ks = 1 flips bit 1 of the chunk)
    unsigned fa_off[2][KS], fb_off[NT][KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // AVS_F16X2: the chunks of a row alternate hi8 | lo8; sub-step pair (ks, ks + 1) = (hi, lo) of the 8 elements
      // this lane half feeds
      const int chunk = SPLIT == 2 ? 2 * (ks & ~1) + 2 * lh + (ks & 1) : 2 * ks + lh;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = wr * 64 + mt * 32 + lr;
        fa_off[mt][ks] = (unsigned)(row * CPRR + (chunk ^ ((row >> SH) & (CPRR - 1)))) * 16u;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int row = wc * WCOLS + nt * 32 + lr;
        fb_off[nt][ks] = (unsigned)((A_ROWS + row) * CPRR + (chunk ^ ((row >> SH) & (CPRR - 1)))) * 16u;
      }
    }
    stage(0);
    if (steps > 1) stage(1);
    int cur = 0;  // buffer holding step s
    for (int s = 0; s < steps; ++s) {
      if (s + 1 < steps)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const unsigned bbase = lds_base + (unsigned)cur * (BUF * 16u);
      uint4 fa[KS][2], fb[KS][NT];
      uint2 frem[AP8 ? 2 : 1];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if constexpr (AP8) {
            // sub-step 1 of an AVS_F16P8 row: this lane half's 8 remainder bytes (the slot "lo 0-7" holds all 16)
            if (ks == 1) {
              const int row = wr * 64 + mt * 32 + lr;
              frem[mt] = avs_lds_read_b64(bbase + (unsigned)(row * CPRR + (1 ^ ((row >> SH) & (CPRR - 1)))) * 16u + 8u * lh);
              continue;
            }
          }
          fa[ks][mt] = avs_lds_read_b128(bbase + fa_off[mt][ks]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fb[ks][nt] = avs_lds_read_b128(bbase + fb_off[nt][ks]);
      }
      // the DMA of step s+2 is issued while the fragment reads are in flight; the MFMAs of the first half-step
      // start as soon as ITS fragments have landed (LDS reads return in order; a scalar load in between can only
      // make the counted wait more conservative)
      if (s + 2 < steps) stage(cur == 0 ? 2 : cur - 1);  // (s+2) % 3 == (cur + 2) % 3
      // The fragment reads above are asynchronous inline asm: after each hand-counted wait the fragments it covers are
      // passed THROUGH an empty asm ("+v"), so that no use of them - not even a register copy the compiler may want -
      // can be scheduled ahead of the wait.  (A copy placed before the wait reads a register whose LDS data has not
      // landed: seen as size-dependent garbage when a runtime branch made the compiler copy the fragments.)
      if constexpr (SPLIT == 2) {
        // operands stored as fp16 hi | lo runs: sub-step kp holds the hi fragments, kp + 1 the lo fragments of the same
        // 16 reduction elements; hi*hi starts as soon as the hi fragments have landed
        static_assert(KS == 2, "the pipelined loop runs on 64-byte rows: one hi / lo pair of sub-steps per step");
        {
          constexpr int kp = 0;
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 + NT) : "memory");
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) avs_pin(fa[kp][mt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) avs_pin(fb[kp][nt]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(avs_f16x8, fa[kp][mt]),
                                                                   __builtin_bit_cast(avs_f16x8, fb[kp][nt]),
                                                                   acc[mt][nt], 0, 0, 0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if constexpr (AP8) {   // the fp16 lo fragments from the hi fragments and the remainder bytes
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              avs_pin2(frem[mt]);
              fa[kp + 1][mt] = avs_f16p8_lo8(fa[kp][mt], frem[mt]);
            }
          } else {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) avs_pin(fa[kp + 1][mt]);
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) avs_pin(fb[kp + 1][nt]);
          if constexpr (!AP8) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(avs_f16x8, fa[kp + 1][mt]),
                                                                   __builtin_bit_cast(avs_f16x8, fb[kp][nt]),
                                                                   acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(avs_f16x8, fa[kp][mt]),
                                                                   __builtin_bit_cast(avs_f16x8, fb[kp + 1][nt]),
                                                                   acc[mt][nt], 0, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
        cur = cur == 2 ? 0 : cur + 1;
        continue;
      }
      if constexpr (SPLIT == 1) {
        // fp32 operands on the bf16 matrix cores: the fragments of two sub-steps (8 reduction elements per lane, the
        // same lane -> k map for A and B) are split into hi / lo and contracted as lo*hi + hi*lo + hi*hi
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) avs_pin(fa[ks][mt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) avs_pin(fb[ks][nt]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kp = 0; kp < KS; kp += 2) {
          bf16x8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            avs_split_bf16(__builtin_bit_cast(float4, fa[kp][mt]), __builtin_bit_cast(float4, fa[kp + 1][mt]), ah[mt], al[mt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            avs_split_bf16(__builtin_bit_cast(float4, fb[kp][nt]), __builtin_bit_cast(float4, fb[kp + 1][nt]), bh[nt], bl[nt]);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        cur = cur == 2 ? 0 : cur + 1;
        continue;
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS)
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((KS - 1) * (2 + NT)) : "memory");
        else
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) avs_pin(fa[ks][mt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) avs_pin(fb[ks][nt]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if constexpr (ES == 2) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  __builtin_bit_cast(bf16x8, fa[ks][mt]), __builtin_bit_cast(bf16x8, fb[ks][nt]),
                  acc[mt][nt], 0, 0, 0);
            } else {
              const float4 a4 = __builtin_bit_cast(float4, fa[ks][mt]);
              const float4 b4 = __builtin_bit_cast(float4, fb[ks][nt]);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[mt][nt], 0, 0, 0);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      cur = cur == 2 ? 0 : cur + 1;
    }
    __syncthreads();  // every wave is done with the operand buffers before the epilogue reuses the LDS
  } else {
    stage(0);
    __syncthreads();  // drains the DMA (vmcnt) and makes every wave's part of the tile visible

    for (int s = 0; s < steps; ++s) {
      const int buf = s & 1;
      const uint4* abuf = lds + buf * BUF;
      const uint4* bbuf = abuf + A_ROWS * CPRR;
      // 1. all fragments of this step into registers (the compiler orders every LDS read behind the
      //    outstanding LDS-DMA, so the reads must come BEFORE the next tile's DMA is issued)
      uint4 fa[KS][2], fb[KS][NT];
  #pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int chunk = SPLIT == 2 ? 2 * (ks & ~1) + 2 * lh + (ks & 1) : 2 * ks + lh;
  #pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int row = wr * 64 + mt * 32 + lr;
          fa[ks][mt] = abuf[row * CPRR + (chunk ^ ((row >> SH) & (CPRR - 1)))];
        }
  #pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int row = wc * WCOLS + nt * 32 + lr;
          fb[ks][nt] = bbuf[row * CPRR + (chunk ^ ((row >> SH) & (CPRR - 1)))];
        }
      }
      // 2. DMA of the next tile into the other buffer: every wave finished reading it before the barrier
      //    that ended the previous step.  It is in flight during the MFMAs below.
      if (s + 1 < steps) stage(buf ^ 1);
      // 3. matrix cores
      if constexpr (SPLIT == 2) {
#pragma unroll
        for (int kp = 0; kp < KS; kp += 2)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const avs_f16x8 ah = __builtin_bit_cast(avs_f16x8, fa[kp][mt]), al = __builtin_bit_cast(avs_f16x8, fa[kp + 1][mt]);
              const avs_f16x8 bh = __builtin_bit_cast(avs_f16x8, fb[kp][nt]), bl = __builtin_bit_cast(avs_f16x8, fb[kp + 1][nt]);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[mt][nt], 0, 0, 0);
            }
      }
      if constexpr (SPLIT == 1) {
#pragma unroll
        for (int kp = 0; kp < KS; kp += 2) {
          bf16x8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            avs_split_bf16(__builtin_bit_cast(float4, fa[kp][mt]), __builtin_bit_cast(float4, fa[kp + 1][mt]), ah[mt], al[mt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            avs_split_bf16(__builtin_bit_cast(float4, fb[kp][nt]), __builtin_bit_cast(float4, fb[kp + 1][nt]), bh[nt], bl[nt]);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
            }
        }
      }
      if constexpr (SPLIT == 0) {
  #pragma unroll
      for (int ks = 0; ks < KS; ++ks)
  #pragma unroll
        for (int mt = 0; mt < 2; ++mt)
  #pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if constexpr (ES == 2) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  __builtin_bit_cast(bf16x8, fa[ks][mt]), __builtin_bit_cast(bf16x8, fb[ks][nt]),
                  acc[mt][nt], 0, 0, 0);
            } else {
              const float4 a4 = __builtin_bit_cast(float4, fa[ks][mt]);
              const float4 b4 = __builtin_bit_cast(float4, fb[ks][nt]);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[mt][nt], 0, 0, 0);
            }
          }
      }
      if constexpr (ACC64) {
  #pragma unroll
        for (int i = 0; i < 2; ++i)
  #pragma unroll
          for (int j = 0; j < NT; ++j)
  #pragma unroll
            for (int e = 0; e < 16; ++e) {
              acc64[i][j][e] += (double)acc[i][j][e];
              acc[i][j][e] = 0.f;
            }
      }
      // keep the MFMAs ahead of the wait: without this the scheduler hoists the barrier (and its vmcnt(0))
      // above them and the DMA latency is no longer covered by the matrix work of this step
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    }
  }

  // ---- epilogue ----
  if constexpr (SPLIT == 2) {
    // AVS_F16X2 output.  Statistics (EPI_STATS / EPI_BNLOCAL) are taken in TWO rounds over the fp32 accumulators: column
    // sums -> the mean of the rows a tile holds of each group -> sums of squares ABOUT that mean, so nothing cancels
    // however large mean^2 / var is (a constant frame, a dead channel); tiles of one group are merged by Chan's update
    // in tile order (bn_fold_kernel).  Deterministic: fixed orders, no atomics.
    // Output: every wave stages 32 of its rows at a time in LDS as fp32 (row-major) and reads them back as runs of 8
    // columns: residual added and ReLU applied in fp32, then ONE split into fp16 hi | lo and two 16-byte stores.
    static_assert(EPI == EPI_PLAIN || EPI == EPI_STATS || EPI == EPI_BNLOCAL || EPI == EPI_AFFINE || EPI == EPI_BRELU,
                  "epilogue forms built for AVS_F16X2");
    static_assert(EPI != EPI_BNLOCAL || WR == 4, "the tile-local BatchNorm form runs on the 256-row tiles");
    float* const fl = reinterpret_cast<float*>(lds);
    // masked sums over this lane's 32 rows of column tile nt: rows with lo <= roff < hi (roff = row inside the wave's
    // 64 rows minus 4 * lh); both lane halves return the column's total over the wave's rows
    auto wave_sum = [&](int nt, int lo, int hi) -> float {
      float s = 0.f;
      if (lo <= 0 && hi >= 64) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int e = 0; e < 16; ++e) s += acc[mt][nt][e];
      } else {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int roff = mt * 32 + (e & 3) + 8 * (e >> 2);
            s += (roff >= lo && roff < hi) ? acc[mt][nt][e] : 0.f;
          }
      }
      return s + __shfl_xor(s, 32, 64);
    };
    auto wave_sq = [&](int nt, int lo, int hi, float mean) -> float {
      float s = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int roff = mt * 32 + (e & 3) + 8 * (e >> 2);
          const float d = acc[mt][nt][e] - mean;
          s = (roff >= lo && roff < hi) ? fmaf(d, d, s) : s;
        }
      return s + __shfl_xor(s, 32, 64);
    };
    // per-lane BatchNorm affines of the (up to three) groups this wave's rows lie in (EPI_BNLOCAL)
    float sc0[NT], sf0[NT], sc1[NT], sf1[NT], sc2[NT], sf2[NT];
    int bnd1 = 1 << 30, bnd2 = 1 << 30;   // first roff of the wave's second / third group
    int used = A_ROWS;                    // EPI_BNLOCAL: rows of this tile that exist
    if constexpr (EPI == EPI_BNLOCAL) used = (m0 + p.tile_rows <= p.M ? p.tile_rows : p.M - m0);
    constexpr int G = WCOLS / 8;            // runs of 8 columns per staged row
    constexpr int RSTEP = 64 / G;           // staged rows between a lane's consecutive runs
    constexpr int NU = (32 + RSTEP - 1) / RSTEP;   // runs per lane and half
    // G = 12 (96-column waves): 5 rows x 12 runs = 60 lanes work, the last round covers rows 30, 31 only
    constexpr bool RAGGED = 64 % G != 0;
    static_assert(!RAGGED || EPI == EPI_BRELU, "ragged run maps: the bias + ReLU form only (no residual rows)");
    constexpr bool NORM = EPI == EPI_BNLOCAL || EPI == EPI_AFFINE || EPI == EPI_BRELU;
    float* const wreg = fl + wave * (32 * H2_P);
    const bool relu = NORM && (EPI == EPI_BRELU || p.act == AVS_ACT_RELU);
    if constexpr (EPI == EPI_BRELU) {   // bias per column + ReLU (the folded-BatchNorm convolutions of Inception-v3)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + wc * WCOLS + nt * 32 + lr;
        sc0[nt] = 1.f;
        sf0[nt] = col < p.N ? bias[col] : 0.f;
        sc1[nt] = sc2[nt] = 1.f;
        sf1[nt] = sf2[nt] = 0.f;
      }
    }
    // a lane's runs: staged row rl0 + it * RSTEP of a half, columns 8 * grp .. + 7
    const int rl0 = lane / G, grp = lane % G;
    const int col8 = n0 + wc * WCOLS + grp * 8;
    const int row_lim = (EPI == EPI_BNLOCAL ? m0 + used : p.M);   // rows at or past this one do not exist
    const bool col_ok = col8 < p.N && (!RAGGED || lane < RSTEP * G);
    // avs_conv2d_nhwc_split: columns >= relu_cols (> 0) skip the ReLU (a stacked head whose activation follows a pooling)
    const bool relu_l = relu && (p.relu_cols <= 0 || col8 < p.relu_cols);
    // AVS_F16P8 (output / residual of the given-affine form): byte offset of this lane's run of 8 columns inside a row -
    // 48 bytes per 16 columns: hi halves of columns 0-7 | of columns 8-15 | the 16 remainder bytes - and from its hi
    // halves to its remainder bytes
    const int p8_hi = (col8 >> 4) * 48 + ((col8 >> 3) & 1) * 16;
    const int p8_rem = 32 - ((col8 >> 3) & 1) * 8;
    // the residual rows of a half are in flight before that half is staged: one memory latency per half instead of one
    // per run.  Rows / columns that do not exist read the residual's first run (never used): no divergent branch
    // around the loads (and hipcc 7.2 crashes in machine copy propagation on a zero-filling else branch).
    uint4 rhi[2][NORM ? NU : 1], rlo[2][NORM ? NU : 1];
    auto res_fetch = [&](auto half) {
      constexpr int mt = decltype(half)::value;
      if constexpr (NORM) {
        const long long row0 = (long long)m0 + wr * 64 + mt * 32 + rl0;
        if (EPI == EPI_AFFINE && p.res_p8) {   // AVS_F16P8 residual: 16 bytes of hi halves + 8 remainder bytes per run
          const char* base = p.residual + row0 * p.ldr * 3 + p8_hi;
          const long long step = (long long)RSTEP * p.ldr * 3;
#pragma unroll
          for (int it = 0; it < NU; ++it) {
            const bool ok = col_ok && row0 + it * RSTEP < row_lim;
            const char* rp = ok ? base + it * step : p.residual + p8_hi;
            rhi[mt][it] = *reinterpret_cast<const uint4*>(rp);
            const uint2 q = *reinterpret_cast<const uint2*>(rp + p8_rem);
            rlo[mt][it] = make_uint4(q.x, q.y, 0u, 0u);
          }
          return;
        }
        const char* base = p.residual + (row0 * p.ldr + col8) * 4;
        const long long step = (long long)RSTEP * p.ldr * 4;
#pragma unroll
        for (int it = 0; it < NU; ++it) {
          const bool ok = col_ok && row0 + it * RSTEP < row_lim;
          const uint4* rp = reinterpret_cast<const uint4*>(ok ? base + it * step : p.residual);
          rhi[mt][it] = rp[0];
          rlo[mt][it] = rp[1];
        }
      }
    };
    if (NORM && p.residual) res_fetch(std::integral_constant<int, 0>{});   // in flight during the statistics rounds
    if constexpr (EPI == EPI_STATS) {
      const int rpg = p.rows_per_group;
      float* const T1 = fl;                      // [WR][2][BN] sums of the wave rows' (at most two) groups
      float* const T2 = fl + WR * 2 * BN;        // [WR][2][BN] sums of squares about the tile's group mean
      const int tile_end = (m0 + A_ROWS < p.M ? m0 + A_ROWS : p.M);
      const int r_first = m0 + wr * 64;
      const bool live = r_first < p.M;
      int g0 = 0, g1 = -1, r_end = r_first;
      if (live) {
        r_end = (r_first + 64 < p.M ? r_first + 64 : p.M);     // one past the wave's last existing row
        g0 = r_first / rpg;
        g1 = (r_end - 1) / rpg;                                 // g1 - g0 <= 1 (rows_per_group >= 64)
      }
      // round 1: column sums per group of the wave's rows
      for (int g = g0; g <= g1; ++g) {
        const int lo = (g * rpg > r_first ? g * rpg : r_first) - r_first - 4 * lh;
        const int hi = ((g + 1) * rpg < r_end ? (g + 1) * rpg : r_end) - r_first - 4 * lh;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float s = wave_sum(nt, lo, hi);
          if (lh == 0) T1[(wr * 2 + (g - g0)) * BN + wc * WCOLS + nt * 32 + lr] = s;
        }
      }
      __syncthreads();
      // the sum of group g over the tile = the wave rows that hold rows of it, added in wave-row order
      auto tile_total = [&](const float* T, int g, int cc) -> float {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < WR; ++r) {
          const int rf = m0 + r * 64;
          if (rf >= p.M) break;
          const int re = (rf + 64 < p.M ? rf + 64 : p.M);
          const int a = rf / rpg, b = (re - 1) / rpg;
          if (g >= a && g <= b) s += T[(r * 2 + (g - a)) * BN + cc];
        }
        return s;
      };
      // round 2: sums of squares about the mean of the rows this TILE holds of the group
      for (int g = g0; g <= g1; ++g) {
        const int lo = (g * rpg > r_first ? g * rpg : r_first) - r_first - 4 * lh;
        const int hi = ((g + 1) * rpg < r_end ? (g + 1) * rpg : r_end) - r_first - 4 * lh;
        const int n_g = ((g + 1) * rpg < tile_end ? (g + 1) * rpg : tile_end) - (g * rpg > m0 ? g * rpg : m0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int cc = wc * WCOLS + nt * 32 + lr;
          const float mean = tile_total(T1, g, cc) / (float)n_g;
          const float s = wave_sq(nt, lo, hi, mean);
          if (lh == 0) T2[(wr * 2 + (g - g0)) * BN + cc] = s;
        }
      }
      __syncthreads();
      {
        const int g_lo = m0 / rpg;
        const int nj = (tile_end - 1) / rpg - g_lo + 1;  // <= p.stat_slots
        for (int i = t; i < nj * BN; i += 256) {
          const int j = i / BN, cc = i - j * BN;
          if (n0 + cc < p.N) {
            float* dst = p.stat_part + (((long long)tm * p.stat_slots + j) * 2) * p.N + n0 + cc;
            dst[0] = tile_total(T1, g_lo + j, cc);
            dst[p.N] = tile_total(T2, g_lo + j, cc);
          }
        }
      }
      __syncthreads();   // the tables alias the staging regions
    }
    if constexpr (EPI == EPI_BNLOCAL) {
      const int rpg = p.rows_per_group;
      const int ng = used / rpg;                   // whole groups (M is a multiple of rpg)
      float* const R1 = fl;                        // [WR][ng][BN]
      float* const R2 = fl + WR * BNLOCAL_MAX_GROUPS * BN;
      float* const TAB = R2 + WR * BNLOCAL_MAX_GROUPS * BN;   // [ng][scale | shift][BN]
      const int w_first = wr * 64;
      const bool live = w_first < used;
      int k0 = 0, k1 = -1, w_end = w_first;
      if (live) {
        w_end = (w_first + 64 < used ? w_first + 64 : used);
        k0 = w_first / rpg;
        k1 = (w_end - 1) / rpg;
      }
      for (int k = k0; k <= k1; ++k) {
        const int lo = (k * rpg > w_first ? k * rpg : w_first) - w_first - 4 * lh;
        const int hi = ((k + 1) * rpg < w_end ? (k + 1) * rpg : w_end) - w_first - 4 * lh;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float s = wave_sum(nt, lo, hi);
          if (lh == 0) R1[(wr * ng + k) * BN + wc * WCOLS + nt * 32 + lr] = s;
        }
      }
      __syncthreads();
      auto tile_total = [&](const float* R, int k, int cc) -> float {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < WR; ++r) {
          if (r * 64 >= used) break;
          const int re = (r * 64 + 64 < used ? r * 64 + 64 : used);
          const int a = (r * 64) / rpg, b = (re - 1) / rpg;
          if (k >= a && k <= b) s += R[(r * ng + k) * BN + cc];
        }
        return s;
      };
      const float inv_n = 1.f / (float)rpg;
      for (int k = k0; k <= k1; ++k) {
        const int lo = (k * rpg > w_first ? k * rpg : w_first) - w_first - 4 * lh;
        const int hi = ((k + 1) * rpg < w_end ? (k + 1) * rpg : w_end) - w_first - 4 * lh;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int cc = wc * WCOLS + nt * 32 + lr;
          const float s = wave_sq(nt, lo, hi, tile_total(R1, k, cc) * inv_n);
          if (lh == 0) R2[(wr * ng + k) * BN + cc] = s;
        }
      }
      __syncthreads();
      for (int i = t; i < ng * BN; i += 256) {
        const int k = i / BN, cc = i - k * BN;
        const float mean = tile_total(R1, k, cc) * inv_n;
        const float var = tile_total(R2, k, cc) * inv_n;
        const int col = n0 + cc;
        const float sc = (col < p.N ? p.gamma[col] : 0.f) / sqrtf(var + p.eps);
        TAB[(2 * k) * BN + cc] = sc;
        TAB[(2 * k + 1) * BN + cc] = (col < p.N ? p.beta[col] : 0.f) - mean * sc;
      }
      __syncthreads();
      {
        const int kmax = ng > 0 ? ng - 1 : 0;
        const int kb0 = k0 < kmax ? k0 : kmax;
        const int kb1 = kb0 + 1 < kmax ? kb0 + 1 : kmax, kb2 = kb0 + 2 < kmax ? kb0 + 2 : kmax;
        bnd1 = (kb0 + 1) * rpg - w_first - 4 * lh;
        bnd2 = bnd1 + rpg;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int cc = wc * WCOLS + nt * 32 + lr;
          sc0[nt] = TAB[(2 * kb0) * BN + cc], sf0[nt] = TAB[(2 * kb0 + 1) * BN + cc];
          sc1[nt] = TAB[(2 * kb1) * BN + cc], sf1[nt] = TAB[(2 * kb1 + 1) * BN + cc];
          sc2[nt] = TAB[(2 * kb2) * BN + cc], sf2[nt] = TAB[(2 * kb2 + 1) * BN + cc];
        }
      }
      __syncthreads();   // the tables alias the staging regions
    }
    if constexpr (EPI == EPI_AFFINE) {
      // the given affines of the (up to three) groups this wave's 64 rows lie in
      const int rpg = p.rows_per_group;
      const int r_first = m0 + wr * 64;
      const int gl = (p.M - 1) / rpg;                       // last group
      const int ga = r_first / rpg < gl ? r_first / rpg : gl;
      const int gb = ga + 1 < gl ? ga + 1 : gl, gc = ga + 2 < gl ? ga + 2 : gl;
      bnd1 = (ga + 1) * rpg - r_first - 4 * lh;
      bnd2 = bnd1 + rpg;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + wc * WCOLS + nt * 32 + lr;
        const bool cv = col < p.N;
        sc0[nt] = cv ? p.gamma[(long long)ga * p.N + col] : 0.f, sf0[nt] = cv ? p.beta[(long long)ga * p.N + col] : 0.f;
        sc1[nt] = cv ? p.gamma[(long long)gb * p.N + col] : 0.f, sf1[nt] = cv ? p.beta[(long long)gb * p.N + col] : 0.f;
        sc2[nt] = cv ? p.gamma[(long long)gc * p.N + col] : 0.f, sf2[nt] = cv ? p.beta[(long long)gc * p.N + col] : 0.f;
      }
    }
    // ---- staging + stores, 32 rows of the wave at a time
    if constexpr (NORM) {
      // normalise in place first: the affines' registers are free before the residual rows of the second half arrive
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int roff = mt * 32 + (e & 3) + 8 * (e >> 2);
            const bool in1 = roff >= bnd1, in2 = roff >= bnd2;
            const float sc = in2 ? sc2[nt] : (in1 ? sc1[nt] : sc0[nt]);
            const float sf = in2 ? sf2[nt] : (in1 ? sf1[nt] : sf0[nt]);
            acc[mt][nt][e] = fmaf(acc[mt][nt][e], sc, sf);
          }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          wreg[((e & 3) + 8 * (e >> 2) + 4 * lh) * H2_P + nt * 32 + lr] = acc[mt][nt][e];
      // half 1's residual rows: in flight while half 0 is processed (its accumulators have just left their registers)
      if (mt == 0 && NORM && p.residual) res_fetch(std::integral_constant<int, 1>{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const long long row0 = (long long)m0 + wr * 64 + mt * 32 + rl0;
      char* ybase = y + (row0 * p.ldc + col8) * 4;
      long long ystep = (long long)RSTEP * p.ldc * 4;
      if (p.nsplit > 0 && col8 >= p.nsplit) {   // the second destination of avs_conv2d_nhwc_split (a lane's run is whole)
        ybase = p.y2 + (row0 * p.ldc2 + (col8 - p.nsplit)) * 4;
        ystep = (long long)RSTEP * p.ldc2 * 4;
      }
#pragma unroll
      for (int it = 0; it < NU; ++it) {
        const float* src = wreg + (rl0 + it * RSTEP) * H2_P + grp * 8;
        const float4 f0 = *reinterpret_cast<const float4*>(src);
        const float4 f1 = *reinterpret_cast<const float4*>(src + 4);
        const long long row = row0 + it * RSTEP;
        if (!(col_ok && row < row_lim)) continue;
        if constexpr (RAGGED) {
          if (rl0 + it * RSTEP >= 32) continue;   // (the last round of a ragged map: past the staged half)
        }
        float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
        if constexpr (NORM) {
          if (p.residual) {
            float rv[8];
            // (through an empty asm: the conversions below then stay HERE - the scheduler otherwise expands every packed
            //  residual run to floats right behind the loads and the 128-column tiles spill)
            avs_pin(rhi[mt][it]);
            avs_pin(rlo[mt][it]);
            if (EPI == EPI_AFFINE && p.res_p8)
              avs_f16p8_join8(rhi[mt][it], make_uint2(rlo[mt][it].x, rlo[mt][it].y), rv);
            else
              avs_f16x2_join8(rhi[mt][it], rlo[mt][it], rv);
            if constexpr (EPI == EPI_AFFINE) {
              if (p.res_scale) {   // the residual is a raw convolution output: its BatchNorm rides in the add
                const long long gq = (row / p.rows_per_group) * p.N + col8;
                const float4 a0 = *reinterpret_cast<const float4*>(p.res_scale + gq);
                const float4 a1 = *reinterpret_cast<const float4*>(p.res_scale + gq + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(p.res_shift + gq);
                const float4 b1 = *reinterpret_cast<const float4*>(p.res_shift + gq + 4);
                const float as[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                const float bs[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) rv[j] = fmaf(rv[j], as[j], bs[j]);
              }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rv[j];
          }
          if (relu_l) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
          }
        }
        if (EPI == EPI_AFFINE && p.y_p8) {   // AVS_F16P8 output: 3 bytes per value
          uint4 hi;
          uint2 rem;
          avs_f16p8_split8(v, hi, rem);
          char* dst = y + (row * p.ldc) * 3 + p8_hi;
          *reinterpret_cast<uint4*>(dst) = hi;
          *reinterpret_cast<uint2*>(dst + p8_rem) = rem;
          continue;
        }
        uint4 hi, lo;
        avs_f16x2_split8(v, hi, lo);
        uint4* dst = reinterpret_cast<uint4*>(ybase + it * ystep);
        dst[0] = hi;
        dst[1] = lo;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  } else {
  // Register e of a 32x32 tile is row (e&3) + 8*(e>>2) + 4*lh, column lr.
  constexpr int E_CPRW = BN / 8;           // 16-byte chunks per bf16 tile row
  constexpr int E_RSTEP = 256 / E_CPRW;    // tile rows covered by one pass of the 256 threads
  constexpr int E_NIT = A_ROWS / E_RSTEP;
  // EPI_STATS: this tile's column sums per group.  A lane owns one column of each 32-wide tile: it sums its 2 x 16
  // rows per group (fp32 accumulators, before the bf16 rounding), the two lane halves are folded by one shuffle and
  // lanes 0..31 park the pair in LDS behind the staging tile: sred[wave row][group slot of the wave][sum | sumsq][BN].
  // stats_fold() (after the tile's next barrier) adds the wave rows in a fixed order and stores the tile's slots.
  // Rows past M hold exact zeros and add nothing.
  float* const sred = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + (size_t)CT_SLOTS * 16);
  if constexpr (EPI == EPI_STATS) {
    const int r_first = m0 + wr * 64;
    if (r_first < p.M) {
      const int r_last = (r_first + 63 < p.M ? r_first + 63 : p.M - 1);
      const int g0 = r_first / p.rows_per_group, g1 = r_last / p.rows_per_group;  // g1 - g0 <= 1 (launcher)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int cc = wc * WCOLS + nt * 32 + lr;
        if (g0 == g1) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const float v = acc[mt][nt][e];
              s1 += v;
              s2 = fmaf(v, v, s2);
            }
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 32, 64);
          if (lh == 0) {
            sred[((wr * 2 + 0) * 2 + 0) * BN + cc] = s1;
            sred[((wr * 2 + 0) * 2 + 1) * BN + cc] = s2;
          }
        } else {
          for (int g = g0; g <= g1; ++g) {
            const int lo = g * p.rows_per_group - r_first - 4 * lh, hi = lo + p.rows_per_group;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const int roff = mt * 32 + (e & 3) + 8 * (e >> 2);  // row - r_first - 4*lh
                const float v = (roff >= lo && roff < hi) ? acc[mt][nt][e] : 0.f;
                s1 += v;
                s2 = fmaf(v, v, s2);
              }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lh == 0) {
              sred[((wr * 2 + (g - g0)) * 2 + 0) * BN + cc] = s1;
              sred[((wr * 2 + (g - g0)) * 2 + 1) * BN + cc] = s2;
            }
          }
        }
      }
    }
  }
  // after a barrier that follows the block above: slot j of this row tile = group (m0 / rows_per_group) + j
  auto stats_fold = [&]() {
    if constexpr (EPI == EPI_STATS) {
      const int rpg = p.rows_per_group;
      const int m_last = (m0 + A_ROWS < p.M ? m0 + A_ROWS : p.M) - 1;
      const int g_lo = m0 / rpg;
      const int nj = m_last / rpg - g_lo + 1;  // <= p.stat_slots
      for (int i = t; i < nj * BN; i += 256) {
        const int j = i / BN, cc = i - j * BN;
        const int g = g_lo + j;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < WR; ++r) {
          const int r_first = m0 + r * 64;
          if (r_first >= p.M) break;
          const int r_last = (r_first + 63 < p.M ? r_first + 63 : p.M - 1);
          const int g0 = r_first / rpg, g1 = r_last / rpg;
          if (g >= g0 && g <= g1) {
            s1 += sred[((r * 2 + (g - g0)) * 2 + 0) * BN + cc];
            s2 += sred[((r * 2 + (g - g0)) * 2 + 1) * BN + cc];
          }
        }
        if (n0 + cc < p.N) {
          float* dst = p.stat_part + (((long long)tm * p.stat_slots + j) * 2) * p.N + n0 + cc;
          dst[0] = s1;
          dst[p.N] = s2;
        }
      }
    }
  };
  if constexpr (EPI == EPI_BNLOCAL) {
    static_assert(EPI != EPI_BNLOCAL || (ES == 2 && WR == 4), "the local BatchNorm epilogue is built on the 256-row bf16 tiles");
    char* ct = reinterpret_cast<char*>(lds);
    float* red = reinterpret_cast<float*>(lds);                   // [WR][groups][sum | sumsq][BN], dead before staging
    float* tab = reinterpret_cast<float*>(ct + CT_SLOTS * 16);    // [groups][scale | shift][BN]
    const int rpg = p.rows_per_group;
    const int used = (m0 + p.tile_rows <= p.M ? p.tile_rows : p.M - m0);  // rows of this tile that exist
    const int ng = used / rpg;                                            // whole groups (M is a multiple of rpg)
    // 0. every residual row of this thread in flight BEFORE the statistics rounds (their four barriers hide the memory
    //    latency; rows that do not exist read the residual's first bytes, never used)
    uint4 r4[E_NIT];
    {
      const int srow = t / E_CPRW, sch = t - srow * E_CPRW;
      const int col = n0 + sch * 8;
      if (p.residual && col < p.N) {
#pragma unroll
        for (int it = 0; it < E_NIT; ++it) {
          const int lrow = srow + it * E_RSTEP;
          r4[it] = *reinterpret_cast<const uint4*>(p.residual + (lrow < used ? ((long long)(m0 + lrow) * p.ldr + col) * 2 : 0));
        }
      }
    }
    // 1. zero the per-wave partial sums
    for (int i = t; i < WR * ng * 2 * BN; i += 256) red[i] = 0.f;
    __syncthreads();
    // 2. per-wave masked column sums of the groups this wave's 64 rows touch
    {
      const int w_first = wr * 64;
      if (w_first < used) {
        const int w_last = (w_first + 63 < used ? w_first + 63 : used - 1);
        const int k0 = w_first / rpg, k1 = w_last / rpg;
        if (k0 == k1) {
          // the wave's 64 rows lie in one group (always so for groups of >= 64 rows that divide the tile evenly):
          // plain column sums - rows past `used` hold exact zeros
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const float v = acc[mt][nt][e];
                s1 += v;
                s2 = fmaf(v, v, s2);
              }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lh == 0) {
              const int cc = wc * WCOLS + nt * 32 + lr;
              red[((wr * ng + k0) * 2 + 0) * BN + cc] = s1;
              red[((wr * ng + k0) * 2 + 1) * BN + cc] = s2;
            }
          }
        } else {
          for (int k = k0; k <= k1; ++k) {
            const int lo = k * rpg - w_first - 4 * lh, hi = lo + rpg;   // in units of roff
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              float s1 = 0.f, s2 = 0.f;
#pragma unroll
              for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                  const int roff = mt * 32 + (e & 3) + 8 * (e >> 2);
                  const float v = (roff >= lo && roff < hi) ? acc[mt][nt][e] : 0.f;
                  s1 += v;
                  s2 = fmaf(v, v, s2);
                }
              s1 += __shfl_xor(s1, 32, 64);
              s2 += __shfl_xor(s2, 32, 64);
              if (lh == 0) {
                const int cc = wc * WCOLS + nt * 32 + lr;
                red[((wr * ng + k) * 2 + 0) * BN + cc] = s1;
                red[((wr * ng + k) * 2 + 1) * BN + cc] = s2;
              }
            }
          }
        }
      }
    }
    __syncthreads();
    // 3. folded affine per (group, column): the four row-waves' partial sums added in a fixed order
    {
      const float inv_n = 1.f / (float)rpg;
      for (int i = t; i < ng * BN; i += 256) {
        const int k = i / BN, cc = i - k * BN;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < WR; ++r) {
          s1 += red[((r * ng + k) * 2 + 0) * BN + cc];
          s2 += red[((r * ng + k) * 2 + 1) * BN + cc];
        }
        const float mean = s1 * inv_n;
        const float var = fmaxf(s2 * inv_n - mean * mean, 0.f);
        const int col = n0 + cc;
        const float sc = (col < p.N ? p.gamma[col] : 0.f) / sqrtf(var + p.eps);
        tab[(2 * k) * BN + cc] = sc;
        tab[(2 * k + 1) * BN + cc] = (col < p.N ? p.beta[col] : 0.f) - mean * sc;
      }
    }
    __syncthreads();
    // 4. normalise into the bf16 staging tile (which overwrites the partial sums)
    const bool relu_now = p.act == AVS_ACT_RELU && p.residual == nullptr;
    {
      char* cbase = ct + (wr * 64 + 4 * lh) * CT_PITCH + (wc * WCOLS + lr) * 2;
      const int rel0 = wr * 64 + 4 * lh;        // tile row of roff = 0
      const int kmax = ng > 0 ? ng - 1 : 0;
      const int kbase = (wr * 64) / rpg < kmax ? (wr * 64) / rpg : kmax;   // first group of this wave's rows
      const int klast = (wr * 64 + 63) / rpg < kmax ? (wr * 64 + 63) / rpg : kmax;
      if (kbase == klast) {
        // one group for the whole wave: the affine of a column is two registers
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int cc = wc * WCOLS + nt * 32 + lr;
          const float sc = tab[(2 * kbase) * BN + cc], sf = tab[(2 * kbase + 1) * BN + cc];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {   // rows r, r + 1 of one column: one conversion, ReLU on the packed pair
              const int roff = mt * 32 + (e & 3) + 8 * (e >> 2);
              unsigned pk = avs_pack_bf16x2(fmaf(acc[mt][nt][e], sc, sf), fmaf(acc[mt][nt][e + 1], sc, sf));
              if (relu_now) pk = avs_relu_bf16x2(pk);
              *reinterpret_cast<unsigned short*>(cbase + roff * CT_PITCH + nt * 64) = (unsigned short)pk;
              *reinterpret_cast<unsigned short*>(cbase + (roff + 1) * CT_PITCH + nt * 64) = (unsigned short)(pk >> 16);
            }
        }
      } else {
        // up to three groups in the wave's 64 rows: their affines in registers, chosen per row by two compares
        const int b1 = (kbase + 1) * rpg - rel0, b2 = b1 + rpg;
        const int kb1 = kbase + 1 < kmax ? kbase + 1 : kmax, kb2 = kbase + 2 < kmax ? kbase + 2 : kmax;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int cc = wc * WCOLS + nt * 32 + lr;
          const float sc0 = tab[(2 * kbase) * BN + cc], sf0 = tab[(2 * kbase + 1) * BN + cc];
          const float sc1 = tab[(2 * kb1) * BN + cc], sf1 = tab[(2 * kb1 + 1) * BN + cc];
          const float sc2 = tab[(2 * kb2) * BN + cc], sf2 = tab[(2 * kb2 + 1) * BN + cc];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
              float v2[2];
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const int roff = mt * 32 + ((e + h) & 3) + 8 * ((e + h) >> 2);
                const bool g1 = roff >= b1, g2 = roff >= b2;
                const float sc = g2 ? sc2 : (g1 ? sc1 : sc0);
                const float sf = g2 ? sf2 : (g1 ? sf1 : sf0);
                v2[h] = fmaf(acc[mt][nt][e + h], sc, sf);
              }
              const int roff = mt * 32 + (e & 3) + 8 * (e >> 2);
              unsigned pk = avs_pack_bf16x2(v2[0], v2[1]);
              if (relu_now) pk = avs_relu_bf16x2(pk);
              *reinterpret_cast<unsigned short*>(cbase + roff * CT_PITCH + nt * 64) = (unsigned short)pk;
              *reinterpret_cast<unsigned short*>(cbase + (roff + 1) * CT_PITCH + nt * 64) = (unsigned short)(pk >> 16);
            }
        }
      }
    }
    __syncthreads();
    // 5. 16-byte row-major stores (+ residual, + ReLU)
    {
      const int srow = t / E_CPRW, sch = t - srow * E_CPRW;
      const int col = n0 + sch * 8;
      const bool relu_res = p.act == AVS_ACT_RELU;
      if (col < p.N) {
        // (the residual rows were fetched at the top of the epilogue)
#pragma unroll
        for (int it = 0; it < E_NIT; ++it) {
          const int lrow = srow + it * E_RSTEP;
          if (lrow >= used) break;
          const long long row = m0 + lrow;
          uint4 v = *reinterpret_cast<const uint4*>(ct + lrow * CT_PITCH + sch * 16);
          if (p.residual) {
            unsigned vv[4] = {v.x, v.y, v.z, v.w};
            const unsigned rv[4] = {r4[it].x, r4[it].y, r4[it].z, r4[it].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float lo = __uint_as_float(vv[j] << 16) + __uint_as_float(rv[j] << 16);
              const float hi = __uint_as_float(vv[j] & 0xffff0000u) + __uint_as_float(rv[j] & 0xffff0000u);
              vv[j] = avs_pack_bf16x2(lo, hi);
              if (relu_res) vv[j] = avs_relu_bf16x2(vv[j]);
            }
            v = make_uint4(vv[0], vv[1], vv[2], vv[3]);
          }
          *reinterpret_cast<uint4*>(y + (row * p.ldc + col) * 2) = v;
        }
      }
    }
    return;
  }
  if constexpr (ES == 2) {
    // bf16: through LDS, then 16-byte row-major stores.  Every per-element LDS address is a per-lane base plus
    // a compile-time offset (padded pitch, no swizzle arithmetic): for the short reductions this epilogue,
    // not the matrix work, is what a tile costs.
    constexpr int CPRW = BN / 8;          // 16-byte chunks per tile row
    constexpr int RSTEP = 256 / CPRW;     // tile rows covered by one pass of the 256 threads
    char* ct = reinterpret_cast<char*>(lds);
    {
      char* cbase = ct + (wr * 64 + 4 * lh) * CT_PITCH + (wc * WCOLS + lr) * 2;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float bcol = 0.f;
          if constexpr (EPI == EPI_ANY) {
            const int col = n0 + wc * WCOLS + nt * 32 + lr;
            bcol = (p.bias_mode == AVS_BIAS_COL && col < p.N) ? bias[col] : 0.f;
          }
          if constexpr (EPI == EPI_BRELU) {
            const int col = n0 + wc * WCOLS + nt * 32 + lr;
            bcol = col < p.N ? bias[col] : 0.f;
          }
#pragma unroll
          for (int e = 0; e < 16; e += 2) {   // rows r, r + 1 of one column: one conversion, ReLU on the packed pair
            const int roff = mt * 32 + (e & 3) + 8 * (e >> 2);
            float v2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              float v = acc[mt][nt][e + h];
              if constexpr (EPI == EPI_ANY) {
                v = fmaf(v, p.alpha, bcol);
                if (p.bias_mode == AVS_BIAS_ROW) {
                  const int row = m0 + wr * 64 + 4 * lh + roff + h;
                  if (row < p.M) v += bias[row];
                }
              }
              if constexpr (EPI == EPI_BRELU) v = v + bcol;
              v2[h] = v;
            }
            unsigned pk = avs_pack_bf16x2(v2[0], v2[1]);
            if constexpr (EPI == EPI_ANY) {
              if (p.act == AVS_ACT_RELU) pk = avs_relu_bf16x2(pk);
            }
            if constexpr (EPI == EPI_BRELU) pk = avs_relu_bf16x2(pk);
            *reinterpret_cast<unsigned short*>(cbase + roff * CT_PITCH + nt * 64) = (unsigned short)pk;
            *reinterpret_cast<unsigned short*>(cbase + (roff + 1) * CT_PITCH + nt * 64) = (unsigned short)(pk >> 16);
          }
        }
    }
    __syncthreads();
    const int srow = t / CPRW, sch = t - srow * CPRW;
    const char* srcp = ct + srow * CT_PITCH + sch * 16;
    const int col = n0 + sch * 8;
    char* dst = y + ((long long)(m0 + srow) * p.ldc + col) * 2;
    const long long dstep = (long long)RSTEP * p.ldc * 2;
    const bool vec_ok = ((p.ldc * 2) % 16 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
    if (!AVS_DEBUG_BIT(p, 1)) {
      if (vec_ok && m0 + A_ROWS <= p.M && n0 + BN <= p.N) {
#pragma unroll
        for (int it = 0; it < A_ROWS / RSTEP; ++it)
          *reinterpret_cast<uint4*>(dst + it * dstep) = *reinterpret_cast<const uint4*>(srcp + it * RSTEP * CT_PITCH);
      } else {
        for (int it = 0; it < A_ROWS / RSTEP; ++it) {
          const int row = m0 + srow + it * RSTEP;
          if (row >= p.M || col >= p.N) continue;
          const char* sp = srcp + it * RSTEP * CT_PITCH;
          char* dp = dst + it * dstep;
          if (vec_ok && col + 8 <= p.N) {
            *reinterpret_cast<uint4*>(dp) = *reinterpret_cast<const uint4*>(sp);
          } else {
            for (int j = 0; j < 8 && col + j < p.N; ++j)
              reinterpret_cast<unsigned short*>(dp)[j] = reinterpret_cast<const unsigned short*>(sp)[j];
          }
        }
      }
    }
    stats_fold();  // behind the barrier above: every wave's column sums are in LDS
  } else {
    if constexpr (EPI == EPI_STATS) __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + wc * WCOLS + nt * 32 + lr;
        if (col >= p.N) continue;
        float bcol = 0.f;
        if constexpr (EPI == EPI_ANY) bcol = (p.bias_mode == AVS_BIAS_COL) ? bias[col] : 0.f;
        if constexpr (EPI == EPI_BRELU) bcol = bias[col];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wr * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (row >= p.M) continue;
          float v;
          if constexpr (ACC64)
            v = (float)(acc64[mt][nt][e] * (double)p.alpha) + bcol;
          else if constexpr (EPI == EPI_ANY)
            v = fmaf(acc[mt][nt][e], p.alpha, bcol);
          else if constexpr (EPI == EPI_BRELU)
            v = fmaxf(acc[mt][nt][e] + bcol, 0.f);
          else
            v = acc[mt][nt][e];
          if constexpr (EPI == EPI_ANY) {
            if (p.bias_mode == AVS_BIAS_ROW) v += bias[row];
            if (p.act == AVS_ACT_RELU) v = fmaxf(v, 0.f);
          }
          *reinterpret_cast<float*>(y + ((long long)row * p.ldc + col) * 4) = v;
        }
      }
    stats_fold();
  }
  }  // SPLIT != 2
}

#ifdef AVS_STUDY
static int g_debug_flags = 0;
extern "C" void avs_debug_flags(int flags) { g_debug_flags = flags; }
#endif
// The rules' thresholds: compile-time constants in the shipped library (no process-global mutable state); the
// kernel-study build makes them settable (tools/).
#ifdef AVS_STUDY
#define AVS_RULE static int
#else
#define AVS_RULE static constexpr int
#endif
AVS_RULE g_rowb_threshold_bytes = 2048;  // reductions of at most this many bytes per row use 64-byte steps
AVS_RULE g_tall_min_tiles = 2048;        // 256-row tiles by rule: at least this many of them (~4 full waves of workgroups)
AVS_RULE g_tall_min_k_bytes = 1024;      // ... and, at BN = 128, a reduction long enough to be bound by the loop
AVS_RULE g_pipe3 = 1;                    // the 3-buffer hand-counted pipeline for the 64-byte-row variants
AVS_RULE g_bnlocal = 1;                  // 0: avs_conv2d_bnlocal_tile_rows declines every shape
static constexpr int AVS_RULE_AFFINE_128_MAX_K = 128;   // the given-affine 1x1 form: 128-row tiles up to this reduction length
#ifdef AVS_STUDY
extern "C" void avs_tune_short_reduction_bytes(int bytes) { g_rowb_threshold_bytes = bytes; }
extern "C" void avs_tune_tall_rule(int min_tiles, int min_k_bytes) {
  if (min_tiles > 0) g_tall_min_tiles = min_tiles;
  if (min_k_bytes >= 0) g_tall_min_k_bytes = min_k_bytes;
}
extern "C" void avs_tune_pipeline(int enabled) { g_pipe3 = enabled; }
extern "C" void avs_tune_bnlocal(int enabled) { g_bnlocal = enabled; }
#endif

// The shapes the nine-tap form takes (AVS_F16X2 convolution + statistics on the 256-row tiles): 3x3 / stride 1 / pad 1,
// output = input geometry, a dense NHWC input at most 63 pixels wide (a tile's rows + W + 1 rows of halo on either side
// fit the 384-row A buffer), the caller not forcing the 256-row tile's classic walk (AVS_TILE_256 keeps it).
static bool igemm_tap9_ok(const IgemmParams& p, int es) {
  return (p.variant & 3) != AVS_TILE_256 && p.KW == 3 && p.K == 9 * p.cin && p.sh == 1 && p.sw == 1 && p.ph == 1 && p.pw == 1 &&
         p.W <= 63 && p.HoWo == p.H * p.W && p.Wo == p.W && p.x_px_stride == p.cin &&
         p.x_row_stride == (long long)p.W * p.cin && p.x_img_stride == (long long)p.HoWo * p.cin && p.cin % (64 / es) == 0 &&
         (long long)(384 + 64) * p.cin * es < (1ll << 31);
}

// The shapes the shifted-row form takes under the bias + ReLU epilogue (AVS_F16X2, 256-row tiles): any KH x KW / stride 1 with
// "same" padding (output = input geometry) whose reach ph W + pw fits the 64 halo rows either side of a tile - Inception-v3's
// 1x7 / 7x1 at 17x17, 3x3 at 35x35 and 8x8, 1x3 / 3x1 at 8x8; the input dense in pixels (a channel slice of a wider NHWC
// tensor is fine: pixel stride >= cin); at most 32 taps; the caller not forcing the classic walk (AVS_TILE_256 keeps it).
static bool igemm_taps_ok(const IgemmParams& p) {
  if ((p.variant & 3) == AVS_TILE_256 || p.cin <= 0 || p.KW <= 0 || p.K % (p.cin * p.KW) != 0) return false;
  const int kh = p.K / (p.cin * p.KW);
  return kh * p.KW > 1 && kh * p.KW <= 32 && p.sh == 1 && p.sw == 1 && 2 * p.ph + 1 == kh && 2 * p.pw + 1 == p.KW &&
         p.HoWo == p.H * p.W && p.Wo == p.W && p.ph * p.W + p.pw <= 64 && p.x_px_stride >= p.cin &&
         p.x_row_stride == (long long)p.W * p.x_px_stride && p.x_img_stride == (long long)p.HoWo * p.x_px_stride &&
         p.cin % 16 == 0 && (long long)(384 + 64) * p.x_px_stride * 4 + (long long)p.cin * 4 < (1ll << 31);
}

// every dispatcher returns whether a kernel was launched: a (dtype, tile, epilogue) combination that has no instantiation
// is an error of the launcher's rules, reported as AVS_E_UNSUPPORTED - never a silent AVS_OK with an untouched output
#define AVS_LAUNCH_RET(K)                                     \
  do {                                                        \
    hipLaunchKernelGGL(K, grid, dim3(256), 0, stream, p);     \
    return true;                                              \
  } while (0)
template <int ES, int BN, bool ACC64, bool SP, int ROWB, bool PIPE, int WR, bool FK>
static bool igemm_dispatch_epi4(int epi, dim3 grid, hipStream_t stream, const IgemmParams& p) {
  if constexpr (ES == 2 && WR == 4) {
    if constexpr (SP && ROWB == 64 && PIPE && FK) {   // bf16 3x3 / 1 layers: the nine-tap form
      if ((epi == EPI_BNLOCAL || epi == EPI_STATS) && igemm_tap9_ok(p, ES)) {
        if (epi == EPI_BNLOCAL)
          AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_BNLOCAL, PIPE, WR, FK, 0, true>));
        else
          AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_STATS, PIPE, WR, FK, 0, true>));
      }
    }
    if (epi == EPI_BNLOCAL) {
      AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_BNLOCAL, PIPE, WR, FK>));
    }
  }
  if constexpr (ES == 4 && !ACC64) {
    if (p.split == 2) {         // AVS_F16X2: operands stored as fp16 hi | lo runs, three fp16 MFMAs per product
      if (epi == EPI_PLAIN)
        AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_PLAIN, PIPE, WR, FK, 2>));
      else if (epi == EPI_STATS) {
        if constexpr (SP && ROWB == 64 && PIPE && WR == 4 && FK) {
          if (igemm_tap9_ok(p, ES)) {   // 3x3 / 1 / pad 1 on a dense input: one A fetch per channel block serves all nine taps
            AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_STATS, PIPE, WR, FK, 2, true>));
          }
        }
        if constexpr (!SP && ROWB == 64 && PIPE && FK) {
          if (p.x_p8) {   // AVS_F16P8 input (validated by igemm_launch): A fragments fetched into registers
            AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_STATS, PIPE, WR, FK, 2, false, true>));
          }
        }
        AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_STATS, PIPE, WR, FK, 2>));
      } else if (epi == EPI_BRELU) {
        if constexpr (SP && ROWB == 64 && PIPE && WR == 4 && FK) {
          if (igemm_taps_ok(p)) {   // stride-1 "same" filters: one A fetch per channel block serves every tap
            AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_BRELU, PIPE, WR, FK, 2, true>));
          }
        }
        AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_BRELU, PIPE, WR, FK, 2>));
      } else if constexpr (WR == 4) {
        if (epi == EPI_BNLOCAL) {
          if constexpr (SP && ROWB == 64 && PIPE && FK) {
            if (igemm_tap9_ok(p, ES)) {
              AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_BNLOCAL, PIPE, WR, FK, 2, true>));
            }
          }
          AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_BNLOCAL, PIPE, WR, FK, 2>));
        }
        if constexpr (!SP && ROWB == 64 && PIPE) {   // (the launcher only sends 1x1 shapes on the pipelined 256-row tiles here)
          if (epi == EPI_AFFINE)
            AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_AFFINE, PIPE, WR, FK, 2>));
        }
      } else if constexpr (!SP && ROWB == 64 && PIPE && BN == 128) {   // AVS_TILE_128: the given-affine form on 128-row tiles
        if (epi == EPI_AFFINE)
          AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_AFFINE, PIPE, WR, FK, 2>));
      }
      return false;
    }
    if (p.split || WR == 4) {   // AVS_F32_SPLIT: the same tiles, products as three bf16 MFMAs (WR = 4 exists for it only)
      if (epi == EPI_PLAIN)
        AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_PLAIN, PIPE, WR, FK, 1>));
      else if (epi == EPI_STATS)
        AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_STATS, PIPE, WR, FK, 1>));
      else if (epi == EPI_BRELU)
        AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_BRELU, PIPE, WR, FK, 1>));
      else if constexpr (WR == 2)
        AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_ANY, PIPE, WR, FK, 1>));
    }
  }
  if constexpr (!(ES == 4 && WR == 4)) {   // (fp32 on 256-row tiles exists as fp32-split only: handled above)
    if (epi == EPI_PLAIN)
      AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_PLAIN, PIPE, WR, FK>));
    else if (epi == EPI_STATS)
      AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_STATS, PIPE, WR, FK>));
    else if (epi == EPI_BRELU)
      AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_BRELU, PIPE, WR, FK>));
    else if constexpr (WR == 2)
      AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_ANY, PIPE, WR, FK>));
  }
  return false;
}


// FASTK (the scalar tap walk over buffer_load ... lds): cin a whole number of reduction steps of bke elements, at most 32
// taps, and the buffer window - a tile's rows (at most 256, spread over whole images) plus the tap walk - below 2 GiB
static bool igemm_fastk_ok(const IgemmParams& p, int es, int bke, int bn) {
  const long long rows = 256;
  long long extent;
  if (p.lin_stride >= 0)
    extent = rows * p.lin_stride + p.K;
  else
    extent = (rows / p.HoWo + 2) * p.x_img_stride + (long long)(p.K / (p.cin * p.KW) + p.ph) * p.x_row_stride +
             (long long)(p.KW + p.pw) * p.x_px_stride + p.cin;
  const bool window_ok = extent * es < (1ll << 31) && (long long)bn * p.ldb * es + (long long)p.K * es < (1ll << 31) &&
                         p.x_img_stride >= 0 && p.x_row_stride >= 0 && p.x_px_stride >= 0;
  return !(p.variant & AVS_STAGING_GENERIC) && window_ok && p.cin % bke == 0 && p.K % bke == 0 && p.K / p.cin <= 32 &&
         p.K % p.cin == 0;
}

template <int ES, int BN, bool ACC64, bool SP, int ROWB, bool PIPE, int WR>
static bool igemm_dispatch_epi3(int epi, dim3 grid, hipStream_t stream, const IgemmParams& p) {
  if (igemm_fastk_ok(p, ES, ROWB / ES, BN))
    return igemm_dispatch_epi4<ES, BN, ACC64, SP, ROWB, PIPE, WR, true>(epi, grid, stream, p);
  else
    return igemm_dispatch_epi4<ES, BN, ACC64, SP, ROWB, PIPE, WR, false>(epi, grid, stream, p);
}

// AVS_F16X2 bias + ReLU on 256 x 96 tiles (igemm_launch's rule: cout = 96, 160, 192, 288 ... with enough rows)
static bool igemm_dispatch_brelu96(bool spatial, dim3 grid, hipStream_t stream, const IgemmParams& p) {
#ifdef AVS_STUDY
  const_cast<IgemmParams&>(p).debug = g_debug_flags;
#endif
  const bool fk = igemm_fastk_ok(p, 4, 16, 96);
  if (spatial) {
    if (fk && igemm_taps_ok(p)) AVS_LAUNCH_RET((igemm_kernel<4, 96, false, true, 64, EPI_BRELU, true, 4, true, 2, true>));
    if (fk) AVS_LAUNCH_RET((igemm_kernel<4, 96, false, true, 64, EPI_BRELU, true, 4, true, 2>));
    AVS_LAUNCH_RET((igemm_kernel<4, 96, false, true, 64, EPI_BRELU, true, 4, false, 2>));
  }
  if (fk) AVS_LAUNCH_RET((igemm_kernel<4, 96, false, false, 64, EPI_BRELU, true, 4, true, 2>));
  AVS_LAUNCH_RET((igemm_kernel<4, 96, false, false, 64, EPI_BRELU, true, 4, false, 2>));
}

template <int ES, int BN, bool ACC64, bool SP, int ROWB, bool PIPE>
static bool igemm_dispatch_epi2(int epi, dim3 grid, hipStream_t stream, const IgemmParams& p) {
  if constexpr (ACC64) {
    AVS_LAUNCH_RET((igemm_kernel<ES, BN, ACC64, SP, ROWB, EPI_ANY, false>));
  } else {
    if constexpr (ROWB == 64 && PIPE) {
      if (p.tall) {  // igemm_launch only sets it for the epilogue forms compiled at WR = 4 (bf16; fp32 only as fp32-split)
        return igemm_dispatch_epi3<ES, BN, ACC64, SP, ROWB, PIPE, 4>(epi, grid, stream, p);
      }
    }
    return igemm_dispatch_epi3<ES, BN, ACC64, SP, ROWB, PIPE, 2>(epi, grid, stream, p);
  }
}

template <int ES, int BN, bool ACC64, bool SP, int ROWB>
static bool igemm_dispatch_epi(int epi, dim3 grid, hipStream_t stream, const IgemmParams& p) {
  if constexpr (ROWB == 64 && !ACC64) {
    if (g_pipe3 && p.K * ES > 2 * ROWB) {  // at least three steps, else there is nothing to pipeline
      return igemm_dispatch_epi2<ES, BN, ACC64, SP, ROWB, true>(epi, grid, stream, p);
    }
  }
  return igemm_dispatch_epi2<ES, BN, ACC64, SP, ROWB, false>(epi, grid, stream, p);
}

template <int ES, int BN, bool ACC64>
static bool igemm_dispatch(bool spatial, dim3 grid, hipStream_t stream, const IgemmParams& p) {
#ifdef AVS_STUDY
  const_cast<IgemmParams&>(p).debug = g_debug_flags;
#endif
  // 64-byte rows: short reductions, the 256-row tiles, and shapes whose channel count fits the scalar tap walk only
  // at the 64-byte step (cin = 96, 160, 288 ... of Inception-v3): the cheaper staging is worth more than the longer step
  const bool fast64_only = !(p.variant & AVS_STAGING_GENERIC) && p.cin % (64 / ES) == 0 && p.cin % (128 / ES) != 0 && p.K / p.cin <= 32;
  const bool short_k = p.tall || fast64_only || (long long)p.K * ES <= g_rowb_threshold_bytes;
  int epi = EPI_ANY;
  if (p.tile_rows)
    epi = EPI_BNLOCAL;
  else if (p.affine)
    epi = EPI_AFFINE;
  else if (p.bias_mode == AVS_BIAS_NONE && p.act == AVS_ACT_NONE && p.alpha == 1.0f)
    epi = p.stat_part ? EPI_STATS : EPI_PLAIN;
  else if (p.bias_mode == AVS_BIAS_COL && p.act == AVS_ACT_RELU && p.alpha == 1.0f && !p.stat_part)
    epi = EPI_BRELU;
  if (short_k) {
    if (spatial)
      return igemm_dispatch_epi<ES, BN, ACC64, true, 64>(epi, grid, stream, p);
    else
      return igemm_dispatch_epi<ES, BN, ACC64, false, 64>(epi, grid, stream, p);
  } else {
    if (spatial)
      return igemm_dispatch_epi<ES, BN, ACC64, true, 128>(epi, grid, stream, p);
    else
      return igemm_dispatch_epi<ES, BN, ACC64, false, 128>(epi, grid, stream, p);
  }
}

// plan_only: validate the shape and choose the tile (p.tall, p.tiles_n, *tiles_m_out) as for a launch, launch nothing
static int igemm_launch(int dtype, IgemmParams& p, int batch, hipStream_t stream, const char* who,
                        bool plan_only = false, long long* tiles_m_out = nullptr) {
  const int es = dtype == AVS_BF16 ? 2 : 4;
  p.split = dtype == AVS_F32_SPLIT ? 1 : (dtype == AVS_F16X2 ? 2 : 0);
  const int ce = dtype == AVS_F16X2 ? 8 : 16 / es;   // AVS_F16X2: whole hi | lo runs of 8 slots
  AVS_REQUIRE(dtype == AVS_F32 || dtype == AVS_BF16 || dtype == AVS_F32_ACC64 || dtype == AVS_F32_SPLIT ||
                  dtype == AVS_F16X2, AVS_E_ARG, "%s: bad dtype %d", who, dtype);
  if (dtype == AVS_F16X2) {
    const bool fixed = p.alpha == 1.0f &&
                       ((p.bias_mode == AVS_BIAS_NONE && (p.act == AVS_ACT_NONE || p.tile_rows || p.affine)) ||
                        (p.bias_mode == AVS_BIAS_COL && p.act == AVS_ACT_RELU && !p.stat_part && !p.tile_rows && !p.affine));
    AVS_REQUIRE(fixed, AVS_E_UNSUPPORTED,
                "%s: AVS_F16X2 takes alpha = 1 and either no bias / activation (or a BatchNorm form) or bias per column + ReLU", who);
    AVS_REQUIRE(p.N % 8 == 0 && p.ldc % 8 == 0 && (p.sC % 8) == 0, AVS_E_SHAPE,
                "%s: AVS_F16X2 needs cout and the output strides in multiples of 8 slots", who);
    AVS_REQUIRE(plan_only || ((((uintptr_t)p.x) | ((uintptr_t)p.w) | ((uintptr_t)p.y)) & 31u) == 0, AVS_E_ALIGN,
                "%s: AVS_F16X2 operands must be 32-byte aligned", who);
  }
  AVS_REQUIRE(p.M >= 0 && p.N > 0 && p.K > 0 && batch > 0, AVS_E_SHAPE, "%s: bad sizes M=%d N=%d K=%d batch=%d", who,
              p.M, p.N, p.K, batch);
  if (p.M == 0) return AVS_OK;
  AVS_REQUIRE(plan_only || (p.x && p.w && p.y), AVS_E_ARG, "%s: null operand", who);
  AVS_REQUIRE(p.bias_mode == AVS_BIAS_NONE || p.bias, AVS_E_ARG, "%s: bias mode %d without bias", who, p.bias_mode);
  AVS_REQUIRE(p.cin % ce == 0 && p.K % ce == 0, AVS_E_SHAPE,
              "%s: cin=%d / K=%d must be multiples of %d elements (16 bytes)", who, p.cin, p.K, ce);
  AVS_REQUIRE(plan_only || (avs_aligned16(p.x) && avs_aligned16(p.w)), AVS_E_ALIGN, "%s: x / w must be 16-byte aligned",
              who);
  AVS_REQUIRE((p.x_img_stride * es) % (ce * es) == 0 && (p.x_row_stride * es) % (ce * es) == 0 &&
                  (p.x_px_stride * es) % (ce * es) == 0 && (p.ldb * es) % (ce * es) == 0 && (p.sA * es) % (ce * es) == 0 &&
                  (p.sB * es) % (ce * es) == 0,
              AVS_E_ALIGN, "%s: strides must be multiples of %d bytes", who, ce * es);
  if (p.nsplit > 0) {                                             // two destinations: each row stride covers its own columns
    AVS_REQUIRE(p.ldc >= p.nsplit && p.ldc2 >= p.N - p.nsplit, AVS_E_SHAPE,
                "%s: output row strides %lld / %lld < the %d / %d columns of the two destinations", who, p.ldc, p.ldc2,
                p.nsplit, p.N - p.nsplit);
  } else {
    AVS_REQUIRE(p.ldc >= p.N, AVS_E_SHAPE, "%s: output row stride %lld < N=%d", who, p.ldc, p.N);
  }
  AVS_REQUIRE(batch <= 65535, AVS_E_SHAPE, "%s: batch %d > 65535", who, batch);

  // exact fp32 with few tiles (the scorer's linear layers and their gradients: 1800 rows x 512 .. 2048 columns = 16 .. 240
  // tiles of 128 x 128 on 256 CUs x 3 workgroups): 64-wide column tiles double the workgroups; the same sums in the same order
  const bool few_f32 = dtype == AVS_F32 && batch == 1 && !p.stat_part && !p.tile_rows && !p.affine &&
                       (((long long)p.M + 127) / 128) * ((p.N + 127) / 128) < 256;
  const bool narrow = p.N <= 64 || dtype == AVS_F32_ACC64 || few_f32;
  int bn = narrow ? 64 : 128;
  p.tiles_n = (p.N + bn - 1) / bn;
  // AVS_F16X2 bias + ReLU (Inception-v3's folded-BatchNorm convolutions): cout = 96, 160, 192, 288 ... leave a 128-wide
  // column tile 25 - 37 % empty; 96-wide tiles on the 256-row form when they save at least 15 % of the padded width and the
  // 256-row rule below holds for them (a tuning choice: the outputs do not depend on it; AVS_TILE_128 keeps the 128-row tiles)
  bool wide96 = false;
  if (dtype == AVS_F16X2 && !narrow && p.alpha == 1.0f && p.bias_mode == AVS_BIAS_COL && p.act == AVS_ACT_RELU && !p.stat_part &&
      !p.tile_rows && !p.affine && batch == 1 && g_pipe3 && (long long)p.K * 4 >= g_tall_min_k_bytes &&
      ((p.variant & 3) == AVS_TILE_AUTO || (p.variant & 3) == AVS_TILE_256)) {
    const int t96 = (p.N + 95) / 96;
    const long long tall_tiles96 = ((long long)p.M + 255) / 256 * t96;
    if (t96 * 96 * 100 <= p.tiles_n * 128 * 85 && ((p.variant & 3) == AVS_TILE_256 || tall_tiles96 >= g_tall_min_tiles)) {
      wide96 = true;
      bn = 96;
      p.tiles_n = t96;
    }
  }
  // 256-row tiles (WR = 4): bf16, the compile-time epilogue forms, a reduction of at least three 64-byte steps (the
  // variants are built on the 3-buffer pipeline), and enough rows that the grid still fills the chip several times
  p.tall = 0;
  {
    const bool fixed_epi = p.alpha == 1.0f &&
                           ((p.bias_mode == AVS_BIAS_NONE && p.act == AVS_ACT_NONE) ||
                            (p.bias_mode == AVS_BIAS_COL && p.act == AVS_ACT_RELU && !p.stat_part));
    const bool can = (dtype == AVS_BF16 || dtype == AVS_F32_SPLIT || dtype == AVS_F16X2) && fixed_epi && g_pipe3 &&
                     (long long)p.K * es > 128 && batch == 1;
    const long long tall_tiles = ((long long)p.M + 255) / 256 * p.tiles_n;
    const int tile_mode = p.variant & 3;   // AVS_TILE_AUTO: by rule; AVS_TILE_128: never; AVS_TILE_256: wherever it exists
    AVS_REQUIRE(tile_mode != AVS_TILE_224 || p.tile_rows, AVS_E_UNSUPPORTED,
                "%s: AVS_TILE_224 is a tile of the tile-local BatchNorm form (avs_conv2d_nhwc_bnlocal)", who);
    if (can && (tile_mode == AVS_TILE_256 || (tile_mode == AVS_TILE_AUTO && tall_tiles >= g_tall_min_tiles &&
                                              (narrow || (long long)p.K * es >= g_tall_min_k_bytes))))
      p.tall = 1;
    if (wide96) p.tall = 1;
  }
  // EPI_BNLOCAL (validated by bnlocal_plan): 256-row tiles at a pitch of tile_rows; EPI_AFFINE: 256-row tiles, or 128-row
  // ones for wide outputs when the caller asks (AVS_TILE_128) or the reduction is short
  // (short reductions, K <= 128: three 128-row workgroups per CU overlap their loops and their epilogue traffic better
  //  than two 256-row ones - l2.conv3 +5 %, l1.conv3 +2 %, bit-identical outputs)
  if (p.tile_rows) p.tall = 1;
  // AVS_F16P8 operands: the input of the 1x1 convolution + statistics form on the 256-row tiles (A fragments in registers),
  // the output / residual of the given-affine form
  AVS_REQUIRE(!(p.y_p8 || p.res_p8) || (p.affine && p.N % 16 == 0 && p.ldc % 16 == 0 && p.ldr % 16 == 0), AVS_E_UNSUPPORTED,
              "%s: an AVS_F16P8 output / residual is taken by avs_conv2d_nhwc_affine (cout and row strides in multiples of 16)", who);
  if (p.x_p8) {
    AVS_REQUIRE(p.stat_part && !p.tile_rows && !p.affine && batch == 1 && g_pipe3 && !(p.variant & AVS_STAGING_GENERIC) &&
                    (p.tall || (long long)p.K * 4 <= g_rowb_threshold_bytes) &&   // (64-byte reduction steps)
                    p.KW == 1 && p.K == p.cin && p.K % 32 == 0 && p.K >= 64 &&
                    p.sh == 1 && p.sw == 1 && p.ph == 0 && p.pw == 0 && p.x_row_stride == (long long)p.Wo * p.x_px_stride &&
                    p.x_img_stride == (long long)p.HoWo * p.x_px_stride && p.x_px_stride % 16 == 0 &&
                    256 * p.x_px_stride * 3 + (long long)p.K * 3 < (1ll << 31),
                AVS_E_UNSUPPORTED,
                "%s: an AVS_F16P8 input is taken by avs_conv2d_nhwc_bnstats for 1x1 / stride-1 convolutions on dense rows, "
                "cin a multiple of 32 (>= 64), the input row stride a multiple of 16", who);
  }
  if (p.affine) {
    const int tile_mode = p.variant & 3;
    p.tall = (!narrow && (tile_mode == AVS_TILE_128 || (tile_mode == AVS_TILE_AUTO && p.K <= AVS_RULE_AFFINE_128_MAX_K))) ? 0 : 1;
    // the 128-row form exists on 64-byte reduction steps only: a longer reduction keeps the 256-row tile whatever the caller
    // asks for (the variant is a tuning hint: results do not depend on it)
    if (!p.tall && (long long)p.K * 4 > g_rowb_threshold_bytes) p.tall = 1;
  }
  const int tile_rows = p.tile_rows ? p.tile_rows : (p.tall ? 256 : 128);
  const long long tiles_m = ((long long)p.M + tile_rows - 1) / tile_rows;
  const long long total = tiles_m * p.tiles_n;
  AVS_REQUIRE(total < (1ll << 31), AVS_E_SHAPE, "%s: too many tiles", who);
  if (tiles_m_out) *tiles_m_out = tiles_m;
  if (plan_only) return AVS_OK;
  // the per-tap bounds tests are only needed when a tap can leave the image
  const bool spatial = !(p.ph == 0 && p.pw == 0 && (p.HoWo / p.Wo - 1) * p.sh + (p.K / (p.cin * p.KW)) - 1 < p.H &&
                         (p.Wo - 1) * p.sw + p.KW - 1 < p.W);
  // dense 1x1 / stride-1 input (and every plain GEMM): output row m reads input row m
  p.lin_stride = -1;
  if (!spatial && p.KW == 1 && p.K == p.cin && p.sh == 1 && p.sw == 1) {
    if (p.HoWo == 1)
      p.lin_stride = p.x_img_stride;
    else if (p.x_row_stride == (long long)p.Wo * p.x_px_stride && p.x_img_stride == (long long)p.HoWo * p.x_px_stride)
      p.lin_stride = p.x_px_stride;
  }
  dim3 grid((unsigned)total, 1, (unsigned)batch);
  if (p.tile_rows && dtype == AVS_F16X2) {
    // a group of 193 .. 224 rows: the tile that fits it (local224.hip), unless the caller asks for the 256-row form
    const bool fits = igemm_h2_local224_ok(p, dtype);
    AVS_REQUIRE(fits || p.cluster <= 1, AVS_E_UNSUPPORTED, "%s: the clustered form runs on the 224-row tile only", who);
    AVS_REQUIRE(fits || (p.variant & 3) != AVS_TILE_224, AVS_E_UNSUPPORTED,
                "%s: AVS_TILE_224 takes AVS_F16X2, groups of 193..224 rows, cout in multiples of 128, cin in multiples of 16", who);
    if (fits) {
      igemm_h2_local224_launch(p, spatial, grid, stream);
      AVS_CHECK_LAUNCH(who);
      return AVS_OK;
    }
  }
  bool launched;
  if (wide96) {
    launched = igemm_dispatch_brelu96(spatial, grid, stream, p);
  } else if (dtype == AVS_BF16) {
    launched = narrow ? igemm_dispatch<2, 64, false>(spatial, grid, stream, p) : igemm_dispatch<2, 128, false>(spatial, grid, stream, p);
  } else if (dtype == AVS_F32_ACC64) {
    launched = igemm_dispatch<4, 64, true>(spatial, grid, stream, p);
  } else {
    launched = narrow ? igemm_dispatch<4, 64, false>(spatial, grid, stream, p) : igemm_dispatch<4, 128, false>(spatial, grid, stream, p);
  }
  AVS_REQUIRE(launched, AVS_E_UNSUPPORTED,
              "%s: no kernel for this combination (dtype %d, %s tile, K = %d, N = %d, variant %d, %s%s%s): nothing was launched", who,
              dtype, p.tall ? "256-row" : "128-row", p.K, p.N, p.variant, p.affine ? "given-affine " : "", p.tile_rows ? "tile-local " : "",
              p.stat_part ? "statistics" : "");
  AVS_CHECK_LAUNCH(who);
  return AVS_OK;
}

static int conv_fill_params(const avs_conv_desc* d, const void* d_x, const void* d_w, const float* d_bias, void* d_y,
                            IgemmParams& p, const char* who) {
  AVS_REQUIRE(d != nullptr, AVS_E_ARG, "%s: null descriptor", who);
  AVS_REQUIRE(d->n >= 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->kh > 0 && d->kw > 0 && d->sh > 0 && d->sw > 0 &&
                  d->ph >= 0 && d->pw >= 0 && d->ho > 0 && d->wo > 0 && d->cout > 0,
              AVS_E_SHAPE, "%s: non-positive extent", who);
  // every output tap row/pixel must stay inside the padded input
  AVS_REQUIRE((d->ho - 1) * d->sh - d->ph + d->kh - 1 < d->h + d->ph &&
                  (d->wo - 1) * d->sw - d->pw + d->kw - 1 < d->w + d->pw,
              AVS_E_SHAPE, "%s: output extent %dx%d exceeds what input %dx%d allows", who, d->ho, d->wo, d->h, d->w);
  const long long rows = (long long)d->n * d->ho * d->wo;
  AVS_REQUIRE(rows < (1ll << 31), AVS_E_SHAPE, "%s: %lld output pixels exceed int32", who, rows);
  p.x = (const char*)d_x;
  p.w = (const char*)d_w;
  p.y = (char*)d_y;
  p.bias = d_bias;
  p.M = (int)rows;
  p.N = d->cout;
  p.K = d->kh * d->kw * d->cin;
  p.HoWo = d->ho * d->wo;
  p.Wo = d->wo;
  p.H = d->h;
  p.W = d->w;
  p.cin = d->cin;
  p.KW = d->kw;
  p.sh = d->sh;
  p.sw = d->sw;
  p.ph = d->ph;
  p.pw = d->pw;
  p.x_img_stride = d->x_img_stride;
  p.x_row_stride = d->x_row_stride;
  p.x_px_stride = d->x_px_stride;
  p.ldb = d->w_row_stride;
  AVS_REQUIRE(d->w_layout == AVS_W_ROWS || d->w_layout == AVS_W_KSTEP32, AVS_E_ARG, "%s: bad w_layout %d", who,
              d->w_layout);
  p.w_kstep = d->w_layout == AVS_W_KSTEP32 ? 1 : 0;
  AVS_REQUIRE((d->variant & ~7) == 0, AVS_E_ARG, "%s: bad variant %d", who, d->variant);
  p.variant = d->variant;
  AVS_REQUIRE((d->formats & ~(AVS_X_F16P8 | AVS_Y_F16P8 | AVS_RES_F16P8)) == 0, AVS_E_ARG, "%s: bad formats %d", who, d->formats);
  AVS_REQUIRE(d->formats == 0 || d->dtype == AVS_F16X2, AVS_E_UNSUPPORTED, "%s: AVS_F16P8 operands belong to the AVS_F16X2 forms", who);
  p.x_p8 = (d->formats & AVS_X_F16P8) ? 1 : 0;
  p.y_p8 = (d->formats & AVS_Y_F16P8) ? 1 : 0;
  p.res_p8 = (d->formats & AVS_RES_F16P8) ? 1 : 0;
  AVS_REQUIRE(!p.w_kstep || (d->dtype != AVS_F32_ACC64 && p.K % (d->dtype == AVS_BF16 ? 32 : 16) == 0), AVS_E_UNSUPPORTED,
              "%s: the reduction-step-major weight layout needs a reduction that is a multiple of a 64-byte step (K = %d)",
              who, p.K);
  p.ldc = d->y_px_stride;
  p.alpha = d->alpha;
  p.act = d->act;
  p.bias_mode = d_bias ? AVS_BIAS_COL : AVS_BIAS_NONE;
  AVS_REQUIRE(p.ldb >= p.K, AVS_E_SHAPE, "%s: w_row_stride %lld < kh*kw*cin=%d", who, p.ldb, p.K);
  return AVS_OK;
}

extern "C" int avs_conv2d_nhwc(const avs_conv_desc* d, const void* d_x, const void* d_w, const float* d_bias,
                               void* d_y, avs_stream_t stream) {
  IgemmParams p{};
  int st = conv_fill_params(d, d_x, d_w, d_bias, d_y, p, "avs_conv2d_nhwc");
  if (st != AVS_OK) return st;
  return igemm_launch(d->dtype, p, 1, (hipStream_t)stream, "avs_conv2d_nhwc");
}

// Several convolutions that read the SAME input as one contraction: their filters stacked into one weight matrix, output
// columns [0, n_split) to d_y and [n_split, cout) to d_y2 - e.g. an Inception block's 1x1 heads, of which one writes its
// slice of the block's concatenated output and the others feed further convolutions (features/extractors.py:26,73-90).
extern "C" int avs_conv2d_nhwc_split(const avs_conv_desc* d, const void* d_x, const void* d_w, const float* d_bias,
                                     void* d_y, int n_split, void* d_y2, int64_t y2_px_stride, int relu_cols,
                                     avs_stream_t stream) {
  const char* who = "avs_conv2d_nhwc_split";
  IgemmParams p{};
  int st = conv_fill_params(d, d_x, d_w, d_bias, d_y, p, who);
  if (st != AVS_OK) return st;
  AVS_REQUIRE(d->dtype == AVS_F16X2, AVS_E_UNSUPPORTED, "%s: built for AVS_F16X2", who);
  AVS_REQUIRE(n_split > 0 && n_split < p.N && n_split % 8 == 0 && n_split <= p.ldc && y2_px_stride >= p.N - n_split &&
                  y2_px_stride % 8 == 0,
              AVS_E_SHAPE, "%s: 0 < n_split < cout in multiples of 8 slots, both destinations wide enough (n_split = %d)", who, n_split);
  AVS_REQUIRE(d_y2 && (((uintptr_t)d_y2) & 31u) == 0, AVS_E_ALIGN, "%s: the second destination must be 32-byte aligned", who);
  p.y2 = (char*)d_y2;
  p.ldc2 = y2_px_stride;
  p.nsplit = n_split;
  AVS_REQUIRE(relu_cols >= 0 && relu_cols % 8 == 0 && relu_cols <= p.N, AVS_E_SHAPE, "%s: relu_cols in multiples of 8, at most cout", who);
  p.relu_cols = relu_cols;
  return igemm_launch(d->dtype, p, 1, (hipStream_t)stream, who);
}

// ---- convolution + deterministic BatchNorm batch statistics (EPI_STATS) ----
// One thread per (group, channel): the group's slots are added in row-tile order, then folded into the affine
//   scale = gamma / sqrt(var + eps), shift = beta - mean * scale   (biased variance, E[y^2] - E[y]^2 on fp32 sums).
// chan = 1 (AVS_F16X2): a slot holds (sum, sum of squares about the TILE's mean of the group's rows it holds); the tiles
// of a group are merged by Chan's update in tile order - no E[y^2] - E[y]^2 anywhere.
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ part, long long total, int c, int slots,
                                                      int tile_rows, long long rpg, long long rows,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                      int chan) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long g = i / c;
    const int ch = (int)(i - g * c);
    const long long first = g * rpg, last = (first + rpg < rows ? first + rpg : rows) - 1;
    float s1 = 0.f, s2 = 0.f;
    if (chan) {
      float n = 0.f, mean = 0.f, m2 = 0.f;
      for (long long tl = first / tile_rows; tl <= last / tile_rows; ++tl) {
        const long long j = g - (tl * tile_rows) / rpg;
        const float* src = part + ((tl * slots + j) * 2) * c + ch;
        const long long lo = first > tl * tile_rows ? first : tl * tile_rows;
        const long long hi = last + 1 < (tl + 1) * tile_rows ? last + 1 : (tl + 1) * tile_rows;
        const float nt = (float)(hi - lo);
        const float mt = src[0] / nt;
        const float d = mt - mean, nn = n + nt;
        mean = mean + d * (nt / nn);
        m2 = m2 + src[c] + d * d * (n * nt / nn);
        n = nn;
      }
      const float sc = gamma[ch] / sqrtf(m2 / n + eps);
      scale[i] = sc;
      shift[i] = beta[ch] - mean * sc;
      continue;
    }
    for (long long tl = first / tile_rows; tl <= last / tile_rows; ++tl) {
      const long long j = g - (tl * tile_rows) / rpg;
      const float* src = part + ((tl * slots + j) * 2) * c + ch;
      s1 += src[0];
      s2 += src[c];
    }
    const float inv_n = 1.f / (float)(last - first + 1);
    const float mean = s1 * inv_n;
    const float var = fmaxf(s2 * inv_n - mean * mean, 0.f);
    const float sc = gamma[ch] / sqrtf(var + eps);
    scale[i] = sc;
    shift[i] = beta[ch] - mean * sc;
  }
}

static int bnstats_plan(const avs_conv_desc* d, int64_t rpg, IgemmParams& p, int64_t* ws_bytes, int* tile_rows,
                        const char* who) {
  AVS_REQUIRE(d != nullptr && (d->dtype == AVS_BF16 || d->dtype == AVS_F32 || d->dtype == AVS_F32_SPLIT ||
                               d->dtype == AVS_F16X2), AVS_E_ARG, "%s: bf16 / fp32 / f16x2 only", who);
  AVS_REQUIRE(d->act == AVS_ACT_NONE && d->alpha == 1.0f, AVS_E_ARG, "%s: no activation / scaling", who);
  AVS_REQUIRE(rpg > 0 && rpg < (1ll << 30), AVS_E_ARG, "%s: rows_per_group must be positive", who);
  AVS_REQUIRE(rpg >= STATS_MIN_GROUP_ROWS, AVS_E_UNSUPPORTED,
              "%s: groups of fewer than %d rows take the separate statistics pass (avs_bn_batch_stats)", who,
              STATS_MIN_GROUP_ROWS);
  long long tiles_m = 0;
  p.stat_part = reinterpret_cast<float*>(16);  // selects EPI_STATS in the plan (same tile choice as the launch)
  const int st = igemm_launch(d->dtype, p, 1, nullptr, who, true, &tiles_m);
  p.stat_part = nullptr;
  if (st != AVS_OK) return st;
  *tile_rows = p.tall ? 256 : 128;
  p.stat_slots = (int)((*tile_rows + rpg - 2) / rpg + 1);
  *ws_bytes = (int64_t)((tiles_m * p.stat_slots * 2 * p.N * 4 + 255) / 256 * 256);
  return AVS_OK;
}

extern "C" int64_t avs_conv2d_bnstats_workspace_bytes(const avs_conv_desc* d, int64_t rows_per_group) {
  const char* who = "avs_conv2d_bnstats_workspace_bytes";
  IgemmParams p{};
  int st = conv_fill_params(d, (const void*)16, (const void*)16, nullptr, (void*)16, p, who);
  if (st != AVS_OK) return st;
  if (p.M == 0) return 0;
  int64_t ws = 0;
  int tile_rows = 0;
  st = bnstats_plan(d, rows_per_group, p, &ws, &tile_rows, who);
  return st == AVS_OK ? ws : (int64_t)st;
}

extern "C" int avs_conv2d_nhwc_bnstats(const avs_conv_desc* d, const void* d_x, const void* d_w, void* d_y,
                                       int64_t rows_per_group, const float* d_gamma, const float* d_beta, float eps,
                                       float* d_scale, float* d_shift, void* d_ws, int64_t ws_bytes,
                                       avs_stream_t stream) {
  const char* who = "avs_conv2d_nhwc_bnstats";
  IgemmParams p{};
  int st = conv_fill_params(d, d_x, d_w, nullptr, d_y, p, who);
  if (st != AVS_OK) return st;
  if (p.M == 0) return AVS_OK;
  int64_t need = 0;
  int tile_rows = 0;
  st = bnstats_plan(d, rows_per_group, p, &need, &tile_rows, who);
  if (st != AVS_OK) return st;
  AVS_REQUIRE(d_gamma && d_beta && d_scale && d_shift && d_ws, AVS_E_ARG, "%s: null pointer", who);
  AVS_REQUIRE(ws_bytes >= need, AVS_E_WORKSPACE, "%s: workspace %lld < %lld bytes", who, (long long)ws_bytes,
              (long long)need);
  AVS_REQUIRE(avs_aligned16(d_ws), AVS_E_ALIGN, "%s: workspace must be 16-byte aligned", who);
  p.stat_part = reinterpret_cast<float*>(d_ws);
  p.rows_per_group = (int)rows_per_group;
  st = igemm_launch(d->dtype, p, 1, (hipStream_t)stream, who);
  if (st != AVS_OK) return st;
  AVS_REQUIRE((p.tall ? 256 : 128) == tile_rows, AVS_E_ARG, "%s: tile choice changed between plan and launch", who);
  const long long groups = ((long long)p.M + rows_per_group - 1) / rows_per_group;
  const long long total = groups * p.N;
  long long gx = avs_cdiv(total, 256);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, p.stat_part, total, p.N,
                     p.stat_slots, tile_rows, (long long)rows_per_group, (long long)p.M, d_gamma, d_beta, eps, d_scale,
                     d_shift, d->dtype == AVS_F16X2 ? 1 : 0);
  AVS_CHECK_LAUNCH(who);
  return AVS_OK;
}

// ---- convolution + whole BatchNorm in one launch, statistics local to a tile (EPI_BNLOCAL) ----

static int bnlocal_plan(const avs_conv_desc* d, int64_t rpg, IgemmParams& p, const char* who) {
  int st = conv_fill_params(d, (const void*)16, (const void*)16, nullptr, (void*)16, p, who);
  if (st != AVS_OK) return st;
  AVS_REQUIRE(rpg > 0, AVS_E_ARG, "%s: rows_per_group must be positive", who);
  if (p.M == 0) return AVS_OK;
  const int bn = p.N <= 64 ? 64 : 128;
  const long long per_tile = 256 / rpg;
  const bool h2 = d->dtype == AVS_F16X2;
  const int es = h2 ? 4 : 2;
  const bool ok = (g_bnlocal != 0) && (g_pipe3 != 0) && (d->dtype == AVS_BF16 || h2) && p.N % bn == 0 && p.M % rpg == 0 && rpg <= 256 &&
                  per_tile * rpg * 4 >= 256 * 3 && per_tile <= BNLOCAL_MAX_GROUPS && (long long)p.K * es > 128 &&
                  d->alpha == 1.0f && (p.ldc * es) % (h2 ? 32 : 16) == 0;
  AVS_REQUIRE(ok, AVS_E_UNSUPPORTED,
              "%s: needs bf16 / f16x2, cout a multiple of %d, equal groups of 43..256 rows that fill 3/4 of a 256-row "
              "tile, a reduction of more than 128 bytes, aligned output rows", who, bn);
  p.tiles_n = p.N / bn;
  p.tile_rows = (int)(per_tile * rpg);
  return AVS_OK;
}

// ---- clustered tile-local BatchNorm (AVS_F16X2, the 224-row kernel): a group of `cluster` frames whose maps fill one
// 224-row tile each (14 x 14: the reference's 4-frame micro-batches at ResNet-50's layer 3, features/extractors.py:48) ----
static int bncluster_plan(const avs_conv_desc* d, int64_t rpg, int cluster, IgemmParams& p, const char* who) {
  int st = conv_fill_params(d, (const void*)16, (const void*)16, nullptr, (void*)16, p, who);
  if (st != AVS_OK) return st;
  AVS_REQUIRE(rpg > 0 && cluster >= 2 && cluster <= 16, AVS_E_ARG, "%s: rows_per_group > 0 and 2 <= cluster <= 16", who);
  if (p.M == 0) return AVS_OK;
  const bool ok = (g_bnlocal != 0) && (g_pipe3 != 0) && d->dtype == AVS_F16X2 && rpg % cluster == 0 && p.M % rpg == 0 &&
                  rpg / cluster > 192 && rpg / cluster <= 224 && p.N % 128 == 0 && (long long)p.K * 4 > 128 &&
                  d->alpha == 1.0f && (p.ldc * 4) % 32 == 0 && (d->variant & AVS_STAGING_GENERIC) == 0;
  AVS_REQUIRE(ok, AVS_E_UNSUPPORTED,
              "%s: needs AVS_F16X2, equal groups of `cluster` tiles of 193..224 rows each, cout a multiple of 128, a "
              "reduction of more than 128 bytes, aligned output rows", who);
  p.tiles_n = p.N / 128;
  p.tile_rows = (int)(rpg / cluster);
  p.rows_per_group = (int)rpg;
  p.cluster = cluster;
  return AVS_OK;
}

extern "C" int64_t avs_conv2d_bncluster_workspace_bytes(const avs_conv_desc* d, int64_t rows_per_group, int cluster) {
  IgemmParams p{};
  const int st = bncluster_plan(d, rows_per_group, cluster, p, "avs_conv2d_bncluster_workspace_bytes");
  if (st != AVS_OK) return st;
  if (p.M == 0) return 64;
  // a 64-byte header (the error counter) + 8-byte granules [tiles_m][tiles_n][4 waves][64 lanes]
  return 64 + (int64_t)(p.M / p.tile_rows) * p.tiles_n * 4 * 64 * 8;
}

extern "C" int avs_conv2d_nhwc_bncluster(const avs_conv_desc* d, const void* d_x, const void* d_w, void* d_y,
                                         int64_t rows_per_group, int cluster, const float* d_gamma, const float* d_beta,
                                         float eps, const void* d_residual, int64_t ldr, void* d_xchg, int64_t xchg_bytes,
                                         uint32_t epoch, avs_stream_t stream) {
  const char* who = "avs_conv2d_nhwc_bncluster";
  IgemmParams p{};
  int st = bncluster_plan(d, rows_per_group, cluster, p, who);
  if (st != AVS_OK) return st;
  if (p.M == 0) return AVS_OK;
  AVS_REQUIRE(d_x && d_w && d_y && d_gamma && d_beta && d_xchg, AVS_E_ARG, "%s: null pointer", who);
  AVS_REQUIRE(epoch != 0, AVS_E_ARG, "%s: epoch 0 is the exchange buffer's initial state", who);
  AVS_REQUIRE((((uintptr_t)d_y) & 31u) == 0 && (((uintptr_t)d_xchg) & 63u) == 0, AVS_E_ALIGN,
              "%s: y must be 32-byte, the exchange buffer 64-byte aligned", who);
  AVS_REQUIRE(!d_residual || ((((uintptr_t)d_residual) & 31u) == 0 && ldr % 8 == 0 && ldr >= p.N), AVS_E_ALIGN,
              "%s: an AVS_F16X2 residual must be 32-byte aligned with a row stride in multiples of 8 slots, at least cout long", who);
  const int64_t need = 64 + (int64_t)(p.M / p.tile_rows) * p.tiles_n * 4 * 64 * 8;
  AVS_REQUIRE(xchg_bytes >= need, AVS_E_WORKSPACE, "%s: exchange buffer %lld < %lld bytes", who, (long long)xchg_bytes,
              (long long)need);
  p.x = (const char*)d_x;
  p.w = (const char*)d_w;
  p.y = (char*)d_y;
  p.gamma = d_gamma;
  p.beta = d_beta;
  p.eps = eps;
  p.residual = (const char*)d_residual;
  p.ldr = ldr;
  p.bias_mode = AVS_BIAS_NONE;
  p.epoch = epoch;
  p.xerr = reinterpret_cast<unsigned*>(d_xchg);
  p.xchg = reinterpret_cast<unsigned long long*>((char*)d_xchg + 64);
  return igemm_launch(d->dtype, p, 1, (hipStream_t)stream, who);
}

extern "C" int avs_conv2d_bnlocal_tile_rows(const avs_conv_desc* d, int64_t rows_per_group) {
  IgemmParams p{};
  const int st = bnlocal_plan(d, rows_per_group, p, "avs_conv2d_bnlocal_tile_rows");
  return st == AVS_OK ? p.tile_rows : st;
}

extern "C" int avs_conv2d_nhwc_bnlocal(const avs_conv_desc* d, const void* d_x, const void* d_w, void* d_y,
                                       int64_t rows_per_group, const float* d_gamma, const float* d_beta, float eps,
                                       const void* d_residual, int64_t ldr, avs_stream_t stream) {
  const char* who = "avs_conv2d_nhwc_bnlocal";
  IgemmParams p{};
  int st = bnlocal_plan(d, rows_per_group, p, who);
  if (st != AVS_OK) return st;
  if (p.M == 0) return AVS_OK;
  AVS_REQUIRE(d_x && d_w && d_y && d_gamma && d_beta, AVS_E_ARG, "%s: null pointer", who);
  AVS_REQUIRE(avs_aligned16(d_y), AVS_E_ALIGN, "%s: y must be 16-byte aligned", who);
  const bool h2 = d->dtype == AVS_F16X2;
  AVS_REQUIRE(!d_residual || (avs_aligned16(d_residual) && (ldr * 2) % 16 == 0 && ldr >= p.N), AVS_E_ALIGN,
              "%s: residual rows must be 16-byte aligned and at least cout long", who);
  AVS_REQUIRE(!h2 || !d_residual || ((((uintptr_t)d_residual) & 31u) == 0 && ldr % 8 == 0), AVS_E_ALIGN,
              "%s: an AVS_F16X2 residual must be 32-byte aligned with a row stride in multiples of 8 slots", who);
  p.x = (const char*)d_x;
  p.w = (const char*)d_w;
  p.y = (char*)d_y;
  p.rows_per_group = (int)rows_per_group;
  p.gamma = d_gamma;
  p.beta = d_beta;
  p.eps = eps;
  p.residual = (const char*)d_residual;
  p.ldr = ldr;
  p.bias_mode = AVS_BIAS_NONE;
  return igemm_launch(d->dtype, p, 1, (hipStream_t)stream, who);
}

// ---- 1x1 convolution with a GIVEN per-group affine (+ residual, + its affine, + ReLU): one streaming pass (EPI_AFFINE) ----
extern "C" int avs_conv2d_nhwc_affine(const avs_conv_desc* d, const void* d_x, const void* d_w, void* d_y,
                                      int64_t rows_per_group, const float* d_scale, const float* d_shift,
                                      const void* d_residual, int64_t ldr, const float* d_res_scale,
                                      const float* d_res_shift, avs_stream_t stream) {
  const char* who = "avs_conv2d_nhwc_affine";
  IgemmParams p{};
  int st = conv_fill_params(d, d_x, d_w, nullptr, d_y, p, who);
  if (st != AVS_OK) return st;
  if (p.M == 0) return AVS_OK;
  AVS_REQUIRE(d->dtype == AVS_F16X2, AVS_E_UNSUPPORTED, "%s: built for AVS_F16X2", who);
  AVS_REQUIRE(d->kh == 1 && d->kw == 1 && d->ph == 0 && d->pw == 0 && (long long)p.K * 4 > 128 && d->alpha == 1.0f,
              AVS_E_UNSUPPORTED, "%s: 1x1 convolutions without padding, more than 32 input channels, alpha = 1", who);
  AVS_REQUIRE(rows_per_group >= 32 && rows_per_group < (1ll << 30), AVS_E_UNSUPPORTED,
              "%s: groups of at least 32 rows (a wave's 64 rows then lie in at most three groups)", who);
  AVS_REQUIRE(d_scale && d_shift && avs_aligned16(d_scale) && avs_aligned16(d_shift), AVS_E_ARG,
              "%s: scale / shift [groups, cout] must be given, 16-byte aligned", who);
  AVS_REQUIRE((d_res_scale == nullptr) == (d_res_shift == nullptr) && (!d_res_scale || d_residual), AVS_E_ARG,
              "%s: residual scale and shift go together, with a residual", who);
  AVS_REQUIRE(!d_res_scale || (avs_aligned16(d_res_scale) && avs_aligned16(d_res_shift)), AVS_E_ALIGN,
              "%s: residual scale / shift must be 16-byte aligned", who);
  AVS_REQUIRE(!d_residual || ((((uintptr_t)d_residual) & 31u) == 0 && ldr % 8 == 0 && ldr >= p.N), AVS_E_ALIGN,
              "%s: the residual must be 32-byte aligned with a row stride in multiples of 8 slots, at least cout", who);
  AVS_REQUIRE(g_pipe3, AVS_E_UNSUPPORTED, "%s: needs the pipelined tile variants", who);
  p.affine = 1;
  p.rows_per_group = (int)rows_per_group;
  p.gamma = d_scale;
  p.beta = d_shift;
  p.residual = (const char*)d_residual;
  p.ldr = ldr;
  p.res_scale = d_res_scale;
  p.res_shift = d_res_shift;
  p.bias_mode = AVS_BIAS_NONE;
  st = igemm_launch(d->dtype, p, 1, (hipStream_t)stream, who);
  return st;
}

extern "C" int avs_gemm_nt(int dtype, int m, int n, int k, const void* d_a, int64_t lda, int64_t stride_a,
                           const void* d_b, int64_t ldb, int64_t stride_b, void* d_c, int64_t ldc, int64_t stride_c,
                           const float* d_bias, int bias_mode, int64_t stride_bias, float alpha, int act, int batch,
                           avs_stream_t stream) {
  IgemmParams p{};
  p.x = (const char*)d_a;
  p.w = (const char*)d_b;
  p.y = (char*)d_c;
  p.bias = d_bias;
  p.M = m;
  p.N = n;
  p.K = k;
  p.HoWo = 1;
  p.Wo = 1;
  p.H = 1;
  p.W = 1;
  p.cin = k;
  p.KW = 1;
  p.sh = p.sw = 1;
  p.ph = p.pw = 0;
  p.x_img_stride = lda;
  p.x_row_stride = 0;
  p.x_px_stride = 0;
  p.ldb = ldb;
  p.ldc = ldc;
  p.sA = stride_a;
  p.sB = stride_b;
  p.sC = stride_c;
  p.sBias = stride_bias;
  p.alpha = alpha;
  p.act = act;
  p.bias_mode = bias_mode;
  AVS_REQUIRE(bias_mode >= AVS_BIAS_NONE && bias_mode <= AVS_BIAS_ROW, AVS_E_ARG, "avs_gemm_nt: bad bias mode %d",
              bias_mode);
  AVS_REQUIRE(lda >= 0 && ldb >= k, AVS_E_SHAPE, "avs_gemm_nt: lda=%lld ldb=%lld k=%d", (long long)lda,
              (long long)ldb, k);
  return igemm_launch(dtype, p, batch, (hipStream_t)stream, "avs_gemm_nt");
}
