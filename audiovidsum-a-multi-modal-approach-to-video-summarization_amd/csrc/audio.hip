// Audio front-end kernels: reflect pad, the STFT as a dense real DFT on the
// fp64 matrix cores (frames read in place, row stride = hop), and the bandwidth-bound tail
// |X|^2 -> mel -> log / dB.
#include "avs_internal.h"
#include <math.h>

__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ x, long long t, int pad,
                                                          float* __restrict__ out, long long out_len) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < out_len;
       i += (long long)gridDim.x * blockDim.x) {
    long long j = i - pad;
    float v = 0.f;
    if (i < t + 2ll * pad) {
      if (j < 0) j = -j;
      if (j >= t) j = 2 * (t - 1) - j;
      v = x[j];
    }
    out[i] = v;
  }
}

extern "C" int avs_reflect_pad_f32(const float* d_x, int64_t t, int pad, float* d_out, int64_t out_len,
                                   avs_stream_t stream) {
  AVS_REQUIRE(t > pad && pad >= 0 && out_len >= t + 2ll * pad, AVS_E_SHAPE,
              "avs_reflect_pad_f32: need t > pad and out_len >= t + 2*pad (t=%lld pad=%d out_len=%lld)", (long long)t,
              pad, (long long)out_len);
  AVS_REQUIRE(d_x && d_out, AVS_E_ARG, "avs_reflect_pad_f32: null pointer");
  long long gx = avs_cdiv(out_len, 256);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(reflect_pad_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, d_x, (long long)t, pad,
                     d_out, (long long)out_len);
  AVS_CHECK_LAUNCH("avs_reflect_pad_f32");
  return AVS_OK;
}

// ---------------------------------------------------------------------------
// STFT as a dense real DFT on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).
// Why fp64: log2(mel + 1e-6) is ill-conditioned in the rare bins whose |X|^2 is ~1e-4 of their
// neighbours'; an fp32 accumulation (dense or FFT — torch.stft's included) is 1e-4..2e-3 off
// there.  fp32 samples and the windowed basis are exact in fp64 and so are their products, so
// the spectrum below is the correctly rounded value of the defining formula.
// Frames are read in place from the reflect-padded waveform (row stride = hop).
// Block = 4 waves = 64 frames x 64 basis columns; wave = 16 frames x 4 column tiles.
// Lane l feeds A[row l&15][k l>>4] and B[k l>>4][col l&15]; result register j of a lane is
// row (l>>4) + 4*j, column l&15.
// ---------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stft_f64_kernel(const float* __restrict__ xpad, long long frames, int hop,
                                                       int nfft, const double* __restrict__ basis_t, int ncols,
                                                       int ncols_pad, float* __restrict__ spec) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lr = lane & 15, lq = lane >> 4;
  const long long r0 = (long long)blockIdx.x * 64 + wave * 16;
  const int c0 = blockIdx.y * 64;
  const long long fr = r0 + lr;
  const bool row_ok = fr < frames;
  const float* __restrict__ xrow = xpad + (row_ok ? fr : 0) * hop;
  f64x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < nfft; k0 += 4) {
    const int k = k0 + lq;
    const double a = (row_ok && k < nfft) ? (double)xrow[k] : 0.0;
    const double* __restrict__ brow = basis_t + (long long)(k < nfft ? k : 0) * ncols_pad + c0 + lr;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const double b = k < nfft ? brow[nt * 16] : 0.0;
      acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[nt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int col = c0 + nt * 16 + lr;
    if (col >= ncols) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long row = r0 + lq + 4 * j;
      if (row < frames) spec[row * ncols + col] = (float)acc[nt][j];
    }
  }
}

extern "C" int avs_stft_f64(const float* d_xpad, int64_t xpad_len, int64_t frames, int hop, int nfft,
                            const double* d_basis_t, int ncols, int ncols_pad, float* d_spec, avs_stream_t stream) {
  AVS_REQUIRE(frames >= 0 && hop > 0 && nfft > 0 && ncols > 0 && ncols_pad >= ncols && ncols_pad % 64 == 0,
              AVS_E_SHAPE, "avs_stft_f64: frames=%lld hop=%d nfft=%d ncols=%d ncols_pad=%d", (long long)frames, hop,
              nfft, ncols, ncols_pad);
  if (frames == 0) return AVS_OK;
  AVS_REQUIRE(d_xpad && d_basis_t && d_spec, AVS_E_ARG, "avs_stft_f64: null pointer");
  AVS_REQUIRE((frames - 1) * hop + nfft <= xpad_len, AVS_E_SHAPE,
              "avs_stft_f64: %lld frames of %d samples at hop %d overrun the %lld-sample waveform", (long long)frames,
              nfft, hop, (long long)xpad_len);
  const long long bx = avs_cdiv(frames, 64);
  AVS_REQUIRE(bx < (1ll << 31), AVS_E_SHAPE, "avs_stft_f64: too many frames");
  hipLaunchKernelGGL(stft_f64_kernel, dim3((unsigned)bx, ncols_pad / 64), dim3(256), 0, (hipStream_t)stream, d_xpad,
                     (long long)frames, hop, nfft, d_basis_t, ncols, ncols_pad, d_spec);
  AVS_CHECK_LAUNCH("avs_stft_f64");
  return AVS_OK;
}

// One block = FPB STFT frames.  Power spectrum staged in LDS, then each thread
// owns (frame, mel) pairs and sums its filter's non-zero bins in ascending order.
#define AVS_MEL_FPB 8
#define AVS_MEL_MAXBINS 1025

__global__ __launch_bounds__(256) void power_mel_kernel(const float* __restrict__ spec, long long frames, int nbins,
                                                        const float* __restrict__ fb, const int* __restrict__ fb_lo,
                                                        const int* __restrict__ fb_hi, int nmel, int mode,
                                                        float* __restrict__ out, float* __restrict__ gmax) {
  extern __shared__ float pw[];  // [FPB][nbins]
  const long long f0 = (long long)blockIdx.x * AVS_MEL_FPB;
  const int nf = (int)((frames - f0) < AVS_MEL_FPB ? (frames - f0) : AVS_MEL_FPB);
  for (int i = threadIdx.x; i < nf * nbins; i += blockDim.x) {
    const int f = i / nbins, k = i - f * nbins;
    const float re = spec[(f0 + f) * 2 * nbins + k];
    const float im = spec[(f0 + f) * 2 * nbins + nbins + k];
    const float pwr = re * re + im * im;
    pw[i] = mode == 3 ? sqrtf(pwr) : pwr;  // mode 3 (VGGish): magnitude spectrum
  }
  __syncthreads();
  float lmax = 0.f;
  for (int i = threadIdx.x; i < nf * nmel; i += blockDim.x) {
    const int f = i / nmel, m = i - f * nmel;
    const int lo = fb_lo[m], hi = fb_hi[m];
    float a = 0.f;
    for (int k = lo; k < hi; ++k) a += pw[f * nbins + k] * fb[(long long)k * nmel + m];
    float v;
    if (mode == 0) {
      v = log2f(a + 1e-6f);
    } else if (mode == 1) {
      const float cl = fmaxf(a, 1e-10f);
      lmax = fmaxf(lmax, cl);
      v = 10.f * log10f(cl);
    } else if (mode == 3) {
      v = logf(a + 0.01f);
    } else {
      v = a;
    }
    out[(f0 + f) * nmel + m] = v;
  }
  if (mode == 1) {
    lmax = avs_wave_max(lmax);
    // positive floats order like their bit patterns
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(gmax), __float_as_uint(lmax));
  }
}

extern "C" int avs_power_mel_f32(const float* d_spec, int64_t frames, int nbins, const float* d_fb, const int* d_fb_lo,
                                 const int* d_fb_hi, int nmel, int mode, float* d_out, float* d_max,
                                 avs_stream_t stream) {
  AVS_REQUIRE(frames >= 0 && nbins > 0 && nbins <= AVS_MEL_MAXBINS && nmel > 0 && mode >= 0 && mode <= 3, AVS_E_SHAPE,
              "avs_power_mel_f32: frames=%lld nbins=%d nmel=%d mode=%d", (long long)frames, nbins, nmel, mode);
  if (frames == 0) return AVS_OK;
  AVS_REQUIRE(d_spec && d_fb && d_fb_lo && d_fb_hi && d_out, AVS_E_ARG, "avs_power_mel_f32: null pointer");
  AVS_REQUIRE(mode != 1 || d_max, AVS_E_ARG, "avs_power_mel_f32: mode 1 needs d_max");
  const long long blocks = avs_cdiv(frames, AVS_MEL_FPB);
  AVS_REQUIRE(blocks < (1ll << 31), AVS_E_SHAPE, "avs_power_mel_f32: too many frames");
  const size_t shmem = (size_t)AVS_MEL_FPB * nbins * sizeof(float);
  hipLaunchKernelGGL(power_mel_kernel, dim3((unsigned)blocks), dim3(256), shmem, (hipStream_t)stream, d_spec,
                     (long long)frames, nbins, d_fb, d_fb_lo, d_fb_hi, nmel, mode, d_out, d_max);
  AVS_CHECK_LAUNCH("avs_power_mel_f32");
  return AVS_OK;
}

// ---------------------------------------------------------------------------
// Fused front end of torchaudio's MelSpectrogram / MFCC defaults (n_fft = win = 400, hop 200, center / reflect,
// power 2; features/extractors.py:236-246): waveform -> log2(mel + 1e-6) and / or 10 log10(mel) in ONE kernel.
//   * a block owns 32 STFT frames: its waveform span (32 * 200 + 200 samples, reflect padding resolved by index) is
//     staged ONCE in LDS with 16-byte loads; no padded copy of the waveform, no spectrum in HBM;
//   * the 400-point real DFT is folded: with e[n] = w[n] x[n] + w[400-n] x[400-n] and o[n] = w[n] x[n] - w[400-n] x[400-n]
//     (fp64: products of fp32 values are exact)   Re X[k] = w0 x0 + (-1)^k w200 x200 + sum_{n=1..199} e[n] cos(2 pi k n / 400),
//     Im X[k] = - sum_{n=1..199} o[n] sin(2 pi k n / 400): half the reduction length of the dense DFT, the same exact
//     arithmetic (v_mfma_f64_16x16x4_f64, tables of cos / -sin in fp64);
//   * |X|^2 (fp32, from the fp32-rounded re / im as before) goes to LDS, the sparse mel sum (<= 2 filters per bin,
//     ascending bin order) and the log run on it, and only the [frames, n_mels] results are written.
// Wave w: frames 16 (w & 1) .. + 15, column tiles 7 (w >> 1) .. of the 13 16-bin tiles (208 >= 201 bins).
// ---------------------------------------------------------------------------
#define AVS_FUSED_FPB 32
#define AVS_FUSED_NFFT 400
#define AVS_FUSED_HOP 200
#define AVS_FUSED_BINS 201
#define AVS_FUSED_COLS 208   // bins padded to 13 tiles of 16
#define AVS_FUSED_RE_ROWS 204  // n = 0 .. 200 padded to a multiple of 4
#define AVS_FUSED_IM_ROWS 200  // n = 1 .. 199 padded to a multiple of 4

// SEG: a workgroup's frames come from a block table (first STFT frame, count <= 32, segment) instead of blockIdx * 32 and
// nothing per frame is written: the block's per-mel SUMS over its frames (log2-mel, and the dB value clamped at
// 10 log10(*max_in) - top_db) go to part_log2 / part_db [block, nmel]; segment_fold_kernel adds a segment's blocks in
// block order and divides by its frame count - the time means of a shot (features/extractors.py:232-246 take means over
// time of the per-frame matrices) without the [frames, nmel] matrices ever reaching HBM.  Deterministic.
template <bool SEG>
__global__ __launch_bounds__(256, 2) void stft_mel_fused_kernel(
    const float* __restrict__ x, long long t, long long frames, const double* __restrict__ window,
    const double* __restrict__ cos_t, const double* __restrict__ sin_t, const float* __restrict__ fb,
    const int* __restrict__ fb_lo, const int* __restrict__ fb_hi, int nmel, float* __restrict__ out_log2,
    float* __restrict__ out_db, float* __restrict__ out_pow, float* __restrict__ gmax, const int* __restrict__ blocks,
    float* __restrict__ part_log2, float* __restrict__ part_db, const float* __restrict__ max_in, float top_db,
    float* __restrict__ db_rows, const long long* __restrict__ track_off, const long long* __restrict__ track_len) {
  // SEG batch mode (track_off != nullptr): block rows are (first frame, frames, segment, track); the block's waveform is
  // x + track_off[track] (track_len[track] samples), its maximum gmax[track] - every track of a batch in ONE launch
  const int bstride = (SEG && track_off) ? 4 : 3;
  if (SEG && track_off) {
    const int trk = blocks[4 * blockIdx.x + 3];
    x += track_off[trk];
    t = track_len[trk];
    frames = 1 + t / AVS_FUSED_HOP;
    if (gmax) gmax += trk;
  }
  constexpr int SPAN = AVS_FUSED_FPB * AVS_FUSED_HOP + (AVS_FUSED_NFFT - AVS_FUSED_HOP);   // 6600 samples
  __shared__ __attribute__((aligned(16))) float span[SPAN];
  __shared__ double win[AVS_FUSED_NFFT];
  __shared__ float pw[AVS_FUSED_FPB][AVS_FUSED_BINS + 3];
  const int tid = threadIdx.x;
  long long f0 = SEG ? (long long)blocks[bstride * blockIdx.x] : (long long)blockIdx.x * AVS_FUSED_FPB;
  if (SEG) f0 = f0 < 0 ? 0 : (f0 >= frames ? frames - 1 : f0);   // device table: clamped into the track
  // ---- stage the span: padded position q = f0 * 200 + i is sample q - 200, reflected at both ends
  const long long q0 = f0 * AVS_FUSED_HOP - AVS_FUSED_NFFT / 2;
  for (int i = tid * 4; i < SPAN; i += 256 * 4) {
    const long long j = q0 + i;
    float4 v;
    if (j >= 0 && j + 3 < t) {
      v = *reinterpret_cast<const float4*>(x + j);   // q0 and i are multiples of 4, x is 16-byte aligned
    } else {
      float e[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        long long jj = j + u;
        if (jj < 0) jj = -jj;
        if (jj >= t) jj = 2 * (t - 1) - jj;
        e[u] = (jj >= 0 && jj < t) ? x[jj] : 0.f;   // beyond the last frame's window: unused
      }
      v = make_float4(e[0], e[1], e[2], e[3]);
    }
    *reinterpret_cast<float4*>(span + i) = v;
  }
  for (int i = tid; i < AVS_FUSED_NFFT; i += 256) win[i] = window[i];
  __syncthreads();

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lr = lane & 15, lq = lane >> 4;
  const int fg = wave & 1;
  const int tile0 = (wave >> 1) * 7, ntile = (wave >> 1) ? 6 : 7;   // 13 tiles = 7 + 6
  const float* __restrict__ srow = span + (fg * 16 + lr) * AVS_FUSED_HOP;
  f64x4 re[7], im[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    re[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
    im[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  }
  // real part: n = 0 .. 203 (n = 0 and 200 unfolded, 201 .. 203 zero rows of the table)
  for (int k0 = 0; k0 < AVS_FUSED_RE_ROWS; k0 += 4) {
    const int n = k0 + lq;
    double a = 0.0;
    if (n <= 200) {
      a = win[n] * (double)srow[n];
      if (n >= 1 && n <= 199) a += win[AVS_FUSED_NFFT - n] * (double)srow[AVS_FUSED_NFFT - n];
    }
    const double* __restrict__ brow = cos_t + (long long)n * AVS_FUSED_COLS + tile0 * 16 + lr;
#pragma unroll
    for (int i = 0; i < 7; ++i)
      if (i < ntile) re[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, brow[i * 16], re[i], 0, 0, 0);
  }
  // imaginary part: n = 1 .. 200 (the n = 200 row of the table is zero)
  for (int k0 = 0; k0 < AVS_FUSED_IM_ROWS; k0 += 4) {
    const int n = 1 + k0 + lq;
    double a = 0.0;
    if (n <= 199) a = win[n] * (double)srow[n] - win[AVS_FUSED_NFFT - n] * (double)srow[AVS_FUSED_NFFT - n];
    const double* __restrict__ brow = sin_t + (long long)(n - 1) * AVS_FUSED_COLS + tile0 * 16 + lr;
#pragma unroll
    for (int i = 0; i < 7; ++i)
      if (i < ntile) im[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, brow[i * 16], im[i], 0, 0, 0);
  }
  // |X|^2 of the fp32-rounded spectrum: result register j of a lane is frame lq + 4 j, bin tile * 16 + lr
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    if (i >= ntile) continue;
    const int bin = (tile0 + i) * 16 + lr;
    if (bin >= AVS_FUSED_BINS) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float r = (float)re[i][j], m = (float)im[i][j];
      pw[fg * 16 + lq + 4 * j][bin] = r * r + m * m;
    }
  }
  __syncthreads();
  // ---- mel + log: a thread owns (frame, mel) pairs and sums its filter's non-zero bins in ascending order
  if constexpr (SEG) {
    // a thread owns a mel band and walks the block's frames in order: sums of log2(mel + 1e-6) and of the clamped dB.
    // db_rows (one-pass mode): the track's maximum is not known yet - the block's UNCLAMPED dB rows go to the workspace
    // ([block][32][nmel]) and the maximum of the clamped mel power to gmax; segment_db_sum_kernel clamps and sums them
    // (same values, same order: bit-identical to the two-pass form)
    int nf = blocks[bstride * blockIdx.x + 1];
    nf = nf < 0 ? 0 : (nf > AVS_FUSED_FPB ? AVS_FUSED_FPB : nf);   // (the table is device data: never trust it past the tile)
    const float thr = (part_db && !db_rows) ? 10.f * log10f(*max_in) - top_db : 0.f;
    float lmax = 0.f;
    for (int m = tid; m < nmel; m += 256) {
      const int lo = fb_lo[m], hi = fb_hi[m];
      float s_log = 0.f, s_db = 0.f;
      for (int f = 0; f < nf; ++f) {
        float a = 0.f;
        for (int k = lo; k < hi; ++k) a += pw[f][k] * fb[(long long)k * nmel + m];
        s_log += log2f(a + 1e-6f);
        const float cl = fmaxf(a, 1e-10f);
        const float db = 10.f * log10f(cl);
        if (db_rows) {
          lmax = fmaxf(lmax, cl);
          db_rows[((long long)blockIdx.x * AVS_FUSED_FPB + f) * nmel + m] = db;
        } else {
          s_db += fmaxf(db, thr);
        }
      }
      if (part_log2) part_log2[(long long)blockIdx.x * nmel + m] = s_log;
      if (part_db && !db_rows) part_db[(long long)blockIdx.x * nmel + m] = s_db;
    }
    if (db_rows) {
      lmax = avs_wave_max(lmax);
      if ((tid & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(gmax), __float_as_uint(lmax));
    }
    return;
  }
  const int nf = (int)((frames - f0) < AVS_FUSED_FPB ? (frames - f0) : AVS_FUSED_FPB);
  float lmax = 0.f;
  for (int i = tid; i < nf * nmel; i += 256) {
    const int f = i / nmel, m = i - f * nmel;
    const int lo = fb_lo[m], hi = fb_hi[m];
    float a = 0.f;
    for (int k = lo; k < hi; ++k) a += pw[f][k] * fb[(long long)k * nmel + m];
    const long long o = (f0 + f) * nmel + m;
    if (out_log2) out_log2[o] = log2f(a + 1e-6f);
    if (gmax) {   // the dB feature (and / or only the track's maximum, which its top_db clamp needs)
      const float cl = fmaxf(a, 1e-10f);
      lmax = fmaxf(lmax, cl);
      if (out_db) out_db[o] = 10.f * log10f(cl);
    }
    if (out_pow) out_pow[o] = a;
  }
  if (gmax) {
    lmax = avs_wave_max(lmax);
    // positive floats order like their bit patterns: an integer max, order-independent
    if ((tid & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(gmax), __float_as_uint(lmax));
  }
}

extern "C" int avs_stft_mel_fused_f32(const float* d_wave, int64_t t, const double* d_window, const double* d_cos,
                                      const double* d_sin, const float* d_fb, const int* d_fb_lo, const int* d_fb_hi,
                                      int nmel, float* d_log2mel, float* d_db, float* d_power, float* d_max,
                                      avs_stream_t stream) {
  const char* who = "avs_stft_mel_fused_f32";
  AVS_REQUIRE(t > AVS_FUSED_NFFT / 2, AVS_E_SHAPE, "%s: reflect padding needs more than %d samples, got %lld", who,
              AVS_FUSED_NFFT / 2, (long long)t);
  AVS_REQUIRE(nmel > 0 && nmel <= 1024, AVS_E_SHAPE, "%s: nmel=%d", who, nmel);
  AVS_REQUIRE(d_wave && d_window && d_cos && d_sin && d_fb && d_fb_lo && d_fb_hi, AVS_E_ARG, "%s: null pointer", who);
  AVS_REQUIRE(d_log2mel || d_db || d_power || d_max, AVS_E_ARG, "%s: no output requested", who);
  AVS_REQUIRE(!d_db || d_max, AVS_E_ARG, "%s: the dB output needs d_max", who);
  AVS_REQUIRE(avs_aligned16(d_wave), AVS_E_ALIGN, "%s: the waveform must be 16-byte aligned", who);
  const long long frames = 1 + t / AVS_FUSED_HOP;
  const long long blocks = avs_cdiv(frames, AVS_FUSED_FPB);
  AVS_REQUIRE(blocks < (1ll << 31), AVS_E_SHAPE, "%s: too many frames", who);
  hipLaunchKernelGGL(stft_mel_fused_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_wave,
                     (long long)t, frames, d_window, d_cos, d_sin, d_fb, d_fb_lo, d_fb_hi, nmel, d_log2mel, d_db, d_power,
                     d_max, (const int*)nullptr, (float*)nullptr, (float*)nullptr, (const float*)nullptr, 0.f, (float*)nullptr,
                     (const long long*)nullptr, (const long long*)nullptr);
  AVS_CHECK_LAUNCH(who);
  return AVS_OK;
}

// One-pass mode of avs_stft_mel_segmean_f32: part_db[b, m] = sum over the block's frames (in order) of max(db_rows, threshold),
// the threshold from the track maximum the front-end pass has just found.  Bandwidth-bound: reads the dB rows once.
__global__ __launch_bounds__(256) void segment_db_sum_kernel(const float* __restrict__ db_rows, const int* __restrict__ blocks,
                                                             int bstride, int nmel, const float* __restrict__ gmax,
                                                             float top_db, float* __restrict__ part_db) {
  int nf = blocks[bstride * blockIdx.x + 1];
  nf = nf < 0 ? 0 : (nf > AVS_FUSED_FPB ? AVS_FUSED_FPB : nf);
  if (bstride == 4) gmax += blocks[4 * blockIdx.x + 3];   // batch mode: the block's track
  const float thr = 10.f * log10f(*gmax) - top_db;
  for (int m = threadIdx.x; m < nmel; m += 256) {
    const float* __restrict__ r = db_rows + (long long)blockIdx.x * AVS_FUSED_FPB * nmel + m;
    float s = 0.f;
    for (int f = 0; f < nf; ++f) s += fmaxf(r[(long long)f * nmel], thr);
    part_db[(long long)blockIdx.x * nmel + m] = s;
  }
}

// out[s, m] = (sum of part[b, m] over the blocks b of segment s, in block order) / frames of s
__global__ __launch_bounds__(256) void segment_fold_kernel(const float* __restrict__ part, const int* __restrict__ seg_block,
                                                           const int* __restrict__ seg_frames, int nseg, int nmel,
                                                           float* __restrict__ out, long long ldo) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)nseg * nmel;
       i += (long long)gridDim.x * blockDim.x) {
    const int sgm = (int)(i / nmel), m = (int)(i - (long long)sgm * nmel);
    float a = 0.f;
    for (int b = seg_block[sgm]; b < seg_block[sgm + 1]; ++b) a += part[(long long)b * nmel + m];
    out[sgm * ldo + m] = seg_frames[sgm] > 0 ? a / (float)seg_frames[sgm] : 0.f;
  }
}

extern "C" int64_t avs_stft_mel_segmean_workspace_bytes(int nblocks, int nmel, int want_log2, int want_db, int find_max) {
  if (nblocks < 0 || nmel <= 0) return AVS_E_SHAPE;
  const int64_t per = (int64_t)nblocks * nmel * 4;
  // per-block sums of each requested mean + (one-pass mode) the blocks' unclamped dB rows [nblocks][32][nmel]
  return per * ((want_log2 ? 1 : 0) + (want_db ? 1 : 0)) + ((want_db && find_max) ? per * AVS_FUSED_FPB : 0);
}

extern "C" int avs_stft_mel_segmean_f32(const float* d_wave, int64_t t, const double* d_window, const double* d_cos,
                                        const double* d_sin, const float* d_fb, const int* d_fb_lo, const int* d_fb_hi,
                                        int nmel, const int* d_blocks, int nblocks, const int* d_seg_block,
                                        const int* d_seg_frames, int nseg, float* d_max, int find_max, float top_db,
                                        float* d_mean_log2, int64_t ld_log2, float* d_mean_db, int64_t ld_db, void* d_ws,
                                        int64_t ws_bytes, avs_stream_t stream) {
  const char* who = "avs_stft_mel_segmean_f32";
  AVS_REQUIRE(t > AVS_FUSED_NFFT / 2, AVS_E_SHAPE, "%s: reflect padding needs more than %d samples, got %lld", who,
              AVS_FUSED_NFFT / 2, (long long)t);
  AVS_REQUIRE(nmel > 0 && nmel <= 1024 && nblocks >= 0 && nseg >= 0, AVS_E_SHAPE, "%s: nmel=%d nblocks=%d nseg=%d", who,
              nmel, nblocks, nseg);
  if (nseg == 0) return AVS_OK;
  AVS_REQUIRE(d_wave && d_window && d_cos && d_sin && d_fb && d_fb_lo && d_fb_hi && d_blocks && d_seg_block && d_seg_frames,
              AVS_E_ARG, "%s: null pointer", who);
  AVS_REQUIRE(d_mean_log2 || d_mean_db, AVS_E_ARG, "%s: no output requested", who);
  AVS_REQUIRE(!d_mean_db || d_max, AVS_E_ARG, "%s: the dB mean needs the track's maximum mel power (d_max)", who);
  AVS_REQUIRE((!d_mean_log2 || ld_log2 >= nmel) && (!d_mean_db || ld_db >= nmel), AVS_E_SHAPE, "%s: output rows too short", who);
  AVS_REQUIRE(avs_aligned16(d_wave), AVS_E_ALIGN, "%s: the waveform must be 16-byte aligned", who);
  const int64_t per = (int64_t)nblocks * nmel * 4;
  const bool one_pass = d_mean_db && find_max;
  const int64_t need = avs_stft_mel_segmean_workspace_bytes(nblocks, nmel, d_mean_log2 != nullptr, d_mean_db != nullptr, find_max);
  AVS_REQUIRE(d_ws && ws_bytes >= need, AVS_E_WORKSPACE, "%s: workspace %lld < %lld bytes", who, (long long)ws_bytes,
              (long long)need);
  float* p_log2 = d_mean_log2 ? (float*)d_ws : nullptr;
  float* p_db = d_mean_db ? (float*)((char*)d_ws + (d_mean_log2 ? per : 0)) : nullptr;
  const long long frames = 1 + t / AVS_FUSED_HOP;
  float* rows = one_pass ? (float*)((char*)d_ws + per * ((d_mean_log2 ? 1 : 0) + 1)) : nullptr;
  if (one_pass) AVS_REQUIRE(hipMemsetAsync(d_max, 0, sizeof(float), (hipStream_t)stream) == hipSuccess, AVS_E_HIP, "%s: memset", who);
  if (nblocks > 0) {
    hipLaunchKernelGGL(stft_mel_fused_kernel<true>, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, d_wave,
                       (long long)t, frames, d_window, d_cos, d_sin, d_fb, d_fb_lo, d_fb_hi, nmel, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, one_pass ? d_max : (float*)nullptr, d_blocks, p_log2, p_db, d_max,
                       top_db, rows, (const long long*)nullptr, (const long long*)nullptr);
    if (one_pass)
      hipLaunchKernelGGL(segment_db_sum_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, rows, d_blocks, 3, nmel,
                         d_max, top_db, p_db);
  }
  long long gx = avs_cdiv((long long)nseg * nmel, 256);
  if (gx > 4096) gx = 4096;
  if (d_mean_log2)
    hipLaunchKernelGGL(segment_fold_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, p_log2, d_seg_block,
                       d_seg_frames, nseg, nmel, d_mean_log2, (long long)ld_log2);
  if (d_mean_db)
    hipLaunchKernelGGL(segment_fold_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, p_db, d_seg_block,
                       d_seg_frames, nseg, nmel, d_mean_db, (long long)ld_db);
  AVS_CHECK_LAUNCH(who);
  return AVS_OK;
}

// The same for a BATCH of tracks in one set of launches: d_waves holds the tracks one after another (track i at sample offset
// d_track_off[i], a multiple of 4; d_track_len[i] samples), block rows are (first STFT frame inside the track, frames <= 32,
// segment, track), segments are numbered over the whole batch; d_max [ntracks] receives every track's maximum (find_max = 1
// semantics: the table covers every track).  Four launches for all tracks instead of five per track.
extern "C" int avs_stft_mel_segmean_batch_f32(const float* d_waves, const int64_t* d_track_off, const int64_t* d_track_len,
                                              int ntracks, const double* d_window, const double* d_cos, const double* d_sin,
                                              const float* d_fb, const int* d_fb_lo, const int* d_fb_hi, int nmel,
                                              const int* d_blocks, int nblocks, const int* d_seg_block, const int* d_seg_frames,
                                              int nseg, float* d_max, float top_db, float* d_mean_log2, int64_t ld_log2,
                                              float* d_mean_db, int64_t ld_db, void* d_ws, int64_t ws_bytes, avs_stream_t stream) {
  const char* who = "avs_stft_mel_segmean_batch_f32";
  AVS_REQUIRE(ntracks >= 0 && nmel > 0 && nmel <= 1024 && nblocks >= 0 && nseg >= 0, AVS_E_SHAPE, "%s: ntracks=%d nmel=%d nblocks=%d nseg=%d",
              who, ntracks, nmel, nblocks, nseg);
  if (nseg == 0 || ntracks == 0) return AVS_OK;
  AVS_REQUIRE(d_waves && d_track_off && d_track_len && d_window && d_cos && d_sin && d_fb && d_fb_lo && d_fb_hi && d_blocks &&
                  d_seg_block && d_seg_frames && d_max,
              AVS_E_ARG, "%s: null pointer", who);
  AVS_REQUIRE(d_mean_log2 || d_mean_db, AVS_E_ARG, "%s: no output requested", who);
  AVS_REQUIRE((!d_mean_log2 || ld_log2 >= nmel) && (!d_mean_db || ld_db >= nmel), AVS_E_SHAPE, "%s: output rows too short", who);
  AVS_REQUIRE(avs_aligned16(d_waves), AVS_E_ALIGN, "%s: the waveforms must be 16-byte aligned", who);
  const int64_t per = (int64_t)nblocks * nmel * 4;
  const int64_t need = avs_stft_mel_segmean_workspace_bytes(nblocks, nmel, d_mean_log2 != nullptr, d_mean_db != nullptr, 1);
  AVS_REQUIRE(d_ws && ws_bytes >= need, AVS_E_WORKSPACE, "%s: workspace %lld < %lld bytes", who, (long long)ws_bytes, (long long)need);
  float* p_log2 = d_mean_log2 ? (float*)d_ws : nullptr;
  float* p_db = d_mean_db ? (float*)((char*)d_ws + (d_mean_log2 ? per : 0)) : nullptr;
  float* rows = d_mean_db ? (float*)((char*)d_ws + per * ((d_mean_log2 ? 1 : 0) + 1)) : nullptr;
  AVS_REQUIRE(hipMemsetAsync(d_max, 0, sizeof(float) * ntracks, (hipStream_t)stream) == hipSuccess, AVS_E_HIP, "%s: memset", who);
  if (nblocks > 0) {
    hipLaunchKernelGGL(stft_mel_fused_kernel<true>, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, d_waves, 0ll, 0ll,
                       d_window, d_cos, d_sin, d_fb, d_fb_lo, d_fb_hi, nmel, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                       d_mean_db ? d_max : (float*)nullptr, d_blocks, p_log2, p_db, d_max, top_db, rows,
                       (const long long*)d_track_off, (const long long*)d_track_len);
    if (d_mean_db)
      hipLaunchKernelGGL(segment_db_sum_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, rows, d_blocks, 4, nmel,
                         d_max, top_db, p_db);
  }
  long long gx = avs_cdiv((long long)nseg * nmel, 256);
  if (gx > 4096) gx = 4096;
  if (d_mean_log2)
    hipLaunchKernelGGL(segment_fold_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, p_log2, d_seg_block,
                       d_seg_frames, nseg, nmel, d_mean_log2, (long long)ld_log2);
  if (d_mean_db)
    hipLaunchKernelGGL(segment_fold_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, p_db, d_seg_block,
                       d_seg_frames, nseg, nmel, d_mean_db, (long long)ld_db);
  AVS_CHECK_LAUNCH(who);
  return AVS_OK;
}

__global__ __launch_bounds__(256) void clamp_topdb_kernel(float* __restrict__ x, long long count,
                                                          const float* __restrict__ gmax, float top_db) {
  const float thr = 10.f * log10f(*gmax) - top_db;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (long long)gridDim.x * blockDim.x)
    x[i] = fmaxf(x[i], thr);
}

extern "C" int avs_clamp_topdb_f32(float* d_x, int64_t count, const float* d_max, float top_db, avs_stream_t stream) {
  AVS_REQUIRE(count >= 0, AVS_E_SHAPE, "avs_clamp_topdb_f32: negative count");
  if (count == 0) return AVS_OK;
  AVS_REQUIRE(d_x && d_max, AVS_E_ARG, "avs_clamp_topdb_f32: null pointer");
  long long gx = avs_cdiv(count, 256);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(clamp_topdb_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, d_x, (long long)count,
                     d_max, top_db);
  AVS_CHECK_LAUNCH("avs_clamp_topdb_f32");
  return AVS_OK;
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ x, long long count, float v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (long long)gridDim.x * blockDim.x)
    x[i] = v;
}

extern "C" int avs_fill_f32(float* d_x, int64_t count, float value, avs_stream_t stream) {
  AVS_REQUIRE(count >= 0, AVS_E_SHAPE, "avs_fill_f32: negative count");
  if (count == 0) return AVS_OK;
  AVS_REQUIRE(d_x, AVS_E_ARG, "avs_fill_f32: null pointer");
  long long gx = avs_cdiv(count, 256);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, d_x, (long long)count, value);
  AVS_CHECK_LAUNCH("avs_fill_f32");
  return AVS_OK;
}

// y = rint((clamp(x, lo, hi) - lo) * scale): the 8-bit quantiser of the VGGish post-processor (values kept as
// float, round-half-to-even like torch.round).
__global__ __launch_bounds__(256) void quantize_kernel(const float* __restrict__ x, long long count, float lo, float hi,
                                                       float scale, float* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (long long)gridDim.x * blockDim.x)
    y[i] = rintf((fminf(fmaxf(x[i], lo), hi) - lo) * scale);
}

extern "C" int avs_quantize_f32(const float* d_x, int64_t count, float lo, float hi, float scale, float* d_y,
                                avs_stream_t stream) {
  AVS_REQUIRE(count >= 0 && hi > lo, AVS_E_SHAPE, "avs_quantize_f32: bad arguments");
  if (count == 0) return AVS_OK;
  AVS_REQUIRE(d_x && d_y, AVS_E_ARG, "avs_quantize_f32: null pointer");
  long long gx = avs_cdiv(count, 256);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, d_x, (long long)count, lo,
                     hi, scale, d_y);
  AVS_CHECK_LAUNCH("avs_quantize_f32");
  return AVS_OK;
}

// ---------------------------------------------------------------------------
// Channel mix-down + rational resampling (features/extractors.py:364-378 delegates to
// pydub/ffmpeg "set_channels(1).set_frame_rate(16000)", :326-328 averages channels): polyphase FIR
//   y[i*up + p] = sum_k mono[i*down + k - width] * taps[p][k],   mono[j] = mean_c x[j, c]  (0 outside the clip)
// with `up` phases of `ntaps` windowed-sinc coefficients built on the host (audio.py).  One thread per output
// sample; the taps of a phase are contiguous, the input window of neighbouring outputs overlaps (L1/L2 hits).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, long long t, int channels,
                                                       const float* __restrict__ taps, int up, int down, int ntaps,
                                                       int width, float* __restrict__ y, long long out_len) {
  const float inv_c = 1.f / (float)channels;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < out_len;
       o += (long long)gridDim.x * blockDim.x) {
    const long long i = o / up;
    const int ph = (int)(o - i * up);
    const long long base = i * down - width;
    const float* __restrict__ tp = taps + (long long)ph * ntaps;
    float acc = 0.f;
    for (int k = 0; k < ntaps; ++k) {
      const long long j = base + k;
      if (j < 0 || j >= t) continue;
      float v = 0.f;
      for (int c = 0; c < channels; ++c) v += x[j * channels + c];
      if (channels > 1) v *= inv_c;
      acc = fmaf(v, tp[k], acc);
    }
    y[o] = acc;
  }
}

extern "C" int avs_resample_f32(const float* d_x, int64_t t, int channels, const float* d_taps, int up, int down,
                                int ntaps, int width, float* d_y, int64_t out_len, avs_stream_t stream) {
  AVS_REQUIRE(t >= 0 && channels > 0 && up > 0 && down > 0 && ntaps > 0 && width >= 0 && out_len >= 0, AVS_E_SHAPE,
              "avs_resample_f32: bad extents");
  if (out_len == 0) return AVS_OK;
  AVS_REQUIRE(d_x && d_taps && d_y, AVS_E_ARG, "avs_resample_f32: null pointer");
  AVS_REQUIRE((out_len - 1) / up * down - width < t, AVS_E_SHAPE,
              "avs_resample_f32: %lld output samples need input past the %lld given", (long long)out_len, (long long)t);
  long long gx = avs_cdiv(out_len, 256);
  if (gx > 65536) gx = 65536;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, d_x, (long long)t, channels,
                     d_taps, up, down, ntaps, width, d_y, (long long)out_len);
  AVS_CHECK_LAUNCH("avs_resample_f32");
  return AVS_OK;
}
