"""Mel-spectrogram / MFCC front end on the MI355X.

Replaces torchaudio.transforms.MelSpectrogram / MFCC as called from
features/extractors.py:236-246 (torchaudio is a third-party dependency that is
not in the reference tree; its defaults are restated in SURVEY Appendix A.1-A.4:
n_fft = win = 400, periodic Hann, hop 200, center/reflect, power 2, HTK mel
scale 0..sr/2, norm None, DCT-II ortho, AmplitudeToDB('power', top_db=80)).

Constants are built once per (sample_rate, device): filterbank and DCT in fp32
(the reference's own fp32 evaluation), the window's fp32 values and the cos / -sin
tables of the 400-point DFT in float64.  log2-mel, mel power and the dB spectrogram
come from ONE fused kernel (avs_stft_mel_fused_f32): a block stages the waveform
span of its 32 frames in LDS (reflect padding by index, no padded copy), runs the
real DFT - folded to half its length by the symmetry of the window and the
cosines - on the fp64 matrix cores, and applies |.|^2, the sparse mel sum and the
log without the spectrum ever leaving the chip.  ``spectrum`` keeps the unfused
dense-DFT kernel (the form VGGish's 512-point front end uses) for tests.
"""
import math

import numpy as np
import torch

from . import ops

N_FFT = 400
HOP = 200
N_BINS = N_FFT // 2 + 1  # 201


def _hz_to_mel(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_filterbank(sample_rate, n_mels, n_freqs=N_BINS, f_min=0.0, f_max=None):
    """[n_freqs, n_mels] triangular HTK filters, norm=None.  Evaluated in float32 tensor arithmetic in the
    order torchaudio.functional.melscale_fbanks uses, because the reference's numbers ARE that float32
    evaluation (a float64 evaluation differs by ~1e-5 in some weights and by one non-zero entry)."""
    f_max = float(sample_rate // 2) if f_max is None else f_max
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel(f_min), _hz_to_mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up)).numpy()


def dct_matrix(n_mfcc, n_mels):
    """[n_mfcc, n_mels] DCT-II with ortho normalisation (rows = coefficients), float32 as torchaudio.create_dct."""
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc)).unsqueeze(1)
    d = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    d[0] *= 1.0 / math.sqrt(2.0)
    d *= math.sqrt(2.0 / float(n_mels))
    return d.numpy()


def windowed_dft_basis():
    """[2*N_BINS, N_FFT]: rows 0..200 = w[n] cos(2 pi k n / N), rows 201..401 = -w[n] sin(...)."""
    n = np.arange(N_FFT)
    # the float32 values of torch.hann_window(400) (periodic), which is the window the reference multiplies by
    window = torch.hann_window(N_FFT).double().numpy()
    k = np.arange(N_BINS)[:, None]
    ang = 2.0 * math.pi * ((k * n[None, :]) % N_FFT) / N_FFT
    return np.concatenate([np.cos(ang) * window, -np.sin(ang) * window], 0)


FUSED_COLS = 208      # 201 bins padded to 13 tiles of 16 (avs_stft_mel_fused_f32)
FUSED_RE_ROWS = 204   # n = 0 .. 200 padded to a multiple of 4
FUSED_IM_ROWS = 200   # n = 1 .. 199 padded to a multiple of 4


def folded_dft_tables():
    """Tables of the folded real DFT: cos[n, k] = cos(2 pi k n / 400) for n = 0..200, -sin[n - 1, k] for n = 1..199
    (k = 0..200), zero elsewhere; float64, the argument reduced mod 400 before the evaluation."""
    k = np.arange(N_BINS)[None, :]
    cos_t = np.zeros((FUSED_RE_ROWS, FUSED_COLS), dtype=np.float64)
    n = np.arange(0, N_FFT // 2 + 1)[:, None]
    cos_t[:N_FFT // 2 + 1, :N_BINS] = np.cos(2.0 * math.pi * ((k * n) % N_FFT) / N_FFT)
    sin_t = np.zeros((FUSED_IM_ROWS, FUSED_COLS), dtype=np.float64)
    n = np.arange(1, N_FFT // 2)[:, None]
    sin_t[:N_FFT // 2 - 1, :N_BINS] = -np.sin(2.0 * math.pi * ((k * n) % N_FFT) / N_FFT)
    return cos_t, sin_t


class _SegmentTable(tuple):
    """(blocks, seg_block, seg_frames) device tensors + .span = the (first, last) STFT frame the segments cover."""


class MelPlan:
    """Device-resident constants + the launch sequence for log2-mel and MFCC."""

    _cache = {}

    def __init__(self, sample_rate, n_mels, n_mfcc, device):
        self.sample_rate, self.n_mels, self.n_mfcc, self.device = sample_rate, n_mels, n_mfcc, device
        self.one_pass_means = True   # segment_means: one DFT pass when the segments cover the track (False: always two)
        basis = windowed_dft_basis()  # [402, 400] float64: fp32 window values x cos/sin evaluated in float64
        bt = np.zeros((N_FFT, 448), dtype=np.float64)
        bt[:, :2 * N_BINS] = basis.T
        self.basis_t = torch.from_numpy(bt).to(device)
        fb = mel_filterbank(sample_rate, n_mels).astype(np.float32)
        nz = fb > 0
        lo = np.where(nz.any(0), nz.argmax(0), 0).astype(np.int32)
        hi = np.where(nz.any(0), N_BINS - nz[::-1].argmax(0), 0).astype(np.int32)
        self.fb = torch.from_numpy(fb).to(device)
        self.fb_lo = torch.from_numpy(lo).to(device)
        self.fb_hi = torch.from_numpy(hi).to(device)
        self.dct = torch.from_numpy(dct_matrix(n_mfcc, n_mels).astype(np.float32)).to(device)
        cos_t, sin_t = folded_dft_tables()
        self.cos_t = torch.from_numpy(cos_t).to(device)
        self.sin_t = torch.from_numpy(sin_t).to(device)
        # the float32 values of torch.hann_window(400) (periodic): the window the reference multiplies by
        self.window = torch.hann_window(N_FFT).double().to(device)

    @classmethod
    def get(cls, sample_rate, n_mels, n_mfcc, device):
        key = (sample_rate, n_mels, n_mfcc, str(device))
        if key not in cls._cache:
            cls._cache[key] = cls(sample_rate, n_mels, n_mfcc, device)
        return cls._cache[key]

    @staticmethod
    def num_frames(t):
        return 1 + t // HOP

    def spectrum(self, wave):
        """wave fp32 [T] on device, T > 200 -> (re | im) [frames, 402] of the Hann-windowed STFT."""
        t = wave.numel()
        if t <= N_FFT // 2:
            # torch.stft(center=True, pad_mode="reflect") raises for T <= pad as well
            raise RuntimeError(f"Argument #4: Padding size should be less than the corresponding input dimension, "
                               f"but got: padding ({N_FFT // 2}, {N_FFT // 2}) at dimension 1 of input [1, {t}]")
        frames = self.num_frames(t)
        padded = ops.reflect_pad(wave.contiguous(), N_FFT // 2, t + N_FFT)
        return ops.stft_f64(padded, frames, HOP, N_FFT, self.basis_t, 2 * N_BINS)

    def _fused(self, wave, log2=False, db=False, power=False):
        t = wave.numel()
        if t <= N_FFT // 2:
            # torch.stft(center=True, pad_mode="reflect") raises for T <= pad as well
            raise RuntimeError(f"Argument #4: Padding size should be less than the corresponding input dimension, "
                               f"but got: padding ({N_FFT // 2}, {N_FFT // 2}) at dimension 1 of input [1, {t}]")
        wave = wave.contiguous()
        if wave.data_ptr() % 16:
            wave = wave.clone()   # a slice that starts off a 16-byte boundary: the kernel stages with 16-byte loads
        return ops.stft_mel_fused(wave, self.window, self.cos_t, self.sin_t, self.fb, self.fb_lo,
                                  self.fb_hi, log2, db, power)

    def log2_mel(self, wave):
        """[frames, n_mels] = log2(mel + 1e-6)  (features/extractors.py:241-246)."""
        return self._fused(wave, log2=True)[0]

    def mel_power(self, wave):
        return self._fused(wave, power=True)[2]

    def mfcc(self, wave, top_db=80.0):
        """[frames, n_mfcc] (torchaudio MFCC, log_mels=False)."""
        _, db, _, gmax = self._fused(wave, db=True)
        ops.clamp_topdb(db, gmax, top_db)
        return ops.linear(db, self.dct)

    @staticmethod
    def segment_table(bounds, device):
        """bounds: STFT-frame boundaries [nseg + 1] of consecutive segments (host ints) -> the device tables of
        avs_stft_mel_segmean_f32: blocks int32 [nblocks, 3] (first frame, frames <= 32, segment), first block of each
        segment [nseg + 1], frames per segment [nseg]."""
        blocks, seg_block, seg_frames = [], [0], []
        for sgm, (a, b) in enumerate(zip(bounds[:-1], bounds[1:])):
            a, b = int(a), int(b)
            for f in range(a, b, 32):
                blocks.append((f, min(32, b - f), sgm))
            seg_block.append(len(blocks))
            seg_frames.append(max(0, b - a))
        mk = lambda v, shape: torch.tensor(v, dtype=torch.int32).reshape(shape).to(device)
        table = _SegmentTable((mk(blocks, (len(blocks), 3)), mk(seg_block, (len(seg_block),)), mk(seg_frames, (len(seg_frames),))))
        table.span = (int(bounds[0]), int(bounds[-1])) if len(bounds) else (0, 0)   # the STFT frames the segments cover
        return table

    @staticmethod
    def batch_tables(waves, bounds_per_track, device):
        """A batch of tracks for segment_means_batch: the waveforms one after another (16-byte aligned starts) + per track the
        STFT-frame boundaries of its consecutive segments (each list must start at 0 and end at the track's frame count: the
        clamp is relative to the TRACK's maximum) -> (waves_cat, track_off, track_len, blocks [nblocks, 4], seg_block, seg_frames)."""
        offs, lens, cur = [], [], 0
        for w in waves:
            offs.append(cur)
            lens.append(int(w.numel()))
            cur += (int(w.numel()) + 3) // 4 * 4
        cat = torch.zeros(max(cur, 4), dtype=torch.float32, device=device)
        for w, o in zip(waves, offs):
            cat[o:o + w.numel()] = w.to(device=device, dtype=torch.float32)
        blocks, seg_block, seg_frames = [], [0], []
        sgm = 0
        for trk, (bounds, ln) in enumerate(zip(bounds_per_track, lens)):
            if ln <= N_FFT // 2:
                raise RuntimeError(f"track {trk}: reflect padding needs more than {N_FFT // 2} samples")
            if int(bounds[0]) != 0 or int(bounds[-1]) != 1 + ln // HOP:
                raise ValueError("segment_means_batch: a track's segments must cover it (the clamp is relative to the track's maximum)")
            for a, b in zip(bounds[:-1], bounds[1:]):
                a, b = int(a), int(b)
                for f in range(a, b, 32):
                    blocks.append((f, min(32, b - f), sgm, trk))
                seg_block.append(len(blocks))
                seg_frames.append(max(0, b - a))
                sgm += 1
        mk = lambda v, shape, dt: torch.tensor(v, dtype=dt).reshape(shape).to(device)
        return (cat, mk(offs, (len(offs),), torch.int64), mk(lens, (len(lens),), torch.int64),
                mk(blocks, (len(blocks), 4), torch.int32), mk(seg_block, (len(seg_block),), torch.int32),
                mk(seg_frames, (len(seg_frames),), torch.int32))

    def segment_means_batch(self, tables, out_log2=None, out_mfcc_db=None, top_db=80.0):
        """segment_means for every track of a batch in ONE set of launches (tables = batch_tables(...)); segments are
        numbered over the batch in track order."""
        cat, toff, tlen, blocks, seg_block, seg_frames = tables
        return ops.stft_mel_segmean_batch(cat, toff, tlen, self.window, self.cos_t, self.sin_t, self.fb, self.fb_lo, self.fb_hi,
                                          blocks, seg_block, seg_frames, top_db=top_db, out_log2=out_log2, out_db=out_mfcc_db)[:2]

    def segment_means(self, wave, table, out_log2=None, out_mfcc_db=None, top_db=80.0):
        """Per segment (a shot's slice of the track) the time mean of the log2-mel rows (out_log2 [nseg, >= n_mels]) and
        of the top_db-clamped dB-mel rows (out_mfcc_db; its DCT is the mean of the MFCC rows: the DCT and mfcc_proj are
        linear); features/extractors.py:232-246 pool the per-frame matrices over time.  The clamp is relative to the
        TRACK's maximum: when the segments cover the whole track (the usual case) ONE pass of the DFT finds it while it
        writes the unclamped dB rows to a workspace, and a bandwidth-bound pass clamps and sums them; otherwise the
        maximum takes its own pass over the track first.  Same values, same order: bit-identical either way."""
        t = wave.numel()
        if t <= N_FFT // 2:
            raise RuntimeError(f"Argument #4: Padding size should be less than the corresponding input dimension, "
                               f"but got: padding ({N_FFT // 2}, {N_FFT // 2}) at dimension 1 of input [1, {t}]")
        wave = wave.contiguous()
        if wave.data_ptr() % 16:
            wave = wave.clone()
        args = (wave, self.window, self.cos_t, self.sin_t, self.fb, self.fb_lo, self.fb_hi)
        covered = getattr(table, "span", None) == (0, 1 + t // HOP)
        gmax = ops.stft_mel_max(*args) if (out_mfcc_db is not None and not (covered and self.one_pass_means)) else None
        return ops.stft_mel_segmean(*args, *table, gmax=gmax, top_db=top_db, out_log2=out_log2, out_db=out_mfcc_db)

    def log2_mel_and_mfcc(self, wave, top_db=80.0):
        """Both features of one waveform from ONE pass over it (the spectrum is shared): ([frames, n_mels], [frames, n_mfcc])."""
        mel, db, _, gmax = self._fused(wave, log2=True, db=True)
        ops.clamp_topdb(db, gmax, top_db)
        return mel, ops.linear(db, self.dct)


# --------------------------------------------------------------------------- channel mix-down + resampling (F4)
_RESAMPLE_TAPS = {}


def resample_taps(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """Polyphase windowed-sinc taps fp32 [new, 2*width + orig] and `width` for orig -> new (reduced by their gcd):
    Hann-windowed sinc, cut-off `rolloff` x the lower Nyquist, evaluated in float64.  The reference delegates this
    step to ffmpeg through pydub (features/extractors.py:364-378), which cannot be reproduced bit for bit; this is
    the published sinc_interp_hann design of torchaudio.functional.resample [3P-memory], parity unpinned."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = np.clip(t * base, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    safe = np.where(t == 0, 1.0, t)
    kern = np.where(t == 0, 1.0, np.sin(safe) / safe) * window * (base / orig)
    return kern.astype(np.float32), width, orig, new


def resample_to(x, orig_freq, new_freq):
    """x fp32 [T] or interleaved [T, channels] on device -> mono fp32 [ceil(new*T/orig)] at new_freq."""
    if int(orig_freq) == int(new_freq):
        if x.dim() == 1:
            return x
        taps = torch.ones((1, 1), dtype=torch.float32, device=x.device)   # pure channel mix-down
        return ops.resample(x.contiguous(), taps, 1, 1, 0, x.shape[0])
    key = (int(orig_freq), int(new_freq), str(x.device))
    if key not in _RESAMPLE_TAPS:
        kern, width, orig, new = resample_taps(orig_freq, new_freq)
        _RESAMPLE_TAPS[key] = (torch.from_numpy(kern).to(x.device), width, orig, new)
    taps, width, orig, new = _RESAMPLE_TAPS[key]
    t = x.shape[0]
    out_len = -(-new * t // orig)
    return ops.resample(x.contiguous(), taps, new, orig, width, out_len)
