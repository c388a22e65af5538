"""Mel-spectrogram / MFCC front end on the MI355X.

Replaces torchaudio.transforms.MelSpectrogram / MFCC as called from
features/extractors.py:236-246 (torchaudio is a third-party dependency that is
not in the reference tree; its defaults are restated in SURVEY Appendix A.1-A.4:
n_fft = win = 400, periodic Hann, hop 200, center/reflect, power 2, HTK mel
scale 0..sr/2, norm None, DCT-II ortho, AmplitudeToDB('power', top_db=80)).

Constants are built once per (sample_rate, device) in float64 and rounded to
fp32 (filterbank, DCT) or kept in float64 (the windowed DFT basis); the Hann
window is folded into the basis so the STFT is one dense contraction on the
fp64 matrix cores whose rows are the overlapping frames of the reflect-padded
waveform, read in place (row stride = hop).
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


class MelPlan:
    """Device-resident constants + the launch sequence for log2-mel and MFCC."""

    _cache = {}

    def __init__(self, sample_rate, n_mels, n_mfcc, device):
        self.sample_rate, self.n_mels, self.n_mfcc, self.device = sample_rate, n_mels, n_mfcc, device
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

    def log2_mel(self, wave):
        """[frames, n_mels] = log2(mel + 1e-6)  (features/extractors.py:241-246)."""
        return ops.power_mel(self.spectrum(wave), N_BINS, self.fb, self.fb_lo, self.fb_hi, 0)

    def mel_power(self, wave):
        return ops.power_mel(self.spectrum(wave), N_BINS, self.fb, self.fb_lo, self.fb_hi, 2)

    def mfcc(self, wave, top_db=80.0):
        """[frames, n_mfcc] (torchaudio MFCC, log_mels=False)."""
        gmax = torch.zeros(1, dtype=torch.float32, device=wave.device)
        db = ops.power_mel(self.spectrum(wave), N_BINS, self.fb, self.fb_lo, self.fb_hi, 1, gmax)
        ops.clamp_topdb(db, gmax, top_db)
        return ops.linear(db, self.dct)


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
