"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the audio features.

Follows features/extractors.py:236-246 (_extract_mfcc / _extract_mel) and the
literal behaviour of AudioFeatureExtractor.forward (:195-208).

The arithmetic lives in a third-party dependency that is NOT in /root/reference
and is not version-pinned anywhere (no requirements file): torchaudio
(transforms.MelSpectrogram / MFCC / AmplitudeToDB, functional.melscale_fbanks /
create_dct).  Its published algorithm is restated here in torchaudio's own
float32 op order.  The reference holds no fixtures for
it => PARITY UNPINNED for the torchaudio-defined numbers; what pins this file
are analytic known-answer tests (tests/test_oracle_analytic.py: single-bin sine,
Parseval, filterbank structure 4 empty filters / 394 non-zeros / <= 2 filters
per bin, DCT orthonormality) and the reference call sites' shapes.
"""
import math

import numpy as np
import torch

N_FFT, HOP, N_BINS = 400, 200, 201


def melscale_fbanks(n_freqs=N_BINS, f_min=0.0, f_max=8000.0, n_mels=128, sample_rate=16000):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk'), float32 ops."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    zero = torch.zeros(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(zero, torch.min(down, up))  # [n_freqs, n_mels]


def create_dct(n_mfcc=40, n_mels=128):
    """torchaudio.functional.create_dct(norm='ortho') -> [n_mels, n_mfcc]."""
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc)).unsqueeze(1)
    dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    dct[0] *= 1.0 / math.sqrt(2.0)
    dct *= math.sqrt(2.0 / float(n_mels))
    return dct.t()


def power_spectrogram(wave):
    """Spectrogram(n_fft=400, hop=200, periodic Hann, center, reflect, power=2) -> [201, frames]."""
    window = torch.hann_window(N_FFT)
    spec = torch.stft(wave, N_FFT, HOP, N_FFT, window, center=True, pad_mode="reflect", normalized=False,
                      onesided=True, return_complex=True)
    return spec.abs().pow(2.0)


def mel_spectrogram(wave, sample_rate=16000, n_mels=128):
    fb = melscale_fbanks(N_BINS, 0.0, float(sample_rate // 2), n_mels, sample_rate)
    spec = power_spectrogram(wave)
    return torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)  # [n_mels, frames]


def extract_mel(wave, sample_rate=16000):
    """features/extractors.py:241-246 -> np.float32 [frames, 128]."""
    mel = torch.log2(mel_spectrogram(wave, sample_rate, 128) + 1e-6)
    return mel.permute(1, 0).detach().numpy().reshape(-1, 128)


def extract_mel_f64(wave, sample_rate=16000, n_mels=128):
    """The same feature in float64 from the defining formulas (dense DFT): the yardstick that tells how
    far ANY float32 implementation (torch.stft's included) is from the exact value on a given signal."""
    x = np.asarray(wave, dtype=np.float64)
    xp = np.pad(x, N_FFT // 2, mode="reflect")
    frames = 1 + len(x) // HOP
    fr = np.stack([xp[HOP * f:HOP * f + N_FFT] for f in range(frames)])
    n = np.arange(N_FFT)
    win = torch.hann_window(N_FFT).double().numpy()  # the float32 window values ARE the algorithm's constants
    ang = 2.0 * np.pi * ((np.arange(N_BINS)[:, None] * n[None, :]) % N_FFT) / N_FFT
    re = (fr * win) @ np.cos(ang).T
    im = (fr * win) @ np.sin(ang).T
    fb = melscale_fbanks(N_BINS, 0.0, float(sample_rate // 2), n_mels, sample_rate).double().numpy()
    return np.log2((re * re + im * im) @ fb + 1e-6)


def amplitude_to_db_power(x, top_db=80.0):
    """AmplitudeToDB('power', top_db): 10*log10(clamp(x,1e-10)) - 10*log10(max(1e-10, 1.0)); clamp to max-top_db."""
    x_db = 10.0 * torch.log10(torch.clamp(x, min=1e-10))
    x_db = x_db - 10.0 * math.log10(max(1e-10, 1.0))
    return torch.max(x_db, (x_db.amax() - top_db))


def mfcc(wave, sample_rate=16000, n_mfcc=40, n_mels=128):
    """torchaudio.transforms.MFCC(sample_rate, n_mfcc) -> [n_mfcc, frames]."""
    mel = amplitude_to_db_power(mel_spectrogram(wave, sample_rate, n_mels))
    dct = create_dct(n_mfcc, n_mels)
    return torch.matmul(mel.transpose(-1, -2), dct).transpose(-1, -2)


def extract_mfcc(wave, proj_w, proj_b, sample_rate=16000):
    """features/extractors.py:236-239 with the (random, never trained: SURVEY Q6) mfcc_proj injected."""
    m = mfcc(wave, sample_rate, 40)
    m = m.permute(1, 0) @ proj_w.t() + proj_b
    return m.detach().numpy().reshape(-1, 128)


def audio_forward_literal(waveform):
    """AudioFeatureExtractor.forward as it literally behaves (features/extractors.py:197-208):
    the [1,T] tensor has len() == 1 < 960, so every non-empty clip returns float64 zeros(296)."""
    if len(waveform) < 1:
        return np.zeros(296, dtype=np.float32)
    return np.zeros(296)


def resample_sinc(wave, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """Channel mean + torchaudio.functional.resample (sinc_interp_hann) restated with torch CPU ops [3P-memory,
    parity unpinned]: the step the reference leaves to ffmpeg (features/extractors.py:364-378, :326-328).
    wave: [T] or [T, channels] -> float32 [ceil(new*T/orig)]."""
    x = torch.as_tensor(wave, dtype=torch.float32)
    if x.dim() > 1:
        x = x.mean(dim=1)
    if int(orig_freq) == int(new_freq):
        return x
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = (torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * (base / orig)).float()
    length = x.numel()
    padded = torch.nn.functional.pad(x[None, None], (width, width + orig))
    out = torch.nn.functional.conv1d(padded, kern, stride=orig).transpose(1, 2).reshape(-1)
    return out[:math.ceil(new * length / orig)]
