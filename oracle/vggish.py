"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the VGGish branch of the audio extractor.

Reference call site: features/extractors.py:188 (``torch.hub.load("harritaylor/torchvggish", "vggish")``) and :216
(``self.vggish(waveform)``, dead code at run time: SURVEY Q5).  The model is a third-party dependency that is NOT in
/root/reference, is not version-pinned, and whose weights are a download: restated here from the published algorithm
(torchvggish: vggish_input.py / mel_features.py / vggish.py / vggish_params.py) [3P-memory] => PARITY UNPINNED.
Pinned by analytic known answers in tests/test_oracle_analytic.py: 72 141 184 parameters, framing arithmetic
(one 0.96 s example per 15 360 samples after the first 15 600), mel-matrix structure (64 bands 125-7500 Hz, DC row
zero, every bin feeds at most two bands), log-mel of a sine, quantiser end points.

Front end (float64, as the numpy original): 25 ms periodic-Hann window, 10 ms hop, 512-point rFFT MAGNITUDE,
64-band HTK-mel matrix 125..7500 Hz, log(mel + 0.01), non-overlapping 96-frame examples.
Network: 6 conv3x3+ReLU with 4 max-pools -> [n,512,6,4] -> flatten in (h, w, c) order -> 3 x (Linear + ReLU) -> 128.
Post-processor: PCA (eigen-vectors, means), clip to [-2, 2], quantise to 0..255 (kept as float).
"""
import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000
WINDOW = 400          # 0.025 s
HOP = 160             # 0.010 s
FFT_LEN = 512         # 2 ** ceil(log2(400))
NUM_BINS = FFT_LEN // 2 + 1
NUM_MEL = 64
MEL_MIN_HZ, MEL_MAX_HZ = 125.0, 7500.0
LOG_OFFSET = 0.01
EXAMPLE_FRAMES = 96   # 0.96 s of 10 ms frames, hop 0.96 s
QUANT_MIN, QUANT_MAX = -2.0, 2.0
CONV_PLAN = ((0, 1, 64, True), (3, 64, 128, True), (6, 128, 256, False), (8, 256, 256, True), (11, 256, 512, False),
             (13, 512, 512, True))   # (features.N index, cin, cout, max-pool after)
FC_PLAN = ((0, 512 * 4 * 6, 4096), (2, 4096, 4096), (4, 4096, 128))


def hertz_to_mel(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_matrix():
    """mel_features.spectrogram_to_mel_matrix -> float64 [257, 64]."""
    nyquist = SAMPLE_RATE / 2.0
    bins_mel = hertz_to_mel(np.linspace(0.0, nyquist, NUM_BINS))
    edges = np.linspace(hertz_to_mel(MEL_MIN_HZ), hertz_to_mel(MEL_MAX_HZ), NUM_MEL + 2)
    w = np.empty((NUM_BINS, NUM_MEL))
    for i in range(NUM_MEL):
        lower, center, upper = edges[i:i + 3]
        lower_slope = (bins_mel - lower) / (center - lower)
        upper_slope = (upper - bins_mel) / (upper - center)
        w[:, i] = np.maximum(0.0, np.minimum(lower_slope, upper_slope))
    w[0, :] = 0.0
    return w


def periodic_hann(n=WINDOW):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi / n * np.arange(n))


def num_stft_frames(t):
    return 0 if t < WINDOW else 1 + (t - WINDOW) // HOP


def num_examples(t):
    f = num_stft_frames(t)
    return 0 if f < EXAMPLE_FRAMES else 1 + (f - EXAMPLE_FRAMES) // EXAMPLE_FRAMES


def log_mel_spectrogram(wave):
    """float64 [frames, 64] = log(|rfft_512(hann * frame)| . mel_matrix + 0.01)."""
    x = np.asarray(wave, dtype=np.float64)
    frames = num_stft_frames(len(x))
    if frames == 0:
        return np.zeros((0, NUM_MEL))
    idx = np.arange(WINDOW)[None, :] + HOP * np.arange(frames)[:, None]
    mag = np.abs(np.fft.rfft(x[idx] * periodic_hann(), FFT_LEN))
    return np.log(mag @ mel_matrix() + LOG_OFFSET)


def waveform_to_examples(wave, sample_rate=SAMPLE_RATE):
    """vggish_input.waveform_to_examples -> float32 tensor [n, 1, 96, 64]."""
    x = np.asarray(wave)
    if x.ndim > 1:
        x = x.mean(axis=1)
    if sample_rate != SAMPLE_RATE:
        raise ValueError("the oracle takes 16 kHz input (resampy is not restated)")
    lm = log_mel_spectrogram(x)
    n = 0 if lm.shape[0] < EXAMPLE_FRAMES else 1 + (lm.shape[0] - EXAMPLE_FRAMES) // EXAMPLE_FRAMES
    ex = np.stack([lm[i * EXAMPLE_FRAMES:(i + 1) * EXAMPLE_FRAMES] for i in range(n)]) if n else \
        np.zeros((0, EXAMPLE_FRAMES, NUM_MEL))
    return torch.from_numpy(ex).float()[:, None, :, :]


def network(sd, examples):
    """VGG.forward on [n,1,96,64] with the torchvggish state-dict keys -> [n,128]."""
    x = examples
    for idx, _, _, pool in CONV_PLAN:
        x = F.relu(F.conv2d(x, sd[f"features.{idx}.weight"], sd[f"features.{idx}.bias"], padding=1))
        if pool:
            x = F.max_pool2d(x, 2, 2)
    x = torch.transpose(x, 1, 3)
    x = torch.transpose(x, 1, 2)
    x = x.contiguous().view(x.size(0), -1)
    for idx, _, _ in FC_PLAN:
        x = F.relu(F.linear(x, sd[f"embeddings.{idx}.weight"], sd[f"embeddings.{idx}.bias"]))
    return x


def postprocess(sd, emb):
    """Postprocessor.postprocess: PCA, clip, 8-bit quantise (values 0..255 as float)."""
    pca = torch.mm(sd["pproc.pca_eigen_vectors"], (emb.t() - sd["pproc.pca_means"])).t()
    clipped = torch.clamp(pca, QUANT_MIN, QUANT_MAX)
    return torch.round((clipped - QUANT_MIN) * (255.0 / (QUANT_MAX - QUANT_MIN)))


def vggish_forward(sd, wave, sample_rate=SAMPLE_RATE, postprocess_output=True):
    with torch.no_grad():
        emb = network(sd, waveform_to_examples(wave, sample_rate))
        return postprocess(sd, emb) if postprocess_output else emb
