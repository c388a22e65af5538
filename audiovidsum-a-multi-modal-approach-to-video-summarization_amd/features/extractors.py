"""Drop-in for the reference's features/extractors.py on the MI355X.

Same classes, attributes, method names and return types:

  VisualFeatureExtractor   extractors.py:21-155   .resnet / .inception, forward(list[uint8 HxWx3]) -> f32[4096]
  AudioFeatureExtractor    extractors.py:184-290  .sr / .vggish / .mfcc_proj, forward(np[T]) -> np[296]
  AVProcessor              extractors.py:298-413  process_video(path) -> (np[S,4096], np[S,296])

Deviations forced by the environment: constructors never touch the
network.  Weights come from ``load_state_dict`` (torchvision-compatible keys) or a seeded
synthetic init (``.vggish`` included: the torch.hub model's module tree, vggish.py).  All
arithmetic runs through libavsum_hip.so — there is no CPU fallback.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..audio import MelPlan
from ..cnn import InceptionV3Runner, Inception3, ResNet50Runner, RESNET_MEAN, RESNET_STD, resnet50_trunk
from ..vggish import VGGish

FRAME_INTERVAL = 3  # extractors.py:399
MAX_FRAMES = 100    # extractors.py:400
MICRO_BATCH = 4     # extractors.py:48


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("avsum_amd needs an MI355X (HIP device); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _as_bgr_u8(frame):
    """extractors.py:129-130 / :408-411: force 3 channels."""
    frame = np.asarray(frame)
    if frame.dtype != np.uint8:
        raise TypeError(f"frames must be uint8, got {frame.dtype}")
    if frame.ndim == 2:
        frame = frame[:, :, None]
    if frame.shape[-1] == 1:
        frame = np.repeat(frame, 3, axis=-1)  # cv2.COLOR_GRAY2BGR
    elif frame.shape[-1] == 4:
        frame = frame[:, :, :3]  # cv2.COLOR_BGRA2BGR (drop alpha)
    if frame.shape[-1] != 3:
        raise ValueError(f"cannot interpret frame of shape {frame.shape}")
    return np.ascontiguousarray(frame)


class VisualFeatureExtractor(nn.Module):
    ARITHMETIC = {"f32": (torch.float32, False), "f16x2": (torch.float32, "f16x2"), "f32split": (torch.float32, True),
                  "bf16": (torch.bfloat16, False)}

    def __init__(self, dtype=torch.float32, bn_mode="batch", f32_split=False, arith=None):
        """The arithmetic of the two trunks, by name (`arith`) or by its parts (`dtype`, `f32_split`):
          "f32"       exact fp32 MFMA (the default: the most literal parity, 10 k frames/s on one MI355X);
          "f16x2"     values stored as fp16 hi | lo runs, three fp16 MFMAs per product: fp32-class results (meets the
                      accuracy bars) at 23.6 k frames/s - what bench.py times;  = (torch.float32, f32_split="f16x2");
          "f32split"  fp32 storage, operands split to bf16 hi + lo in the loop;  = (torch.float32, f32_split=True);
          "bf16"      bf16 storage and MFMA: the throughput mode (52 k frames/s), NOT parity-grade;  = (torch.bfloat16)."""
        super().__init__()
        if arith is not None:
            if arith not in self.ARITHMETIC:
                raise ValueError(f"arith must be one of {sorted(self.ARITHMETIC)}")
            dtype, f32_split = self.ARITHMETIC[arith]
        self.resnet = resnet50_trunk()      # nn.Sequential(*resnet50.children()[:-1]), extractors.py:25,29
        self.inception = Inception3()       # fc = Identity, avgpool adaptive, extractors.py:26,32-36
        self.inception.aux_logits = False   # extractors.py:36
        for p in self.inception.parameters():
            p.requires_grad = False         # extractors.py:39-40
        self.inception.eval()               # extractors.py:41 (the ResNet trunk stays in train mode: SURVEY Q2)
        self.compute_dtype = dtype
        self._resnet_runner = ResNet50Runner(self.resnet, dtype, bn_mode, f32_split)
        self._inception_runner = InceptionV3Runner(self.inception, dtype, f32_split)

    def _to_device_u8(self, frames):
        dev = _device()
        shapes = {f.shape for f in frames}
        if len(shapes) != 1:
            raise ValueError("all frames of one call must share a size")
        arr = np.stack(frames)
        return torch.from_numpy(arr).to(dev)

    def embed(self, frames_u8, group_frames=None):
        """Device uint8 [N,H,W,3] -> (resnet fp32 [N,2048], inception fp32 [N,2048]) on device."""
        n, h, w, _ = frames_u8.shape
        r_in = frames_u8 if (h, w) == (224, 224) else ops.resize_bilinear(frames_u8, 224, 224)
        res = self._resnet_runner.forward(r_in, group_frames)
        i_in = frames_u8 if (h, w) == (299, 299) else ops.resize_bilinear(frames_u8, 299, 299)
        inc = self._inception_runner.forward(i_in)
        return res, inc

    def forward(self, frames):
        if len(frames) == 0:
            return np.zeros(4096, dtype=np.float32)
        frames = [_as_bgr_u8(f) for f in frames]
        dev_frames = self._to_device_u8(frames)
        n = dev_frames.shape[0]
        groups = list(range(0, n, MICRO_BATCH)) + [n]
        with torch.no_grad():
            res, inc = self.embed(dev_frames, groups)
            seg = torch.tensor([0, n], dtype=torch.int64, device=res.device)
            out = torch.empty((1, 4096), dtype=torch.float32, device=res.device)
            ops.segment_mean(res, seg, out[:, :2048])
            ops.segment_mean(inc, seg, out[:, 2048:])
        return out[0].cpu().numpy()

    def _preprocess_frame(self, frame):
        """extractors.py:126-140 -> Tensor [1,3,224,224] (resize, (x-mean)/std, no /255)."""
        f = self._to_device_u8([_as_bgr_u8(frame)])
        if f.shape[1:3] != (224, 224):
            f = ops.resize_bilinear(f, 224, 224)
        x = ops.frames_normalize(f, torch.float32, 1.0, RESNET_MEAN, RESNET_STD, 224, 224, 0, 0)
        return x[..., :3].permute(0, 3, 1, 2).contiguous().cpu()

    def _preprocess_inception(self, frame):
        """extractors.py:142-155 -> Tensor [1,3,299,299]."""
        f = self._to_device_u8([_as_bgr_u8(frame)])
        if f.shape[1:3] != (299, 299):
            f = ops.resize_bilinear(f, 299, 299)
        x = ops.frames_normalize(f, torch.float32, 255.0, RESNET_MEAN, RESNET_STD, 299, 299, 0, 0)
        return x[..., :3].permute(0, 3, 1, 2).contiguous().cpu()


class AudioFeatureExtractor(nn.Module):
    def __init__(self, sr=16000, strict_reference=True):
        super().__init__()
        self.sr = sr
        # extractors.py:188 fetches the model with torch.hub (network: never attempted); this is the same
        # module tree (load_state_dict-compatible) running on the MI355X, seeded synthetic weights until loaded
        self.vggish = VGGish()
        self.vggish.eval()
        for p in self.vggish.parameters():
            p.requires_grad = False          # extractors.py:191-192
        self.mfcc_proj = nn.Linear(40, 128)  # random, never trained (extractors.py:193)
        self.strict_reference = strict_reference

    def forward(self, waveform):
        if len(waveform) < 1:
            return np.zeros(296, dtype=np.float32)  # extractors.py:197-198
        if self.strict_reference:
            # extractors.py:199,207-208: the tensor is [1,T] so len() is 1 < 960 and the function
            # returns float64 zeros for EVERY non-empty clip.
            return np.zeros(296)
        # "intent" mode — documented deviation, never used for parity claims
        wave = torch.from_numpy(np.asarray(waveform)).float()
        if wave.numel() < 960:
            wave = torch.nn.functional.pad(wave, (0, 960 - wave.numel()))
        wave = wave.clamp(-1, 1).to(_device())
        plan = MelPlan.get(self.sr, 128, 40, wave.device)
        seg_m = torch.tensor([0, plan.num_frames(wave.numel())], dtype=torch.int64, device=wave.device)
        mfcc_mean = ops.segment_mean(plan.mfcc(wave), seg_m)
        mel_mean = ops.segment_mean(plan.log2_mel(wave), seg_m)
        out = np.zeros(296, dtype=np.float32)
        out[:40] = mfcc_mean[0].cpu().numpy()
        out[40:168] = mel_mean[0].cpu().numpy()
        vg = self.vggish(wave.cpu().numpy(), self.sr)       # [n,128] / [128] / [0,128] (extractors.py:216)
        vg = vg.reshape(-1, 128)
        if vg.shape[0] > 0:
            seg_v = torch.tensor([0, vg.shape[0]], dtype=torch.int64, device=vg.device)
            out[168:] = ops.segment_mean(vg.contiguous(), seg_v)[0].cpu().numpy()   # vggish_feats.mean(0), :234
        return out

    def _extract_mfcc(self, waveform):
        """extractors.py:236-239: MFCC(sr, n_mfcc=40) -> permute -> mfcc_proj -> np [time,128]."""
        wave = torch.as_tensor(waveform).float().to(_device())
        plan = MelPlan.get(self.sr, 128, 40, wave.device)
        proj = self.mfcc_proj.to(wave.device)
        with torch.no_grad():
            out = ops.linear(plan.mfcc(wave), proj.weight.float().contiguous(), proj.bias.float().contiguous())
        return out.cpu().numpy().reshape(-1, 128)

    def _extract_mel(self, waveform):
        """extractors.py:241-246: log2(MelSpectrogram(sr, n_mels=128) + 1e-6) -> np [time,128]."""
        wave = torch.as_tensor(waveform).float().to(_device())
        plan = MelPlan.get(self.sr, 128, 40, wave.device)
        return plan.log2_mel(wave).cpu().numpy().reshape(-1, 128)

    def _align_features(self, mfcc, mel, vggish):
        """extractors.py:248-290.  The reference's fastdtw(..., dist=cdist) call raises and falls back to
        zeros; this build aligns with exact DTW on the Euclidean cost matrix (documented)."""
        from .fusion import compute_dtw, compute_optimal_path
        vggish, mfcc, mel = np.atleast_2d(vggish), np.atleast_2d(mfcc), np.atleast_2d(mel)
        if vggish.size == 0 or mfcc.size == 0 or mel.size == 0:
            return np.zeros(128), np.zeros(128)
        dim = min(vggish.shape[1], mfcc.shape[1], mel.shape[1])
        vggish, mfcc, mel = vggish[:, :dim], mfcc[:, :dim], mel[:, :dim]
        length = min(vggish.shape[0], mfcc.shape[0], mel.shape[0])
        if length == 0:
            return np.zeros(dim), np.zeros(dim)
        vggish, mfcc, mel = vggish[:length], mfcc[:length], mel[:length]
        out = []
        for feat in (mfcc, mel):
            cost = compute_dtw(torch.from_numpy(np.ascontiguousarray(vggish, dtype=np.float32)),
                               torch.from_numpy(np.ascontiguousarray(feat, dtype=np.float32)))
            path = compute_optimal_path(cost)
            out.append(np.array([feat[p[1]] for p in path]))
        return out[0], out[1]


def sample_shot_indices(start, end):
    """extractors.py:395-413: absolute frame index % 3 == 0, at most 100 frames per shot."""
    idx = []
    for i in range(start, end):
        if len(idx) >= MAX_FRAMES:
            break
        if i % FRAME_INTERVAL == 0:
            idx.append(i)
    return idx


class AVProcessor:
    def __init__(self, dtype=torch.float32, bn_mode="batch", strict_reference=True, arith=None):
        """arith: the visual trunks' arithmetic by name (VisualFeatureExtractor: "f32" default, "f16x2", ...)."""
        self.visual_extractor = VisualFeatureExtractor(dtype, bn_mode, arith=arith)
        self.audio_extractor = AudioFeatureExtractor(strict_reference=strict_reference)
        self.sr = self.audio_extractor.sr

    def prepare_audio(self, samples, sample_rate):
        """extractors.py:364-378 + :326-328 without ffmpeg: interleaved PCM [T] / [T, channels] at `sample_rate`
        -> mono float32 numpy at self.sr (channel mean, polyphase resampling on the MI355X)."""
        from ..audio import resample_to
        x = np.asarray(samples)
        if np.issubdtype(x.dtype, np.integer):
            x = x.astype(np.float32) / float(1 << (8 * x.dtype.itemsize - 1))
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(_device())
        return resample_to(x, sample_rate, self.sr).cpu().numpy()

    def _detect_shots_decoded(self, frames):
        """extractors.py:388-393 on decoded frames: ContentDetector restated on the GPU (features/shots.py)."""
        from .shots import detect_shots
        return detect_shots(np.stack([_as_bgr_u8(f) for f in frames]))

    def process_decoded(self, frames, waveform, fps, shots=None):
        """The per-shot loop of process_video (extractors.py:344-362) on already-decoded input:
        frames = indexable of uint8 HxWx3 for the whole video, waveform = mono float array at self.sr,
        shots = [(start_frame, end_frame)] or None to run the shot detector on the frames."""
        if shots is None:
            shots = self._detect_shots_decoded(frames)
        visual, audio = [], []
        for start, end in shots:
            picked = [frames[i] for i in sample_shot_indices(start, min(end, len(frames)))]
            visual.append(self.visual_extractor(picked))
            s0, s1 = int(start / fps * self.sr), int(end / fps * self.sr)
            audio.append(self.audio_extractor(waveform[s0:s1]))
        return np.array(visual), np.array(audio)

    def process_video(self, video_path):
        """extractors.py:304-362: (np[S,4096], np[S,296]) for the shots of one video file.  Decode and audio demux
        are host-side third-party code (cv2, pydub/ffmpeg: SURVEY row A7); everything after the decoded frames and
        PCM samples runs on the MI355X."""
        import os
        import shutil
        import tempfile
        try:
            import cv2
        except ImportError as e:
            raise RuntimeError(
                f"process_video needs cv2 to decode {video_path} ({e}); "
                "use process_decoded(frames, waveform, fps, shots) with decoded input") from e
        cap = cv2.VideoCapture(video_path)
        fps = cap.get(cv2.CAP_PROP_FPS)
        temp_dir = tempfile.mkdtemp()
        audio_path = os.path.join(temp_dir, "audio.wav")
        try:
            try:
                self._extract_audio(video_path, audio_path)
            except Exception as e:
                raise RuntimeError(f"Failed to extract audio from {video_path}") from e
            if not os.path.exists(audio_path):
                raise FileNotFoundError(f"Audio extraction failed for {video_path}")
            waveform = self._load_wav_mono(audio_path)       # torchaudio.load(...).mean(0), extractors.py:326-328
        finally:
            shutil.rmtree(temp_dir, ignore_errors=True)
        shots = self._detect_shots(video_path)
        visual, audio = [], []
        for start, end in shots:
            visual.append(self.visual_extractor(self._extract_frames(cap, start, end)))
            audio.append(self.audio_extractor(waveform[int(start / fps * self.sr):int(end / fps * self.sr)]))
        cap.release()
        return np.array(visual), np.array(audio)

    @staticmethod
    def _load_wav_mono(audio_path):
        """PCM WAV -> float32 [T] in [-1, 1), channel mean (what torchaudio.load + .mean(dim=0) give, :326-328)."""
        import wave
        with wave.open(audio_path, "rb") as wf:
            width, channels = wf.getsampwidth(), wf.getnchannels()
            raw = wf.readframes(wf.getnframes())
        if width == 1:      # 8-bit WAV is unsigned
            pcm = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif width in (2, 4):
            pcm = np.frombuffer(raw, dtype=np.int16 if width == 2 else np.int32).astype(np.float32) / float(1 << (8 * width - 1))
        else:
            raise RuntimeError(f"unsupported WAV sample width {width}")
        return pcm.reshape(-1, channels).mean(axis=1)

    def _extract_audio(self, video_path, audio_path):
        """extractors.py:364-386: demux with pydub/ffmpeg -> mono -> 16 kHz -> WAV at audio_path.  Every failure
        (pydub or ffmpeg missing included) surfaces as RuntimeError("Audio extraction failed: ..."), as there."""
        try:
            from pydub import AudioSegment
            track = AudioSegment.from_file(video_path).set_channels(1).set_frame_rate(16000)
            track.export(audio_path, format="wav", bitrate="256k")
            AudioSegment.from_wav(audio_path)   # the reference re-opens the export as its check (:380)
        except Exception as e:
            raise RuntimeError(f"Audio extraction failed: {str(e)}")

    def _detect_shots(self, video_path):
        """extractors.py:388-393: [(start_frame, end_frame)] from PySceneDetect's ContentDetector.  Where
        PySceneDetect is not installed the same cut rule runs on the MI355X (features/shots.py) over
        the frames cv2 decodes."""
        try:
            from scenedetect import ContentDetector, detect
        except ImportError:
            import cv2
            from .shots import detect_shots
            cap = cv2.VideoCapture(video_path)
            frames = []
            while True:
                ok, frame = cap.read()
                if not ok:
                    break
                frames.append(_as_bgr_u8(frame))
            cap.release()
            return detect_shots(np.stack(frames)) if frames else []
        return [(a.get_frames(), b.get_frames()) for a, b in detect(video_path, ContentDetector())]

    def _extract_frames(self, cap, start, end):
        """extractors.py:395-413: seek to `start`, read frames start..end-1, keep those whose ABSOLUTE index is a
        multiple of 3, stop at a failed read or once 100 are kept; grey / BGRA frames become 3-channel BGR.
        `cap` is anything with cv2.VideoCapture's set / read."""
        CAP_PROP_POS_FRAMES = 1   # cv2's constant (cv2 itself is not needed here)
        cap.set(CAP_PROP_POS_FRAMES, start)
        frames = []
        for index in range(start, end):
            ok, frame = cap.read()
            if not ok or len(frames) >= MAX_FRAMES:
                break
            if index % FRAME_INTERVAL == 0:
                frames.append(_as_bgr_u8(frame))
        return frames
