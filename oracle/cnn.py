"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the visual extractor.

Follows features/extractors.py:43-115 (VisualFeatureExtractor.forward),
:126-140 (_preprocess_frame: NO /255) and :142-155
(_preprocess_inception), with the two trunks restated from torchvision's
published architectures (third-party, absent from /root/reference and not
version-pinned: torchvision.models.resnet50 / inception_v3):

  * resnet50_trunk_forward  — v1.5 bottlenecks, BatchNorm in TRAIN mode, i.e.
    batch statistics over the <=4-frame micro-batch, because the reference only
    calls .eval() on the Inception net (extractors.py:29 vs :41);
  * inception_v3_forward    — eval-mode BatchNorm (eps 1e-3), transform_input
    (SURVEY Q4), fc = Identity, aux head unused.

cv2.resize is absent too; `cv_resize_linear_u8` restates OpenCV's 8-bit
INTER_LINEAR (11-bit fixed point).  PARITY UNPINNED for everything third-party
here: the reference holds no fixtures and its pretrained weights are not
available offline.  Pins: tests/test_oracle_analytic.py checks FLOP/parameter
counts (23 508 032 / 21 785 568 parameters), BN batch-stat identities, and
torch's own nn.BatchNorm2d(train) / F.conv2d as the arithmetic reference.
"""
import numpy as np
import torch
import torch.nn.functional as F

MEAN = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)


# ----------------------------------------------------------------------------- preprocessing
def cv_resize_linear_u8(img, dh, dw):
    """OpenCV resize(INTER_LINEAR) for uint8 HWC [3P-memory]: coefficients rounded to 11 bits,
    horizontal pass in int32, vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2 >> 2."""
    sh, sw = img.shape[:2]
    if (sh, sw) == (dh, dw):
        return img.copy()

    def coefs(dn, sn):
        scale = 1.0 / (dn / sn)
        d = np.arange(dn)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        low = s < 0
        f[low] = 0
        s[low] = 0
        high = s >= sn - 1
        f[high] = 0
        s[high] = sn - 1
        s1 = np.minimum(s + 1, sn - 1)
        c0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
        c1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
        return s, s1, c0, c1

    x0, x1, a0, a1 = coefs(dw, sw)
    y0, y1, b0, b1 = coefs(dh, sh)
    src = img.astype(np.int64)
    hor = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]  # [sh, dw, 3]
    s0, s1 = hor[y0], hor[y1]
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def preprocess_frame(frame):
    """features/extractors.py:126-140 -> [1,3,224,224]; values stay on the 0..255 scale."""
    frame = cv_resize_linear_u8(frame, 224, 224)
    t = torch.from_numpy(frame).permute(2, 0, 1).float()
    return ((t - MEAN) / STD).unsqueeze(0)


def preprocess_inception(frame):
    """features/extractors.py:142-155 -> [1,3,299,299]."""
    frame = cv_resize_linear_u8(frame, 299, 299)
    t = torch.from_numpy(frame).permute(2, 0, 1).float()
    return ((t / 255.0 - MEAN) / STD).unsqueeze(0)


# ----------------------------------------------------------------------------- ResNet-50 (train-mode BN)
def bn_batch(x, w, b, eps=1e-5):
    """nn.BatchNorm2d in training mode: per-channel mean / biased variance over (N,H,W)."""
    mean = x.mean(dim=(0, 2, 3), keepdim=True)
    var = x.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * w.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def bn_eval(x, sd, p, eps):
    s = sd[p + "weight"] / torch.sqrt(sd[p + "running_var"] + eps)
    return x * s.view(1, -1, 1, 1) + (sd[p + "bias"] - sd[p + "running_mean"] * s).view(1, -1, 1, 1)


def resnet50_trunk_forward(sd, x, bn_mode="batch"):
    """sd: state_dict of nn.Sequential(*resnet50.children()[:-1]); x [N,3,224,224] (one micro-batch) -> [N,2048]."""
    def bn(t, p):
        return bn_batch(t, sd[p + "weight"], sd[p + "bias"]) if bn_mode == "batch" else bn_eval(t, sd, p, 1e-5)

    x = F.conv2d(x, sd["0.weight"], None, 2, 3)
    x = torch.relu(bn(x, "1."))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (blocks, stride) in zip(range(4, 8), ((3, 1), (4, 2), (6, 2), (3, 2))):
        for bi in range(blocks):
            p = f"{li}.{bi}."
            s = stride if bi == 0 else 1
            idn = x
            y = torch.relu(bn(F.conv2d(x, sd[p + "conv1.weight"]), p + "bn1."))
            y = torch.relu(bn(F.conv2d(y, sd[p + "conv2.weight"], None, s, 1), p + "bn2."))
            y = bn(F.conv2d(y, sd[p + "conv3.weight"]), p + "bn3.")
            if p + "downsample.0.weight" in sd:
                idn = bn(F.conv2d(x, sd[p + "downsample.0.weight"], None, s), p + "downsample.1.")
            x = torch.relu(y + idn)
    return F.adaptive_avg_pool2d(x, (1, 1)).flatten(1)


# ----------------------------------------------------------------------------- Inception-v3 (eval-mode BN)
def _bc(sd, p, x, stride=1, padding=0):
    x = F.conv2d(x, sd[p + ".conv.weight"], None, stride, padding)
    return torch.relu(bn_eval(x, sd, p + ".bn.", 1e-3))


def _inc_a(sd, p, x):
    b1 = _bc(sd, p + ".branch1x1", x)
    b5 = _bc(sd, p + ".branch5x5_2", _bc(sd, p + ".branch5x5_1", x), padding=2)
    b3 = _bc(sd, p + ".branch3x3dbl_1", x)
    b3 = _bc(sd, p + ".branch3x3dbl_2", b3, padding=1)
    b3 = _bc(sd, p + ".branch3x3dbl_3", b3, padding=1)
    bp = _bc(sd, p + ".branch_pool", F.avg_pool2d(x, 3, 1, 1))
    return torch.cat([b1, b5, b3, bp], 1)


def _inc_b(sd, p, x):
    b3 = _bc(sd, p + ".branch3x3", x, stride=2)
    bd = _bc(sd, p + ".branch3x3dbl_1", x)
    bd = _bc(sd, p + ".branch3x3dbl_2", bd, padding=1)
    bd = _bc(sd, p + ".branch3x3dbl_3", bd, stride=2)
    return torch.cat([b3, bd, F.max_pool2d(x, 3, 2)], 1)


def _inc_c(sd, p, x):
    b1 = _bc(sd, p + ".branch1x1", x)
    b7 = _bc(sd, p + ".branch7x7_1", x)
    b7 = _bc(sd, p + ".branch7x7_2", b7, padding=(0, 3))
    b7 = _bc(sd, p + ".branch7x7_3", b7, padding=(3, 0))
    bd = _bc(sd, p + ".branch7x7dbl_1", x)
    bd = _bc(sd, p + ".branch7x7dbl_2", bd, padding=(3, 0))
    bd = _bc(sd, p + ".branch7x7dbl_3", bd, padding=(0, 3))
    bd = _bc(sd, p + ".branch7x7dbl_4", bd, padding=(3, 0))
    bd = _bc(sd, p + ".branch7x7dbl_5", bd, padding=(0, 3))
    bp = _bc(sd, p + ".branch_pool", F.avg_pool2d(x, 3, 1, 1))
    return torch.cat([b1, b7, bd, bp], 1)


def _inc_d(sd, p, x):
    b3 = _bc(sd, p + ".branch3x3_2", _bc(sd, p + ".branch3x3_1", x), stride=2)
    b7 = _bc(sd, p + ".branch7x7x3_1", x)
    b7 = _bc(sd, p + ".branch7x7x3_2", b7, padding=(0, 3))
    b7 = _bc(sd, p + ".branch7x7x3_3", b7, padding=(3, 0))
    b7 = _bc(sd, p + ".branch7x7x3_4", b7, stride=2)
    return torch.cat([b3, b7, F.max_pool2d(x, 3, 2)], 1)


def _inc_e(sd, p, x):
    b1 = _bc(sd, p + ".branch1x1", x)
    b3 = _bc(sd, p + ".branch3x3_1", x)
    b3 = torch.cat([_bc(sd, p + ".branch3x3_2a", b3, padding=(0, 1)),
                    _bc(sd, p + ".branch3x3_2b", b3, padding=(1, 0))], 1)
    bd = _bc(sd, p + ".branch3x3dbl_2", _bc(sd, p + ".branch3x3dbl_1", x), padding=1)
    bd = torch.cat([_bc(sd, p + ".branch3x3dbl_3a", bd, padding=(0, 1)),
                    _bc(sd, p + ".branch3x3dbl_3b", bd, padding=(1, 0))], 1)
    bp = _bc(sd, p + ".branch_pool", F.avg_pool2d(x, 3, 1, 1))
    return torch.cat([b1, b3, bd, bp], 1)


def inception_v3_forward(sd, x, transform_input=True):
    """sd: state_dict of torchvision Inception3 with fc = Identity; x [N,3,299,299] -> [N,2048]."""
    if transform_input:
        c0 = torch.unsqueeze(x[:, 0], 1) * (0.229 / 0.5) + (0.485 - 0.5) / 0.5
        c1 = torch.unsqueeze(x[:, 1], 1) * (0.224 / 0.5) + (0.456 - 0.5) / 0.5
        c2 = torch.unsqueeze(x[:, 2], 1) * (0.225 / 0.5) + (0.406 - 0.5) / 0.5
        x = torch.cat((c0, c1, c2), 1)
    x = _bc(sd, "Conv2d_1a_3x3", x, stride=2)
    x = _bc(sd, "Conv2d_2a_3x3", x)
    x = _bc(sd, "Conv2d_2b_3x3", x, padding=1)
    x = F.max_pool2d(x, 3, 2)
    x = _bc(sd, "Conv2d_3b_1x1", x)
    x = _bc(sd, "Conv2d_4a_3x3", x)
    x = F.max_pool2d(x, 3, 2)
    for name in ("Mixed_5b", "Mixed_5c", "Mixed_5d"):
        x = _inc_a(sd, name, x)
    x = _inc_b(sd, "Mixed_6a", x)
    for name in ("Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"):
        x = _inc_c(sd, name, x)
    x = _inc_d(sd, "Mixed_7a", x)
    x = _inc_e(sd, "Mixed_7b", x)
    x = _inc_e(sd, "Mixed_7c", x)
    return F.adaptive_avg_pool2d(x, (1, 1)).flatten(1)


# ----------------------------------------------------------------------------- extractor
def visual_forward(resnet_sd, inception_sd, frames, batch_size=4):
    """VisualFeatureExtractor.forward, features/extractors.py:43-115: micro-batches of 4, ResNet in
    train-mode BN per micro-batch, mean over the shot's frames, concat -> float32 [4096]."""
    if len(frames) == 0:
        return np.zeros(4096, dtype=np.float32)
    res, inc = [], []
    with torch.no_grad():
        for i in range(0, len(frames), batch_size):
            batch = frames[i:i + batch_size]
            rb = torch.cat([preprocess_frame(f) for f in batch])
            res.append(resnet50_trunk_forward(resnet_sd, rb).numpy())
            ib = torch.cat([preprocess_inception(f) for f in batch])
            inc.append(inception_v3_forward(inception_sd, ib).numpy())
    res_all = np.concatenate(res, 0)
    inc_all = np.concatenate(inc, 0)
    return np.concatenate([res_all.mean(axis=0), inc_all.mean(axis=0)])
