#!/usr/bin/env python3
"""Kernel study: the LSTM recurrence of the scorer (hidden 256, four recurrences, configs[1]'s 25 sequences) with the
generic kernel (W_hh^T streamed from L2 every step), the two resident splits (avs_lstm_f32's per-call variant) and one
recurrence split over four CUs (avs_lstm_split_f32: 100 recurrences = 400 workgroups, two rounds on 256 CUs); outputs must
be bit-identical.  Usage: python tools/lstm_study.py [videos]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avsum_amd import ops, _abi, synthetic

dev = torch.device("cuda", 0)
L = _abi.lib()
H, ND = 256, 4
lengths = synthetic.config(1, videos=int(sys.argv[1]) if len(sys.argv) > 1 else None)["lengths"]
offs = torch.tensor(synthetic.offsets_of(lengths), dtype=torch.int64, device=dev)
rows = int(offs[-1])
g = torch.Generator(device=dev).manual_seed(0)
xproj = torch.randn(rows, ND * 4 * H, device=dev, generator=g)
whh = torch.randn(ND, H, 4 * H, device=dev, generator=g) / H ** 0.5
ref = None
for mode in (1, 2, 3, 4):   # _abi.LSTM_STREAM, LSTM_RESIDENT_20_8, LSTM_RESIDENT_16_8, LSTM_SPLIT4 (per call)
    out = torch.zeros(rows, ND * H, device=dev)
    for _ in range(2):
        ops.lstm(xproj, whh, H, ND, 0b1010, offs, out, 0, variant=mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.lstm(xproj, whh, H, ND, 0b1010, offs, out, 0, variant=mode)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    if ref is None:
        ref = out.clone()
    print(f"mode {mode}: {ms:7.2f} ms, {ms * 1e3 / max(lengths):5.2f} us per step of the longest sequence ({max(lengths)}), "
          f"bit-identical to the generic kernel: {torch.equal(out, ref)}", flush=True)
print("split-recurrence waits that ran out:", ops.lstm_split_errors(dev))
