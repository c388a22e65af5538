#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass of SQ / GRBM counters (--output-format csv, with --kernel-trace) per kernel:
matrix-core busy share and where the waves' cycles went.

    python profiles/summarize_sq.py <counter_collection.csv> <out.json>

Units (/opt/skills/guides/MI355X_MICROARCH.md, 'Per-instruction cycle constants'): SQ_WAVE_CYCLES, SQ_WAIT_*,
SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs
(32 per v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE is the sum over the 8 XCDs.  So
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
is the fraction of SIMD-cycles in which the matrix pipe was busy (1.0 = the dense peak at the clock the kernel ran at),
and wait / active shares are fractions of the wave-cycles (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES).
"""
import collections
import csv
import json
import re
import sys


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(sys.argv[1])):
        full = r["Kernel_Name"].replace("void ", "")
        name = re.sub(r"<.*", "", full).split("(")[0]
        m = re.match(r"igemm_kernel<(\d+), (\d+), \w+, (\w+), (\d+), (\d+)", full)
        if m:   # element size, column tile, spatial, row bytes, epilogue form
            name = f"igemm_kernel<es={m.group(1)},bn={m.group(2)},spatial={m.group(3)},epi={m.group(5)}>"
            last = re.search(r", (\d+)(?:, (?:true|false)){0,2}>\(", full + "(")   # (trailing bools: nine-tap form, AVS_F16P8 input)
            if m.group(1) == "4" and last and last.group(1) in ("1", "2"):
                name = name[:-1] + f",split={last.group(1)}>"
                if re.search(r", 2, true(?:, false)?>\(", full + "("):
                    name = name[:-1] + ",tap9>"
                if re.search(r", 2, false, true>\(", full + "("):
                    name = name[:-1] + ",p8in>"
        m2 = re.match(r"igemm_h2_local224_kernel<(\w+)", full)
        if m2:   # the 224-row tile-local form (AVS_F16X2): spatial or not
            name = f"igemm_h2_local224_kernel<spatial={m2.group(1)}>"
        if not name.startswith(("igemm", "conv1x1", "bn_", "lstm", "stft", "frames_", "global_avg", "power_mel", "stem")):
            continue
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name].add(r["Dispatch_Id"])
    out = {}
    for name, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)):
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        rec = {"launches": len(calls[name]), "counters": {k: round(v, 1) for k, v in sorted(c.items())}}
        if gui > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            rec["mfma_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 1024.0), 4)
        if wc > 0:
            for key, label in (("SQ_WAIT_ANY", "wait_share"), ("SQ_WAIT_INST_ANY", "issue_stall_share"),
                               ("SQ_ACTIVE_INST_ANY", "active_share")):
                if key in c:
                    rec[label] = round(c[key] / wc, 4)
        out[name] = rec
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for k, v in list(out.items())[:12]:
        print(k, {kk: vv for kk, vv in v.items() if kk != "counters"})


if __name__ == "__main__":
    main()
