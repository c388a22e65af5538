#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) into HBM bytes per
launch per kernel, with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads => x2; both counters are in KiB.

    python profiles/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import re
import sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        full = r["Kernel_Name"].replace("void ", "")
        name = re.sub(r"<.*", "", full).split("(")[0]
        # the contraction kernel serves the bf16 convolutions (element size 2) AND the fp32 scorer GEMMs (4): keep
        # them apart, or the per-launch average mixes 180 convolutions with 7 small GEMMs
        m = re.match(r"igemm_kernel<(\d+)(?:, [^,>]+)*?(?:, (\d+))?>", full)
        if m:
            name = f"igemm_kernel<{m.group(1)}>"
            last = re.search(r", (\d+)(?:, (?:true|false)){0,2}>\(", full + "(")   # (trailing bools: nine-tap form, AVS_F16P8 input)
            if m.group(1) == "4" and last and last.group(1) == "2":
                name = "igemm_kernel<4,split2>"      # AVS_F16X2 convolutions (the last template argument)
                v = re.match(r"igemm_kernel<(\d+), (\d+), \w+, (\w+), (\d+), (\d+)", full)
                sub = f"igemm_kernel<4,split2,bn={v.group(2)},spatial={v.group(3)},epi={v.group(5)}>"
                if re.search(r", 2, true(?:, false)?>\(", full + "("):
                    sub = sub[:-1] + ",tap9>"
                if re.search(r", 2, false, true>\(", full + "("):
                    sub = sub[:-1] + ",p8in>"
                agg[sub][0] += 1
                agg[sub][1] += float(r["Counter_Value"])
        # round 3: the AVS_F16X2 tile-local form of 193..224-row groups runs on its own tile (csrc/local224.hip); it is
        # the same contraction behind the same entry point (avs_conv2d_nhwc_bnlocal) and counts into the same family
        m2 = re.match(r"igemm_h2_local224_kernel<(\w+)", full)
        if m2:
            sub = f"igemm_h2_local224_kernel<spatial={m2.group(1)}>"
            agg[sub][0] += 1
            agg[sub][1] += float(r["Counter_Value"])
            name = "igemm_kernel<4,split2>"
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(fetch, key=lambda k: -fetch[k][1]):
        n = fetch[k][0]
        w = write.get(k, [0, 0.0])[1]
        out[k] = {"launches": n, "fetch_kib_raw": round(fetch[k][1], 1), "write_kib": round(w, 1),
                  "hbm_bytes_per_launch": round((2.0 * fetch[k][1] + w) * 1024.0 / n, 1)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in list(out.items())[:10]:
        print(k, v)


if __name__ == "__main__":
    main()
