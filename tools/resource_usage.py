#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/resource_usage.py igemm.hip [substring filter of the demangled name]
"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                    "audiovidsum-a-multi-modal-approach-to-video-summarization_amd", "csrc")


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    txt = subprocess.run(cmd, cwd=os.path.realpath(CSRC), capture_output=True, text=True).stderr
    recs, cur = [], None
    for line in txt.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            recs.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in recs), capture_output=True,
                           text=True).stdout.splitlines()
    for r, d in zip(recs, names):
        d = d.replace("void ", "").replace("(IgemmParams)", "")
        if flt in d:
            print(f"{d[:100]:100s} V {r.get('VGPRs'):>3} A {r.get('AGPRs'):>3} scratch {r.get('ScratchSize'):>4} "
                  f"sgpr-spill {r.get('SGPRs Spill'):>3} occ {r.get('Occupancy')} lds {r.get('LDS Size')}")


if __name__ == "__main__":
    main()
