#!/usr/bin/env python
"""register / LDS / scratch usage of every kernel in a .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
usage: tools/kernel_resources.py bmt_amd/csrc/gemm_bf16.hip [name filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "--cuda-device-only",
                    "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
for b in blocks:
    mangled = b.split()[0]
    name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    if flt and flt not in name:
        continue
    g = lambda k: (re.search(re.escape(k) + r": (\d+)", b) or [None, "?"])[1]
    print(f"{name[:70]:70s} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} scratch {g('ScratchSize [bytes/lane]'):>4} occ {g('Occupancy [waves/SIMD]')} LDS {g('LDS Size [bytes/block]')}")
if not blocks:
    print(r.stderr[-2000:])
