#!/usr/bin/env python
"""LDS-DMA (`buffer_load_dwordx4 ... lds`) instructions per kernel instantiation in the compiled gfx950 ISA of the translation units whose kernels retire their
rings with COUNTED `s_waitcnt vmcnt(N)` (conv_g8.h, conv_h3.h, conv_pwk.h, conv_h3b.h, conv_bneck.h, conv_headk.h, conv_sppf.h).  A counted wait is only right
while every wave issues exactly the number of loads the count assumes; the compiler once merged identical dummy loads (profiles/r05/r05_dummy_dma_merge.log),
which no host-side emulation can see.  `python scripts/isa_counts.py --write` regenerates tests/golden/isa_lds_dma_counts.json (tests/test_isa_dummy_dma.py
compares against it): do that ONLY for a deliberate change of a kernel's staging."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yolov5_amd", "csrc")
GOLDEN = os.path.join(ROOT, "tests", "golden", "isa_lds_dma_counts.json")
UNITS = ["convg8.hip", "convh3.hip", "bneck.hip", "head.hip", "sppf.hip"]


def counts_of(src, hipcc="/opt/rocm/bin/hipcc"):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, src + ".s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-o", out,
                            os.path.join(CSRC, src)], capture_output=True, text=True, cwd=CSRC)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-3000:])
        counts, cur = {}, None
        for line in open(out):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = m.group(1)
                counts[cur] = 0
            elif cur and "s_endpgm" in line:
                cur = None
            elif cur and re.search(r"buffer_load_dwordx4 .* lds", line):
                counts[cur] += 1
        return {k: v for k, v in counts.items() if v}


if __name__ == "__main__":
    res = {u: counts_of(u) for u in UNITS}
    if "--write" in sys.argv:
        with open(GOLDEN, "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
        print("wrote", GOLDEN)
    else:
        print(json.dumps(res, indent=1, sort_keys=True))
