"""CPU (no GPU): the compiled gfx950 ISA keeps every "dummy" LDS-DMA (y5_common.h y5_bglds16_dummy) as its own instruction.

Counted `s_waitcnt vmcnt(N)` waits are only right when every wave issues exactly the number of loads the count assumes; identical dummies in a row were
once merged into one by the compiler (profiles/r05/r05_dummy_dma_merge.log), which no host-side emulation of the kernels can see.  Two checks: no kernel
source issues a dummy without the helper, and conv_headk.h's ring (three prologue stages + the loop body, each a real path of 4 loads and a dummy path
of 4) compiles to 32 LDS-DMA instructions per instantiation."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yolov5_amd", "csrc")


def test_every_dummy_goes_through_the_helper():
    pat = re.compile(r"y5_bglds16\([^;]*Y5_OOB\s*,\s*dummy\s*\)")
    bad = []
    for f in glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")):
        for n, line in enumerate(open(f), 1):
            if pat.search(line):
                bad.append(f"{os.path.basename(f)}:{n}")
    assert not bad, bad


def _lds_dma_counts(tmp_path, src, pattern):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / (src + ".s")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                        "-o", str(out), os.path.join(CSRC, src)], capture_output=True, text=True, cwd=CSRC)
    assert r.returncode == 0, r.stderr[-3000:]
    counts, cur = {}, None
    for line in open(out):
        m = re.match(pattern, line)
        if m:
            cur = m.group(1)
            counts[cur] = 0
        elif cur and "s_endpgm" in line:
            cur = None
        elif cur and re.search(r"buffer_load_dwordx4 .* lds", line):
            counts[cur] += 1
    return counts


def test_sppf_front_ring_keeps_its_dummy_loads(tmp_path):
    """conv_sppf.h (one instantiation): 33 LDS-DMA instructions with every dummy kept, 17 when the compiler merges them (-DY5_DUMMY_MERGEABLE, the pre-fix build)."""
    counts = _lds_dma_counts(tmp_path, "sppf.hip", r"^(_Z\w*y5_sppf_cv1_pool_kernel\w*):")
    assert list(counts.values()) == [33], counts


def test_headk_ring_keeps_its_dummy_loads(tmp_path):
    counts = _lds_dma_counts(tmp_path, "head.hip", r"^(_Z\w*y5_conv_headk_kernel\w*):")
    assert len(counts) == 2 and all(v == 32 for v in counts.values()), counts


@pytest.mark.parametrize("unit", ["convg8.hip", "convh3.hip", "bneck.hip", "head.hip", "sppf.hip"])
def test_lds_dma_counts_of_the_counted_wait_kernels_are_pinned(unit):
    """every kernel family that retires LDS-DMA with counted waits -- the 8-phase implicit GEMM (conv_g8.h: prologue 14 + loop 8 = 22 per
    instantiation without the virtual-upsample loader), the halo-resident 3x3 and the K-streamed pointwise kernel (conv_h3.h, conv_pwk.h), the fused Bottlenecks
    (conv_h3b.h with its 9-stage ring, W1 streamed into ring stages and the next tile's halo into dead planes; conv_bneck.h), the fused heads and the SPPF front --
    has the LDS-DMA instruction count of EACH instantiation pinned in tests/golden/isa_lds_dma_counts.json (scripts/isa_counts.py --write regenerates it, for a
    deliberate change of a kernel's staging only).  A merged dummy load, a dropped stage or a duplicated one changes a count.  (conv_igemm.h's hundred
    instantiations take minutes to compile and are covered by the source scan above and the emulator's worst-case landing model.)"""
    import json
    import sys

    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import isa_counts

    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    want = json.load(open(isa_counts.GOLDEN))[unit]
    got = isa_counts.counts_of(unit)
    assert got == want, {k: (got.get(k), want.get(k)) for k in set(got) | set(want) if got.get(k) != want.get(k)}
    if unit == "convg8.hip":
        assert got["_Z17y5_conv_g8_kernelILb0ELb0ELb0EEv12Y5ConvParams"] == 22 and got["_Z18y5_conv_g8n_kernelILb0ELb0ELb0EEv12Y5ConvParams"] == 20
        assert got["_Z17y5_conv_g8_kernelILb0ELb1ELb0EEv12Y5ConvParams"] == 22 and got["_Z18y5_conv_g8n_kernelILb0ELb1ELb0EEv12Y5ConvParams"] == 20   # (class-ordered taps)
        assert got["_Z17y5_conv_g8_kernelILb0ELb0ELb1EEv12Y5ConvParams"] == 22 and got["_Z18y5_conv_g8n_kernelILb0ELb0ELb1EEv12Y5ConvParams"] == 20   # (general-C1 loader)
