"""CPU (no GPU): whole emulator test files re-run under the WORST-CASE LDS-DMA landing model (tests/hipemu, Y5_EMU_ASYNC=1: a load lands only when an
`s_waitcnt vmcnt(N)` of its wave covers it -- a counted wait that is one load too lenient reads stale LDS).  The switch is latched per process, hence child
pytest processes.  Covered here: data-gradient and weight-gradient kernels, the 3x3 + pointwise fusion, the stream-K kernels, the fused Detect heads, the SPPF
front, BatchNorm and NMS (the convolution families run case by case in tests/test_emu_conv.py::test_conv_worst_case_dma_landing, the c_ = 128 Bottleneck in
tests/test_emu_bneck.py).  NOT covered, by construction: conv_front.h, conv_bneck.h (c_ = 32 / 64) and the four-wave conv_pw.h ids, whose counted waits also
count their global stores, which this model does not queue (on the hardware loads and stores retire in issue order on one counter)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["test_emu_dgrad.py", "test_emu_wgrad.py", "test_emu_k3pw.py", "test_emu_streamk.py", "test_emu_head.py", "test_emu_sppf.py", "test_emu_bn.py", "test_emu_nms.py"]


@pytest.mark.parametrize("name", FILES)
def test_file_under_worst_case_dma_landing(name):
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join("tests", name), "-q", "-x", "-o", "addopts=", "-p", "no:cacheprovider", "-k", "not worst_case and not async"],
                       env=dict(os.environ, Y5_EMU_ASYNC="1"), capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
