"""CPU (no GPU): whole emulator test files re-run under the WORST-CASE LDS-DMA landing model (tests/hipemu, Y5_EMU_ASYNC=1: a load lands only when an
`s_waitcnt vmcnt(N)` of its wave covers it -- a counted wait that is one operation too lenient reads stale LDS).  The switch is latched per process, hence
child pytest processes.  Covered here: data-gradient and weight-gradient kernels, the 3x3 + pointwise fusion, the stream-K kernels, the fused Detect heads, the
SPPF front, BatchNorm and NMS, and -- since round 6 -- the kernels whose counted waits also count their global STORES and register loads: the fused front
(conv_front.h, with conv_stem.h / conv_k3.h in its two-launch plan) and the c_ = 32 / 64 Bottleneck (conv_bneck.h).  Those mark each such instruction with
Y5_EMU_VM_OP (csrc/y5_common.h), which takes a slot of the emulator's in-order queue as the instruction does on the hardware's counter; a wait widened by ONE
operation in any of the five headers fails these files (mutation run: profiles/r06/r06_store_landing_mutation.log).  The convolution families run case by case
in tests/test_emu_conv.py::test_conv_worst_case_dma_landing, the c_ = 128 Bottleneck in tests/test_emu_bneck.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["test_emu_front.py", "test_emu_bneck.py", "test_emu_dgrad.py", "test_emu_wgrad.py", "test_emu_k3pw.py", "test_emu_streamk.py", "test_emu_head.py", "test_emu_sppf.py", "test_emu_bn.py", "test_emu_nms.py"]


@pytest.mark.parametrize("name", FILES)
def test_file_under_worst_case_dma_landing(name):
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join("tests", name), "-q", "-x", "-o", "addopts=", "-p", "no:cacheprovider", "-k", "not worst_case and not async"],
                       env=dict(os.environ, Y5_EMU_ASYNC="1"), capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
