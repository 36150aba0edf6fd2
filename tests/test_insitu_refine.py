"""CPU (host emulator): the decision logic of Engine._refine_in_situ (yolov5_amd/engine.py, round 6) with the in-situ profile replaced by scripted times --
the emulator's clock means nothing.  A plain convolution of yolov5n gets a hand-made (winner, runner-up) pair; checked: the runner-up is installed only when it is
more than 3 % faster in place AND the range did not get slower as a whole, the installed configuration is the one the plan then launches (same outputs to fp16
accumulation noise: another tile shape adds in another order), decisions persist in the tile-choice cache and are re-applied without a profile pass, and
y5_plan_set_conv_cfg refuses anything but a plain convolution op.  The GPU counterpart is tests/test_gpu_plans.py::test_in_situ_refinement_of_the_tuned_plan."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import detgen
from tests.hipemu.backend import EmuBackend
from tests.test_emu_model import det_model
import yolov5_amd.engine as eng_mod
from yolov5_amd.engine import Engine


class _Lib:
    """the real library with y5_plan_profile_range scripted"""

    def __init__(self, lib, script):
        self._lib, self._script, self.calls = lib, script, 0

    def __getattr__(self, name):
        return getattr(self._lib, name)

    def y5_plan_profile_range(self, plan, lo, hi, iters, st, buf):
        times = self._script[self.calls]
        self.calls += 1
        for k in range(hi - lo):
            buf[k] = times.get(lo + k, 1.0)
        return 0


def _engine(monkeypatch, tmp_path):
    monkeypatch.setenv("Y5_TUNE_CACHE", str(tmp_path / "tune.json"))
    monkeypatch.delenv("Y5_DISABLE", raising=False)
    monkeypatch.delenv("Y5_TUNE_RANK", raising=False)
    m = det_model("yolov5n", 0, True).half()
    return Engine(m, (1, 3, 64, 64), torch.float16, "cpu", want_raw=False, backend=EmuBackend())


@pytest.fixture()
def clean_cache():
    eng_mod._TUNE_CACHE.clear()
    eng_mod._TUNE_FILE_STATE["loaded"] = False
    yield
    eng_mod._TUNE_CACHE.clear()
    eng_mod._TUNE_FILE_STATE["loaded"] = False


def test_refinement_decisions(monkeypatch, tmp_path, clean_cache):
    x = torch.from_numpy(detgen.uniform((1, 3, 64, 64), 0.0, 1.0, name="img", seed=0)).half()
    eng = _engine(monkeypatch, tmp_path)
    z_ref = np.asarray(eng(x)["z"]).astype(np.float32).copy()
    names = eng.op_names
    ops = [i for i, n in enumerate(names) if n.startswith("conv:") and ("7.Conv" in n or "5.Conv" in n)]
    assert len(ops) == 2, names
    slot = {i: [j for j, n in enumerate(names) if n.startswith(("conv:", "conv+pw:", "conv+decode:"))].index(i) for i in ops}
    a, b = ops
    n = eng.lib.y5_plan_size(eng.plan)
    hi = n if eng._stem is None else eng._stem

    def cands(e):
        return [dict(idx=i, slot=slot[i], key=("t", i), best=e.conv_cfgs[slot[i]], second=0) for i in (a, b)]

    # (1) op a: runner-up 10 % faster in place -> installed; op b: 2 % faster -> not; the closing profile is not slower -> the mix is kept
    lib = _Lib(eng.lib, [{a: 1.0, b: 1.0}, {a: 0.9, b: 0.98}, {a: 0.9, b: 1.0}])
    eng.lib = lib
    best_a, best_b = eng.conv_cfgs[slot[a]], eng.conv_cfgs[slot[b]]
    assert best_a != 0
    eng._insitu = cands(eng)
    eng._refine_in_situ(1, hi)
    assert lib.calls == 3 and eng.insitu_timed and eng.insitu_swaps == [(names[a], best_a, 0)]
    assert eng.conv_cfgs[slot[a]] == 0 and eng.conv_cfgs[slot[b]] == best_b
    assert eng_mod._TUNE_CACHE[("t", a, eng_mod._INSITU_MARK, a)] == (0, -1) and eng_mod._TUNE_CACHE[("t", b, eng_mod._INSITU_MARK, b)] == (best_b, -1)
    z_swapped = np.asarray(eng(x)["z"]).astype(np.float32)
    np.testing.assert_allclose(z_swapped, z_ref, rtol=1e-2, atol=1e-2 * max(1.0, np.abs(z_ref).max() / 64))

    # (2) a second engine of the same plan: the stored decisions, no profile pass
    eng2 = _engine(monkeypatch, tmp_path)
    lib2 = _Lib(eng2.lib, [])
    eng2.lib = lib2
    eng2._insitu = cands(eng2)
    eng2._refine_in_situ(1, hi)
    assert lib2.calls == 0 and eng2.conv_cfgs[slot[a]] == 0 and eng2.insitu_swaps == [(names[a], best_a, 0)] and not eng2.insitu_timed
    np.testing.assert_array_equal(np.asarray(eng2(x)["z"]).astype(np.float32), z_swapped)

    # (2b) faster in the runner-up profile, not confirmed by the closing profile: goes back
    eng_mod._TUNE_CACHE.clear()
    eng2b = _engine(monkeypatch, tmp_path)
    lib2b = _Lib(eng2b.lib, [{a: 1.0, b: 1.0}, {a: 0.9, b: 0.9}, {a: 0.9, b: 0.99}])
    eng2b.lib = lib2b
    eng2b._insitu = cands(eng2b)
    eng2b._refine_in_situ(1, hi)
    assert lib2b.calls == 3 and eng2b.insitu_swaps == [(names[a], best_a, 0)] and eng2b.conv_cfgs[slot[b]] == best_b

    # (3) a faster launch but a slower range (the closing profile): everything goes back
    eng_mod._TUNE_CACHE.clear()
    eng3 = _engine(monkeypatch, tmp_path)
    lib3 = _Lib(eng3.lib, [{a: 1.0, b: 1.0}, {a: 0.9, b: 1.0}, {a: 0.9, b: 1.0, a + 1: 1.5}])
    eng3.lib = lib3
    eng3._insitu = cands(eng3)
    eng3._refine_in_situ(1, hi)
    assert lib3.calls == 3 and eng3.insitu_swaps == [] and eng3.conv_cfgs[slot[a]] == best_a
    np.testing.assert_array_equal(np.asarray(eng3(x)["z"]).astype(np.float32), z_ref)

    # (4) switched off / runner-up plans: nothing is touched
    for env in ({"Y5_DISABLE": "insitu_tune"}, {"Y5_TUNE_RANK": "1"}):
        eng_mod._TUNE_CACHE.clear()
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng3.lib = _Lib(lib3._lib, [])
        eng3._insitu = cands(eng3)
        eng3._refine_in_situ(1, hi)
        assert eng3.lib.calls == 0 and eng3.conv_cfgs[slot[a]] == best_a and eng3._insitu == []
        for k in env:
            monkeypatch.delenv(k)


def test_set_conv_cfg_only_on_convolution_ops(monkeypatch, tmp_path, clean_cache):
    eng = _engine(monkeypatch, tmp_path)
    lib = eng.lib
    kinds = {n.split(":")[0] for n in eng.op_names}
    other = next(i for i, n in enumerate(eng.op_names) if not n.startswith("conv:"))
    assert lib.y5_plan_set_conv_cfg(eng.plan, other, 0) != 0, (eng.op_names[other], kinds)
    assert lib.y5_plan_set_conv_cfg(eng.plan, lib.y5_plan_size(eng.plan), 0) != 0
    assert lib.y5_plan_set_conv_cfg(None, 0, 0) != 0
