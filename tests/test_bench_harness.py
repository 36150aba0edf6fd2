"""CPU: bench.py's N > 1 rank logic, executed (VERDICT r2 item 8 -- no multi-GPU node was available to the driver in rounds 1-2, so the branch the
first real 8-GPU run takes had never run anywhere).  `bench.py --gpus 2 --dry-run-emu` re-executes itself under torch.distributed.run exactly as
`--gpus 2` does on a GPU box, every rank reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, installs the CPU emulator instead of a HIP device and
gloo instead of RCCL, and runs the shared code of main(): K timed steps between fences (barrier), max over ranks (all_reduce MAX), the DDP
training probe (`smart_DDP`: parameter broadcast, loss * WORLD_SIZE as train.py:404-405, bucketed all-reduce(mean) of the flat gradient arena
overlapped with the backward plan), rank 0 printing ONE JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_bench_two_rank_dry_run_on_the_emulator():
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-emu", "--model", "yolov5n", "--batch", "2", "--imgsz", "64",
           "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 only, one line
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in r, k
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert r["config"]["global_batch"] == 4 and r["value"] > 0
    assert abs(r["value"] - 4 * 2 / (r["ms_per_step"] * 2 * 1e-3)) < 0.02 * r["value"]     # whole-job images / max-over-ranks time
    t = r["train"]
    assert t["rccl_ranks"] == 2 and t["allreduce_buckets"] >= 1
    # the all-reduced bytes are the flat fp32 gradient arena: every parameter in a 64-float padded slot (yolov5n: 7.49 MB; yolov5s: 28.94 MB)
    sys.path.insert(0, ROOT)
    from yolov5_amd.yolo import DetectionModel

    want = sum(-(-p_.numel() // 64) * 64 * 4 for p_ in DetectionModel("yolov5n.yaml").parameters())
    assert t["allreduce_bytes_per_step"] == want, (t["allreduce_bytes_per_step"], want)
    assert t["loss"] > 0 and t["images_per_sec"] > 0
