"""CPU, 2 processes over gloo: HipDDP (yolov5_amd/torch_utils.py) driven by the training engine's backward plan on the HIP
emulator.  Each rank trains on its own images; the averaged bucket gradients must equal the mean of the two ranks'
single-process gradients, parameters must be broadcast from rank 0, and every rank must end with identical gradients."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import detgen, yolo_oracle as yo
    from tests.hipemu.backend import EmuBackend
    from yolov5_amd.torch_utils import HipDDP
    from yolov5_amd.train_engine import TrainEngine
    from yolov5_amd.yolo import DetectionModel

    torch.manual_seed(100 + rank)  # different initial weights per rank: the broadcast must fix that
    m = DetectionModel("yolov5n.yaml").train()
    if rank == 0:
        m.load_state_dict(yo.det_state_dict(yo.model_cfg("yolov5n"), 0, fused=False))
    ddp = HipDDP(m, bucket_cap_mb=1.0)  # 1.87 M params -> several buckets
    ref_sd = yo.det_state_dict(yo.model_cfg("yolov5n"), 0, fused=False)
    for k, v in m.state_dict().items():
        if v.dtype.is_floating_point:
            assert torch.equal(v, ref_sd[k]), f"rank {rank}: {k} not broadcast from rank 0"
    B = 1
    x = torch.from_numpy(detgen.uniform((B, 3, 64, 64), 0.0, 1.0, name="img", seed=10 + rank)).half()
    ups = None
    res = {}
    for mode in ("ddp", "local"):
        eng = TrainEngine(m, (B, 3, 64, 64), "cpu", backend=EmuBackend())
        eng.grad_sink = ddp if mode == "ddp" else None
        outs = eng.forward(x)
        if ups is None:
            ups = [torch.from_numpy(detgen.uniform(tuple(o.shape), -1, 1, name=f"u{i}", seed=5)).half() for i, o in enumerate(outs)]
        grads = eng.backward(ups)
        res[mode] = [g.float().clone() for g in grads]
    assert len(ddp.buckets) > 2
    torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_hipddp_two_ranks_gloo(tmp_path):
    import socket

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in (0, 1))
    for a, b, l0, l1 in zip(r0["ddp"], r1["ddp"], r0["local"], r1["local"]):
        assert torch.equal(a, b)                                   # every rank holds the same reduced gradient
        torch.testing.assert_close(a, (l0 + l1) / 2, rtol=1e-5, atol=1e-6)  # = mean over ranks of the local gradients
    assert not torch.equal(r0["local"][0], r1["local"][0])         # the ranks really saw different data


def _syncbn_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import detgen, yolo_oracle as yo
    from tests.hipemu.backend import EmuBackend
    from yolov5_amd.train_engine import TrainEngine
    from yolov5_amd.yolo import DetectionModel

    m = DetectionModel("yolov5n.yaml").train()
    m.load_state_dict(yo.det_state_dict(yo.model_cfg("yolov5n"), 0, fused=False))
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m)            # train.py:269-271
    assert any(isinstance(x, torch.nn.SyncBatchNorm) for x in m.modules())
    x = torch.from_numpy(detgen.uniform((2, 3, 64, 64), 0.0, 1.0, name="img", seed=21))[rank:rank + 1].contiguous()
    eng = TrainEngine(m, (1, 3, 64, 64), "cpu", backend=EmuBackend(), dtype=torch.float32)
    outs = eng.forward(x)
    ups = [torch.from_numpy(detgen.uniform((2, *o.shape[1:]), -1, 1, name=f"su{i}", seed=6))[rank:rank + 1].contiguous() for i, o in enumerate(outs)]
    grads = eng.backward(ups)
    bn0 = next(x for x in m.modules() if isinstance(x, torch.nn.SyncBatchNorm))
    torch.save({"outs": [torch.as_tensor(np.array(o)).float().clone() for o in outs], "grads": [g.float().clone() for g in grads],
                "rm": bn0.running_mean.clone(), "rv": bn0.running_var.clone()}, os.path.join(out_dir, f"sync{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_sync_batchnorm_two_ranks_equal_one_process_full_batch(tmp_path):
    """train.py:269-271 `SyncBatchNorm.convert_sync_batchnorm(model)`: two ranks with one image each, statistics exchanged through
    y5_bn_stats -> all-reduce -> y5_bn_silu_fwd_from_sums (and the backward pair), must reproduce ONE process running plain BatchNorm on the
    two-image batch: per-image outputs, running statistics, and parameter gradients (sum of the ranks' local gradients) -- fp32 plan."""
    import socket

    sys.path.insert(0, ROOT)
    from oracle import detgen, yolo_oracle as yo
    from tests.hipemu.backend import EmuBackend
    from yolov5_amd.train_engine import TrainEngine
    from yolov5_amd.yolo import DetectionModel

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_syncbn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(tmp_path, f"sync{k}.pt")) for k in (0, 1)]
    m = DetectionModel("yolov5n.yaml").train()
    m.load_state_dict(yo.det_state_dict(yo.model_cfg("yolov5n"), 0, fused=False))
    x = torch.from_numpy(detgen.uniform((2, 3, 64, 64), 0.0, 1.0, name="img", seed=21))
    eng = TrainEngine(m, (2, 3, 64, 64), "cpu", backend=EmuBackend(), dtype=torch.float32)
    outs = eng.forward(x)
    ups = [torch.from_numpy(detgen.uniform(tuple(o.shape), -1, 1, name=f"su{i}", seed=6)) for i, o in enumerate(outs)]
    grads = eng.backward(ups)
    for i, o in enumerate(outs):
        for k in (0, 1):
            torch.testing.assert_close(r[k]["outs"][i][0], torch.as_tensor(np.array(o))[k].float(), rtol=1e-3, atol=1e-4)   # fp32 sums in a different order
    bn0 = next(b for b in m.modules() if isinstance(b, torch.nn.BatchNorm2d))
    for k in (0, 1):
        torch.testing.assert_close(r[k]["rm"], bn0.running_mean, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(r[k]["rv"], bn0.running_var, rtol=1e-5, atol=1e-8)
    worst = 0.0
    for g0, g1, g in zip(r[0]["grads"], r[1]["grads"], grads):
        ref = g.float()
        err = float((g0 + g1 - ref).norm() / (ref.norm() + 1e-12))
        worst = max(worst, err)
    assert worst < 2e-4, worst
    # and the exchange really happened: a rank's own statistics (one image) differ from the global ones
    assert not torch.allclose(r[0]["outs"][0], r[1]["outs"][0])
