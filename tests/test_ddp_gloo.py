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
