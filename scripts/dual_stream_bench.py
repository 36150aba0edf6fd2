#!/usr/bin/env python
"""Experiment (GPU): yolov5s 64 x 3x640x640 fp16 forward as ONE batch-64 plan vs TWO batch-32 plans on two HIP streams
(the tails / prologues / epilogues of one stream's persistent kernels overlap with the other stream's main loops)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    x = torch.rand((64, 3, 640, 640), device=dev).half()
    m = bench.build_model("yolov5s", dev)
    with torch.no_grad():
        t1 = timeit(lambda: m(x))
        print(f"one plan, batch 64: {t1:.3f} ms")
        for parts in (2, 4):
            ms = [bench.build_model("yolov5s", dev) for _ in range(parts)]
            xs = [c.contiguous() for c in x.chunk(parts)]
            streams = [torch.cuda.Stream(dev) for _ in range(parts)]

            def run():
                cur = torch.cuda.current_stream(dev)
                for mm, xx, s in zip(ms, xs, streams):
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        mm(xx)
                for s in streams:
                    cur.wait_stream(s)

            t2 = timeit(run)
            print(f"{parts} plans of batch {64 // parts} on {parts} streams: {t2:.3f} ms")

            def run_seq():
                for mm, xx in zip(ms, xs):
                    mm(xx)

            t3 = timeit(run_seq)
            print(f"{parts} plans of batch {64 // parts} on one stream: {t3:.3f} ms")



def split_engine_check():
    """The product's SplitEngine (one model, two sub-batch plans) timed the same way."""
    from yolov5_amd.engine import SplitEngine

    dev = torch.device("cuda:0")
    x = torch.rand((64, 3, 640, 640), device=dev).half()
    m = bench.build_model("yolov5s", dev)
    with torch.no_grad():
        se = SplitEngine(m, tuple(x.shape), torch.float16, dev, want_raw=True, parts=2)
        print(f"SplitEngine(parts=2): {timeit(lambda: se(x)):.3f} ms")
        # same two engines driven by hand
        def run():
            cur = torch.cuda.current_stream(dev)
            for i, (eng, s) in enumerate(zip(se.engines, se.streams)):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    eng(x[i * 32:(i + 1) * 32])
            for s in se.streams:
                cur.wait_stream(s)
        print(f"same engines, manual loop: {timeit(run):.3f} ms")
        xs = [c.contiguous() for c in x.chunk(2)]
        def run2():
            cur = torch.cuda.current_stream(dev)
            for eng, xx, s in zip(se.engines, xs, se.streams):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    eng(xx)
            for s in se.streams:
                cur.wait_stream(s)
        print(f"same engines, pre-chunked inputs: {timeit(run2):.3f} ms")


if len(sys.argv) > 1 and sys.argv[1] == "split":
    split_engine_check()
    sys.exit(0)

if __name__ == "__main__":
    main()
