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


if __name__ == "__main__":
    main()
