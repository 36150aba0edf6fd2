"""Host launch-rate probe: how many tiny kernels per second this box's CPU + driver can enqueue (a slow host shows up as
inter-kernel gaps in the forward: 52+ launches per 3 ms)."""
import time, torch
dev = torch.device("cuda:0")
a = torch.zeros(64, device=dev)
for _ in range(1000): a.add_(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20000
for _ in range(n): a.add_(1)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6 * (t1 - t0) / n:.2f} us/launch (host), drain total {1e6 * (t2 - t0) / n:.2f} us/launch (device-side rate)")

# HBM copy bandwidth of this box (1 GiB fp16 tensor -> another, 20 times): read + write bytes per second
n = 512 * 1024 * 1024
src = torch.empty(n, dtype=torch.float16, device=dev).normal_()
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    dst.copy_(src)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"copy 1 GiB: {ms:.3f} ms = {2 * n * 2 / ms / 1e9:.2f} TB/s (read + write)")
# MFMA-heavy: fp16 GEMM 8192^3 through torch (hipBLASLt) as a clock / power reference
a = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
b = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
for _ in range(3):
    a @ b
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    a @ b
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"GEMM 8192^3 fp16: {ms:.3f} ms = {2 * 8192 ** 3 / ms / 1e9:.0f} TFLOP/s")
