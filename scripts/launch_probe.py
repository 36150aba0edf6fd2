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
