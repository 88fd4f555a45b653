"""Achievable HBM copy bandwidth on this box (reference point for the roofline fraction)."""
import torch, time
n = 2_400_000_000 // 4
x = torch.empty(n, dtype=torch.float32, device='cuda'); y = torch.empty_like(x)
x.fill_(1.0)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): y.copy_(x)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print("copy 2.4 GB -> 2.4 GB: %.3f ms, %.2f TB/s (read+write)" % (ms, 2 * n * 4 / ms / 1e9))
s.record()
for _ in range(10): y.fill_(2.0)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print("fill 2.4 GB: %.3f ms, %.2f TB/s (write)" % (ms, n * 4 / ms / 1e9))
s.record()
for _ in range(10): z = x.sum()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print("sum 2.4 GB: %.3f ms, %.2f TB/s (read)" % (ms, n * 4 / ms / 1e9))
