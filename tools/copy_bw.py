"""Achievable HBM copy bandwidth on this GPU (device-to-device copy of a buffer far larger than the 256 MB Infinity Cache)."""
import torch
n = 4 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
a.fill_(1)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    b.copy_(a)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("copy %d MiB: %.3f ms -> %.2f TB/s (read + write)" % (n >> 20, ms, 2 * n / ms / 1e9))
