"""What the device's memory sustains for reads, writes and a mix (torch's own fill / copy / sum kernels, 1 GiB buffers):
the histogram and statistics passes of the default-parameter path write as much as they read."""
import torch
n = 1 << 28                       # 1 GiB of int32
x = torch.empty(n, dtype=torch.int32, device="cuda"); y = torch.empty_like(x)
x.fill_(1); y.fill_(2); torch.cuda.synchronize()
def t(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
gb = n * 4 / 1e9
tw = t(lambda: x.fill_(3)); print(f"write only  {gb / tw:8.0f} GB/s")
tr = t(lambda: x.sum());    print(f"read only   {gb / tr:8.0f} GB/s")
tc = t(lambda: y.copy_(x)); print(f"copy        {2 * gb / tc:8.0f} GB/s (read + write bytes)")
ta = t(lambda: torch.add(x, 1, out=y)); print(f"y = x + 1   {2 * gb / ta:8.0f} GB/s (read + write bytes)")
