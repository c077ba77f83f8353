"""HBM write/copy bandwidth reference on this box (a 612 MB fill_ and copy_, the size of ten correlation pyramids):
the practical ceiling the write-bound kernels are compared with in DESIGN.md.  usage: python tools/hbm_fill_bw.py"""
import torch

x = torch.empty(612 * 1024 * 1024 // 2, dtype=torch.float16, device="cuda")
y = torch.empty_like(x)


def t(name, fn, nbytes, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    us = 1e3 * s.elapsed_time(e) / n
    print(f"{name}: {us:.1f} us  {nbytes / us / 1e6:.2f} TB/s")


t("fill_ 612 MB", lambda: x.fill_(1.0), x.numel() * 2)
t("copy_ 612 MB (read + write)", lambda: y.copy_(x), 2 * x.numel() * 2)
