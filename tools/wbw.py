import torch, time
x = torch.empty(612*1024*1024//2, dtype=torch.float16, device="cuda")
y = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
print("fill 612MB: %.1f us -> %.2f TB/s" % (t(lambda: x.zero_()), 612*1.048576e6/t(lambda: x.zero_())/1e6))
print("copy 612MB: %.1f us -> %.2f TB/s (r+w)" % (t(lambda: y.copy_(x)), 2*612*1.048576e6/t(lambda: y.copy_(x))/1e6))
