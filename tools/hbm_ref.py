import torch, time
for n in (4096, 8192, 16384):
    x = torch.rand((n*n,), dtype=torch.float64, device="cuda")
    for _ in range(3): s = x.sum()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10): s = x.sum()
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)/10
    print("torch.sum %5d^2 f64: %.1f us  %.2f TB/s" % (n, ms*1e3, n*n*8/ms/1e9))
    y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(10): y.copy_(x)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)/10
    print("copy      %5d^2 f64: %.1f us  %.2f TB/s (read+write)" % (n, ms*1e3, 2*n*n*8/ms/1e9))
    del x, y
