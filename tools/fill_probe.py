import torch
x = torch.empty((100000, 100000), dtype=torch.float32, device="cuda")
for name, fn in (("fill_", lambda: x.fill_(1.0)), ("zero_", lambda: x.zero_()), ("copy_", None)):
    if fn is None:
        y = torch.empty((50000, 100000), dtype=torch.float32, device="cuda")
        fn = lambda: x[:50000].copy_(y)
        nbytes = 2 * y.numel() * 4
    else:
        nbytes = x.numel() * 4
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("%-6s %7.3f ms  %.2f TB/s" % (name, ms, nbytes / ms / 1e9))
