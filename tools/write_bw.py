import torch, time
x = torch.empty(16*4096*4096, dtype=torch.float64, device="cuda")
for name, fn in (("fill_", lambda: x.fill_(1.5)), ("zero_", lambda: x.zero_()), ("copy_ (r+w)", None)):
    if fn is None:
        y = torch.empty_like(x)
        fn = lambda: y.copy_(x)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/10
    print(name, ms, "ms", x.numel()*8/ms/1e9, "TB/s written")
