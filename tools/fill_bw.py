import torch, time
for n in (1<<27, 1<<30):
    x = torch.empty(n, dtype=torch.uint8, device="cuda")
    y = torch.empty(n//16, 4, dtype=torch.int32, device="cuda")
    for name, fn in (("fill_u8", lambda: x.fill_(2)), ("fill_i32x4", lambda: y.fill_(0x02020202)), ("zero_", lambda: x.zero_())):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(10):
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        t = min(ts)
        print(f"{name} n={n>>20} MiB: {t*1e3:.1f} us  {n/t/1e9:.2f} TB/s", flush=True)
