import torch, time
x = torch.empty(1 << 30, dtype=torch.float64, device="cuda")  # 8 GiB
for name, fn in [("fill_", lambda: x.fill_(1.0)), ("zero_", lambda: x.zero_()), ("copy 4GiB->4GiB", lambda: x[: 1 << 29].copy_(x[1 << 29 :]))]:
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    nbytes = x.numel() * 8 if "copy" not in name else x.numel() * 8
    print(f"{name}: {dt*1e3:.2f} ms  {nbytes/dt/1e12:.2f} TB/s (bytes moved {nbytes/2**30:.0f} GiB)")
