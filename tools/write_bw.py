import torch
for mb in (268, 537):
    x = torch.empty(mb * 1024 * 1024 // 4, device="cuda")
    y = torch.empty_like(x)
    for name, fn in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: y.copy_(x))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 10 * 1e3
        print(f"{name} {mb} MB: {us:.1f} us = {mb * 1.048576 / us * 1e-3 * 1e3:.2f} TB/s written")
