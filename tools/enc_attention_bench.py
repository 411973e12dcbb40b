import os, sys, torch
sys.path.insert(0, os.getcwd())
from str2str_amd import ops
B, N = 128, 256
qkv = torch.randn(B * N, 960, device="cuda")
for ar in ("f32", "f16x3", "f32", "f16x3"):
    for _ in range(3): ops.encoder_attention(qkv, None, B, N, 4, arith=ar)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.encoder_attention(qkv, None, B, N, 4, arith=ar)
    e.record(); torch.cuda.synchronize()
    print(ar, round(s.elapsed_time(e) / 20 * 1e3, 1), "us")
