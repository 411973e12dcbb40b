import sys, os, torch
sys.path.insert(0, os.getcwd())
from str2str_amd import ops
M, K, N = 32768, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / 16; b = torch.randn(N, device="cuda")
tg = ops.node_tiles(N, whole_row=(N <= 320))
wpk, xp = ops.pack_node_weight(w, tg), ops.pack_planes(x)
out, oxp = torch.empty(M, N, device="cuda"), ops.xp_alloc(M, N, "cuda")
for _ in range(4):
    ops.node_linear(xp, wpk, b, M, K, N, tg, out_f32=out, out_xp=oxp)
torch.cuda.synchronize()
