import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import golden
from str2str_amd.common.rigid_utils import Rigid
from str2str_amd import sampler
from str2str_amd.sampler import forward_backward
from str2str_amd.synth import synth_chain
from str2str_amd.factory import build_synthetic_net, build_diffuser
net = build_synthetic_net(device="cuda", scale=0.02) if "scale" in build_synthetic_net.__code__.co_varnames else build_synthetic_net(device="cuda")
diff = build_diffuser("/tmp/str2str_cache")
for tag in ["n16_s20", "n12_prior", "n24_delta", "cfg1_n64_s20", "n256_s5", "n256_s100"]:
    g = golden(f"traj_free_{tag}.npz")
    N, B = int(g["n_res"]), int(g["B"])
    feats = synth_chain(N)
    rig0 = Rigid.from_tensor_4x4(feats["rigidgroups_gt_frames"][..., 0, :, :].repeat(B, 1, 1, 1))
    torch.manual_seed(int(g["seed"]))
    marks = sorted(int(k[len("rigids_t_step"):]) for k in g if k.startswith("rigids_t_step"))
    trace = [] if marks else None
    print(tag, "start", flush=True)
    a37 = forward_backward(net, diff, feats, rig0, float(g["t_delta"]), num_timesteps=int(g["num_timesteps"]), device="cuda", trace=trace)
    torch.cuda.synchronize()
    print(tag, "ok", len(sampler._GRAPH_CACHE), flush=True)
print("clearing graphs", flush=True)
while sampler._GRAPH_CACHE:
    k = next(iter(sampler._GRAPH_CACHE))
    sampler._GRAPH_CACHE.pop(k)
    gc.collect(); torch.cuda.synchronize()
    print(" popped one", flush=True)
print("done", flush=True)
