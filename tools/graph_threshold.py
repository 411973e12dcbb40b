"""Eager launches vs HIP-graph replay of the network evaluation as a function of chunk size (pairs = replicas x N^2).
    python tools/graph_threshold.py [--replicas 1000] [--steps 30]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--replicas", type=int, default=1000)
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
from str2str_amd.common.rigid_utils import Rigid  # noqa: E402
from str2str_amd.factory import build_diffuser, build_synthetic_net  # noqa: E402
from str2str_amd.sampler import forward_backward  # noqa: E402
from str2str_amd.synth import synth_chain  # noqa: E402

dev = torch.device("cuda")
net = build_synthetic_net(seed=0, sigma_final=0.002, device=dev)
diff = build_diffuser("/tmp/str2str_cache_gt")
for N in (10, 20, 28, 35, 47, 56, 73, 80, 128):
    feats = synth_chain(N)
    B = a.replicas
    rig0 = Rigid.from_tensor_4x4(feats["rigidgroups_gt_frames"][..., 0, :, :].repeat(B, 1, 1, 1))
    res = {}
    for mode in ("0", "1"):
        os.environ["S2S_HIP_GRAPH"] = mode
        for rep in range(2):   # first pass: warm-up (+ capture)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            forward_backward(net, diff, feats, rig0, 1.0, num_timesteps=a.steps, min_t=0.01, probability_flow=True,
                             self_conditioning=True, device=dev, rng="device")
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / (a.steps + 1) * 1e3
    print(f"N={N:4d} B={B}: {B * N * N / 1e6:7.2f} M pairs  eager {res['0']:7.2f} ms/eval  graph {res['1']:7.2f} ms/eval  ratio {res['0'] / res['1']:.2f}", flush=True)
