import time, torch, sys, logging
sys.path.insert(0, ".")
from str2str_amd.factory import build_synthetic_net, build_diffuser
from str2str_amd.sampler import forward_backward
from str2str_amd.common.rigid_utils import Rigid
from str2str_amd.synth import synth_chain
logging.basicConfig(level=logging.WARNING)
N, B, S = 256, 32, 10
feats = synth_chain(N)
rig0 = Rigid.from_tensor_4x4(feats["rigidgroups_gt_frames"][..., 0, :, :].repeat(B, 1, 1, 1))
diff = build_diffuser("/tmp/cache_x")
def build(scale):
    net = build_synthetic_net(seed=0, sigma_final=0.002, device="cuda")
    with torch.no_grad():
        et = net.translator.trunk["edge_transition_1"]
        et.trunk[0].weight.mul_(scale); et.trunk[0].bias.mul_(scale); et.trunk[2].weight.div_(scale)
    return net
def run(net):
    torch.manual_seed(9); torch.cuda.synchronize(); t = time.time()
    forward_backward(net, diff, feats, rig0, 1.0, num_timesteps=S, device="cuda", rng="device"); torch.cuda.synchronize()
    return time.time() - t
plain, hot = build(1.0), build(8e3)
run(plain); t0 = min(run(plain), run(plain))
t_first = run(hot); t1 = min(run(hot), run(hot))
fb, ps = getattr(hot, "range_fallback", None), getattr(hot, "range_prescale", None)
print(f"B={B} N={N} {S} steps: plain {t0:.3f} s; hidden layer x 8e3: first chunk (flag + re-run) {t_first:.3f} s, later chunks {t1:.3f} s = {t1/t0:.3f} x plain; fp32 fallback {fb}, block exponent {ps}")
