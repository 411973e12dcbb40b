import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver via gpurun)")


def golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def T(a):
    return torch.as_tensor(np.asarray(a))


def manifest():
    with open(os.path.join(GOLDEN, "state_dict_manifest.json")) as f:
        return [(k, tuple(s)) for k, s in json.load(f)]


_SD_CACHE = {}


def synth_sd(seed=0, sigma_final=0.02):
    from str2str_amd.synth import synth_state_dict

    key = (seed, sigma_final)
    if key not in _SD_CACHE:
        _SD_CACHE[key] = synth_state_dict(manifest(), seed=seed, sigma_final=sigma_final)
    return _SD_CACHE[key]


def maxdiff(a, b):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max())


def backbone_rmsd(a, b):
    """Un-aligned RMSD over backbone atoms (both in the same frame), per sample -> max."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d2 = ((a - b) ** 2).sum(-1)
    return float(np.sqrt(d2.reshape(d2.shape[0], -1).mean(-1)).max())


@pytest.fixture(scope="session")
def sd_rough():
    return synth_sd(0, 0.02)


@pytest.fixture(scope="session")
def sd_smooth():
    return synth_sd(0, 0.002)


def igso3_f32_noise(rotvec, sigma, L=1000):
    """Per-residue bound on the float32 evaluation noise of the reference's IGSO(3) score.

    The reference sums 1000 float32 terms for f(omega) and f'(omega) (so3.py:21-62, :85-130) and
    divides: s = f' / (f + 1e-4).  Where f is tiny against the sum of |terms| (large omega at small
    sigma) its own result is rounding noise, so any other evaluation order differs by about
        rel(s) ~ eps32 * (sum|terms_f| / |f + 1e-4| + sum|terms_f'| / |f'|).
    Returns (rel_bound [B,N], |s| [B,N]) computed in float64 from the float32-rounded arguments.
    """
    rotvec = torch.as_tensor(np.asarray(rotvec)).float()
    sigma = torch.as_tensor(np.asarray(sigma)).float().reshape(-1, 1, 1)
    omega = torch.linalg.norm(rotvec, dim=-1) + 1e-6
    ls = torch.arange(L)
    arg = (omega[..., None] * (ls + 0.5)).double()
    w = ((2 * ls + 1) * torch.exp(-ls * (ls + 1) * sigma**2 / 2)).double()
    lo = torch.sin((omega / 2).double())[..., None]
    dlo = 0.5 * torch.cos((omega / 2).double())[..., None]
    hi, dhi = torch.sin(arg), (ls + 0.5).double() * torch.cos(arg)
    tf = w * hi / lo
    tdf = w * (lo * dhi - hi * dlo) / lo**2
    f, df = tf.sum(-1), tdf.sum(-1)
    eps = 2.0**-23
    tdf_mag = w * ((lo * dhi).abs() + (hi * dlo).abs()) / lo**2
    rel = 3 * eps * (tf.abs().sum(-1) / (f + 1e-4).abs() + tdf_mag.sum(-1) / df.abs().clamp(min=1e-30))
    # conditioning of the rotation-vector chain itself (matrix -> quaternion -> atan2 -> axis-angle,
    # rotation3d.py:102-161,525-553, float32): a small rotation is recovered from differences of
    # matrix entries (relative error ~ eps/omega); and because the reference does not standardise the
    # quaternion sign, half of the small relative rotations come out as omega = 2*pi - delta with
    # half-angle atan2(.,.) one ulp from pi: relative error of delta ~ (2*pi/delta) * ulp(pi)/(delta/2).
    om = omega.double()
    delta = (2 * np.pi - om).clamp(min=1e-9)
    chain = 8 * eps / om + torch.where(om > np.pi, (2 * np.pi / delta) * (4 * 2.4e-7 / delta), torch.zeros_like(om))
    return (rel + chain).numpy(), (df / (f + 1e-4)).abs().numpy()
