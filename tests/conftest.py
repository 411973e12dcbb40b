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


# ---- achieved parity margins: every parity test records (achieved, bound); written at session end to
#      gpurun_out/parity_margins[_cpu].json (copied into profiles/ for the record)
_MARGINS = {}


def record_margin(name, achieved, bound):
    cur = _MARGINS.get(name)
    if cur is None or achieved > cur["achieved"]:
        _MARGINS[name] = {"achieved": float(achieved), "bound": float(bound)}


def pytest_sessionfinish(session, exitstatus):
    if not _MARGINS:
        return
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        gpu = torch.cuda.is_available()
        with open(os.path.join(out, "parity_margins.json" if gpu else "parity_margins_cpu.json"), "w") as f:
            json.dump({"device": torch.cuda.get_device_name(0) if gpu else "cpu", "margins": _MARGINS}, f, indent=1, sort_keys=True)
    except OSError:
        pass
