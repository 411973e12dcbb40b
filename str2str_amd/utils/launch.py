"""One command, any device count: the entry points start their own ranks.

The reference gets its multi-device run from Lightning (``trainer=ddp`` spawns one process per device behind
``trainer.predict``, reference src/eval.py:129,154 and configs/trainer/ddp.yaml:4-9).  Here ``bench.py --gpus N`` and
``eval.py trainer.devices=N`` do the same thing without Lightning: when they find themselves OUTSIDE a torch.distributed.run job
(no WORLD_SIZE in the environment) and more than one device is asked for, they re-execute themselves under
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` on 127.0.0.1 with a free rendezvous port, one process per GPU,
and pass the job's exit code on.  Inside such a job (the driver's own launch line) nothing is re-executed.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional


def in_distributed_job() -> bool:
    """True when this process was started by torch.distributed.run (or anything else that exports the rendezvous)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return int(so.getsockname()[1])


def launch_command(n: int, script: str, argv: List[str], port: Optional[int] = None) -> List[str]:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n)}", "--master-addr", "127.0.0.1",
            "--master-port", str(port if port is not None else free_port()), script] + list(argv)


def relaunch(n: int, script: str, argv: List[str]) -> int:
    """Run ``script argv`` as ``n`` ranks of one node and return the job's exit code (stdout / stderr are inherited: rank 0's JSON
    line or log goes where the caller's would have gone)."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this host's driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    return subprocess.run(launch_command(n, os.path.abspath(script), argv), env=env).returncode


def resolve_devices(devices) -> int:
    """``trainer.devices`` as Lightning reads it: an int, a digit string, a list of indices, or "auto" / -1 = every visible GPU."""
    if isinstance(devices, (list, tuple)):
        return len(devices)
    if isinstance(devices, str):
        if devices.strip().lstrip("-").isdigit():
            devices = int(devices)
        elif devices.strip() == "auto":
            devices = -1
        else:   # "0,1,2"
            return len([d for d in devices.split(",") if d.strip()])
    if devices is None:
        return 1
    if int(devices) < 0:
        import torch

        return max(1, torch.cuda.device_count())
    return max(1, int(devices))
