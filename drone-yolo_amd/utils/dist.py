"""Rank launcher of the multi-GPU paths (reference: ultralytics/utils/dist.py:13-66 ``find_free_network_port``,
``generate_ddp_file``, ``generate_ddp_command``, ``ddp_cleanup``; call site engine/trainer.py:185-205).

The reference's parent process writes a throw-away script that rebuilds the trainer from its arguments and runs it under
``python -m torch.distributed.run --nproc_per_node N``, one rank per GPU.  The same is done here, with two rules of
the MI355X pool on top: the parent must not have touched the GPU before it starts the ranks (it never ``exec``s; the ranks
are CHILD processes and the parent exits with their return code), and the rendezvous address is 127.0.0.1.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import tempfile
from typing import List, Optional, Sequence, Tuple

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def find_free_network_port() -> int:
    """A free TCP port on localhost for MASTER_PORT — reference dist.py:13-23."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(extra: Optional[dict] = None) -> dict:
    """Environment of the rank processes: dmabuf IPC (the host driver has no legacy IPC), loopback rendezvous."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "4")
    for k, v in (extra or {}).items():
        if v is None:
            env.pop(k, None)
        else:
            env[k] = v
    return env


def visible_device_env(devs: Sequence[int]) -> dict:
    """Environment that makes local rank i run on the i-th REQUESTED GPU (reference: select_device sets CUDA_VISIBLE_DEVICES to the
    ``device`` string, utils/torch_utils.py:183, so ``device="2,3"`` trains on GPUs 2 and 3, not 0 and 1).  Ids are positions in what
    this process can see (an outer HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES list is honoured and mapped through); they are checked
    against the device count without initialising HIP.  Only HIP_VISIBLE_DEVICES is exported (CUDA_VISIBLE_DEVICES is cleared) so the
    two lists cannot compound."""
    devs = [int(d) for d in devs]
    have = visible_gpu_count()
    bad = [d for d in devs if d < 0 or d >= have]
    if bad or len(set(devs)) != len(devs):
        raise RuntimeError(f"device ids {devs}: {'duplicates' if not bad else f'{bad} not among the {have} visible GPU(s)'}")
    outer = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    if outer:
        outer_ids = [x.strip() for x in outer.split(",") if x.strip() != ""]
        ids = [outer_ids[d] for d in devs]
    else:
        ids = [str(d) for d in devs]
    return {"HIP_VISIBLE_DEVICES": ",".join(ids), "CUDA_VISIBLE_DEVICES": None}


def torchrun_command(world_size: int, script: str, argv: Sequence[str] = (), port: Optional[int] = None) -> List[str]:
    """``python -m torch.distributed.run`` argv for ``world_size`` ranks on this node — reference dist.py:56-66."""
    port = port or find_free_network_port()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world_size}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script, *argv]


def generate_ddp_file(trainer_overrides: dict, trainer_cls: str = "drone_yolo_amd.engine.trainer.DetectionTrainer") -> str:
    """Temp script that rebuilds the trainer from its overrides and trains — reference dist.py:25-53."""
    module, name = trainer_cls.rsplit(".", 1)
    content = f"""# Drone-YOLO multi-GPU training temp file (deleted after use)
import sys
sys.path.insert(0, {ROOT!r})
overrides = {trainer_overrides!r}

if __name__ == "__main__":
    from {module} import {name}

    trainer = {name}(overrides=overrides)
    trainer.train()
"""
    d = os.path.join(tempfile.gettempdir(), "dyolo_ddp")
    os.makedirs(d, exist_ok=True)
    with tempfile.NamedTemporaryFile(prefix="_temp_", suffix=f"{os.getpid()}.py", mode="w+", encoding="utf-8", dir=d, delete=False) as f:
        f.write(content)
    return f.name


def generate_ddp_command(world_size: int, trainer_overrides: dict) -> Tuple[List[str], str]:
    """(argv, temp file) — reference dist.py:56-66."""
    file = generate_ddp_file(trainer_overrides)
    return torchrun_command(world_size, file), file


def ddp_cleanup(file: str) -> None:
    """Delete the temp file — reference dist.py:69-72."""
    if os.path.basename(file).startswith("_temp_") and os.path.exists(file):
        os.remove(file)


def visible_gpu_count() -> int:
    """GPUs this process could use, WITHOUT initialising HIP (device_count() reads the topology only on this image)."""
    import torch

    return torch.cuda.device_count()


def launch_ranks(world_size: int, script: str, argv: Sequence[str] = (), env: Optional[dict] = None, allow_cpu_ranks: bool = False) -> int:
    """Start ``world_size`` ranks of ``script`` as child processes and wait for them; returns their exit code.
    Fails loudly (before starting anything) when the node has fewer GPUs than ranks — unless the caller asks for CPU /
    shared-device rehearsal ranks (DYOLO_DIST_BACKEND=gloo)."""
    if not allow_cpu_ranks:
        have = visible_gpu_count()
        if have < world_size:
            raise RuntimeError(f"{world_size} ranks requested but this node exposes {have} GPU(s); one process per GPU, no oversubscription")
    cmd = torchrun_command(world_size, script, argv)
    return subprocess.call(cmd, env=rank_env(env))
