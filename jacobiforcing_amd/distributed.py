"""Process-per-GPU replication: prompts shard over ranks with no data-path collective; the only exchange is the
final throughput gather.  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist


@dataclass
class RankInfo:
    rank: int = 0
    world_size: int = 1
    local_rank: int = 0


def pin_rank_cpus(info: RankInfo, local_world: Optional[int] = None) -> Optional[int]:
    """Give every rank of a node its own contiguous slice of the host cores (and an OMP/torch thread count to match), so
    that eight ranks' host threads — the mailbox poll, the tokenizer-free driver loop, torch's intra-op pool — do not all
    land on the same cores.  JF_PIN_CPUS=0 switches it off.  Returns the number of cores of the slice (None: untouched)."""
    if os.environ.get("JF_PIN_CPUS", "1") == "0" or not hasattr(os, "sched_getaffinity"):
        return None
    lw = int(local_world or os.environ.get("LOCAL_WORLD_SIZE", info.world_size) or 1)
    if lw <= 1:
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = len(cpus) // lw
        if per < 1:
            return None
        mine = cpus[(info.local_rank % lw) * per:(info.local_rank % lw + 1) * per]
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(per, int(os.environ.get("OMP_NUM_THREADS", per)))))
        return per
    except OSError:
        return None


def init_from_env(backend: Optional[str] = None, force: Optional[bool] = None) -> RankInfo:
    """Read RANK / LOCAL_RANK / WORLD_SIZE and create the process group when there is more than one rank — or when
    ``force`` (env JF_DIST_FORCE_INIT=1) asks for a single-rank group: a one-rank "nccl" group still loads RCCL, creates
    the communicator and launches its reduce kernels, which is how the collective path is exercised on a one-GPU box."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if force is None:
        force = os.environ.get("JF_DIST_FORCE_INIT", "0") == "1"
    info = RankInfo(rank, ws, local)
    if (ws > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL's P2P setup)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("JF_FORCE_DEVICE", local)))
        pin_rank_cpus(info)
        dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return info


def backend_name() -> Optional[str]:
    return dist.get_backend() if dist.is_initialized() else None


def barrier(device: Optional[torch.device] = None) -> None:
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def shard_prompts(prompts, info: RankInfo):
    """prompt i -> rank i mod world (independent units, zero exchange during decoding)."""
    return [p for i, p in enumerate(prompts) if i % info.world_size == info.rank]


def gather_throughput(tokens: float, iterations: float, seconds: float, device=None) -> dict:
    """Whole-job numbers: sum of tokens and iterations over ranks, max of the per-rank wall time."""
    if not dist.is_initialized():
        return dict(tokens=float(tokens), iterations=float(iterations), seconds=float(seconds), world_size=1)
    dev = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    s = torch.tensor([float(tokens), float(iterations)], dtype=torch.float64, device=dev)
    m = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return dict(tokens=float(s[0]), iterations=float(s[1]), seconds=float(m[0]), world_size=dist.get_world_size())
