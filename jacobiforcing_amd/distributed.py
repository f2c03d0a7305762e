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


def init_from_env(backend: Optional[str] = None) -> RankInfo:
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("JF_FORCE_DEVICE", local)))
        dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return RankInfo(rank, ws, local)


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
