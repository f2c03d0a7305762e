"""Process-per-GPU replication: prompts shard over ranks with no data-path collective; the only exchange is the
final throughput gather.  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

# the host driver only supports dmabuf IPC (RCCL's P2P setup): must be in the environment before the HSA runtime starts, i.e.
# before the first torch.cuda call of the process — importing this module is early enough, init_from_env would not be
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist


@dataclass
class RankInfo:
    rank: int = 0
    world_size: int = 1
    local_rank: int = 0


def pin_rank_cpus(info: RankInfo, local_world: Optional[int] = None) -> Optional[int]:
    """OPT-IN (JF_PIN_CPUS=1): give every rank of a node its own contiguous slice of the host cores (and an OMP/torch thread
    count to match).  Off by default since round 5: measured on the MI355X box with eight ranks of the real model on one host
    (profiles/idle_gap_8ranks_r05.txt) the slices made the host gap between two iterations WORSE — median 21-45 us and p95
    100-370 us per rank against 8.5-10.8 us / 14-46 us unpinned (one rank alone: 7.9 / 41 us): the kernel's own placement of
    the ranks' main and runtime threads beats a topology-blind slice.  Returns the number of cores of the slice (None:
    untouched)."""
    if os.environ.get("JF_PIN_CPUS", "0") != "1" or not hasattr(os, "sched_getaffinity"):
        return None
    if local_world is None:
        local_world = os.environ.get("LOCAL_WORLD_SIZE")          # set by torch.distributed.run
    if local_world is None:
        # no launcher variable: one rank per GPU means the node holds as many ranks as it has GPUs (never the GLOBAL world
        # size: a multi-node job would otherwise give every rank 1 / world of its node's cores)
        ngpu = torch.cuda.device_count() if torch.cuda.is_available() else 0
        local_world = min(info.world_size, ngpu) if ngpu > 0 else info.world_size
    lw = int(local_world or 1)
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
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("JF_FORCE_DEVICE", local)))
        pin_rank_cpus(info)
        dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return info


def device_identity(index: Optional[int] = None) -> dict:
    """PCI bus id / UUID / architecture / CU count of the GPU this process launches on, asked of the HIP library that runs the
    kernels (jf_device_identity) — what a rank puts into its record so that the N-GPU line can be checked for N devices."""
    import ctypes
    from . import _native
    buf = ctypes.create_string_buffer(192)
    _native.check(_native.lib().jf_device_identity(-1 if index is None else int(index), buf, len(buf)), "jf_device_identity")
    out = {"raw": buf.value.decode()}
    for kv in out["raw"].split():
        k, _, v = kv.partition("=")
        out[k] = int(v) if k == "cus" else v
    return out


def ranks_seen() -> int:
    """World size as the COMMUNICATOR reports it (not the environment): 1 without a process group."""
    return dist.get_world_size() if dist.is_initialized() else 1


def gather_rank_records(record: dict) -> list:
    """Every rank's record on every rank, in rank order (one all_gather of pickled dicts over the job's backend — RCCL on the
    GPU box).  Without a process group: the one record."""
    if not dist.is_initialized():
        return [dict(record)]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, dict(record))
    return out


class DuplicateDeviceError(RuntimeError):
    pass


def check_distinct_devices(records: list, backend: Optional[str], allow_shared: bool = False) -> int:
    """Number of distinct GPUs behind the ranks' records (their ``device`` PCI bus id, UUID as tie-breaker).  Over RCCL
    ("nccl") two ranks on one device is never a valid N-GPU measurement: raise instead of reporting.  ``allow_shared`` is the
    explicit plumbing mode (JF_FORCE_DEVICE + gloo on a one-GPU box): counted, reported, not refused."""
    ids = [(r.get("device", {}).get("pci"), r.get("device", {}).get("uuid")) for r in records]
    distinct = len(set(ids))
    if distinct != len(records) and backend == "nccl" and not allow_shared:
        dup = sorted({i for i in ids if ids.count(i) > 1})
        raise DuplicateDeviceError(f"{len(records)} ranks on {distinct} distinct GPU(s): ranks "
                                   f"{[r.get('rank') for r in records if (r.get('device', {}).get('pci'), r.get('device', {}).get('uuid')) in dup]} "
                                   f"share {dup} — refusing to report an N-GPU number")
    return distinct


def spread(values) -> Optional[dict]:
    """min / mean / max of the ranks' values (None entries skipped)."""
    v = [float(x) for x in values if x is not None]
    return dict(min=min(v), mean=sum(v) / len(v), max=max(v)) if v else None


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
