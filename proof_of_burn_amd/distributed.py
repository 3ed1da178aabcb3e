"""Multi-GPU driver: witnesses are independent (no cross-witness state in the circuit), so a batch is
cut into contiguous slices, one per rank / GPU; the only collective is ONE all-gather of the per-witness results
(status u32 + 32-byte commitment = 36 B per witness) over RCCL/xGMI -- witness vectors never leave the GPU that
produced them.  Backend "nccl" (= RCCL on ROCm) on GPUs, "gloo" for the CPU tests of this plumbing."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend: str | None = None):
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or os.environ.get("POB_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_bounds(total: int, rank: int, world: int):
    """contiguous slice [lo, hi) of rank `rank`; sizes differ by at most one"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_results(status: torch.Tensor, outputs: torch.Tensor):
    """status int32[n_local], outputs uint8[n_local, 32] (same n_local on every rank) -> concatenated over ranks"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return status, outputs
    world = dist.get_world_size()
    if dist.get_backend() == "gloo" and status.is_cuda:      # plumbing tests of the N>1 path on one GPU: gather through host memory
        status, outputs = status.cpu(), outputs.cpu()
    st_all = torch.empty((world * status.shape[0],), dtype=status.dtype, device=status.device)
    out_all = torch.empty((world * outputs.shape[0], 32), dtype=outputs.dtype, device=outputs.device)
    dist.all_gather_into_tensor(st_all, status.contiguous())
    dist.all_gather_into_tensor(out_all, outputs.contiguous())
    return st_all, out_all


class _DevBuf:
    """wrap a raw device pointer (owned by libpob_hip.so) as a torch tensor without copying"""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_results(calc, n: int):
    d_status, d_out = calc.results_device_ptrs()
    st = torch.as_tensor(_DevBuf(d_status, 4 * n), device="cuda").view(torch.int32)
    out = torch.as_tensor(_DevBuf(d_out, 32 * n), device="cuda").view(n, 32)
    return st, out
