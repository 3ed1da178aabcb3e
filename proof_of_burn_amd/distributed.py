"""Multi-GPU driver: witnesses are independent (no cross-witness state in the circuit), so a batch is
cut into contiguous slices, one per rank / GPU; the only collective is ONE all-gather of the per-witness result records
({u32 status, u32 check_status, u32 bad_wire, u8 commitment[32]} = 44 B per witness, packed on the device by libpob_hip.so AFTER the
batch's constraint evaluation) over RCCL/xGMI -- witness vectors
never leave the GPU that produced them.  Backend "nccl" (= RCCL on ROCm) on GPUs, "gloo" for the CPU tests of this plumbing."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

RECORD_BYTES = 44            # include/pob_hip.h POB_RECORD_BYTES
NOT_EVALUATED = 0xFFFFFFFE   # POB_NOT_EVALUATED: check_status / bad_wire of a batch whose evaluation has not run
CLEAN = 0xFFFFFFFF           # check_status / bad_wire of an evaluated witness without findings


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend: str | None = None):
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # every launcher this job supports (torch.distributed.run, the driver's command line, tests through free_port()) sets MASTER_PORT: the ranks
            # of one job must agree on it, so it cannot be picked here, per process
            raise RuntimeError("WORLD_SIZE > 1 without MASTER_PORT: launch through torch.distributed.run, or set MASTER_PORT (distributed.free_port()) for every rank")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or os.environ.get("POB_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def _cpulist(text: str):
    cpus = []
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def bind_rank_to_gpu_numa(local_rank: int, device_index: int | None = None, local_world: int | None = None):
    """Pin this rank's host threads to the CPUs next to its GPU, BEFORE it allocates pinned memory and before the loader pool starts (the pool's threads and the
    first touch of the pinned input / record buffers inherit the mask): 8 ranks x 11.6 MB of inputs per 1.6 ms step is 58 GB/s out of host memory, and a rank whose
    buffers live on the other socket pulls them over the inter-socket links.  The GPU's NUMA node comes from sysfs (/sys/bus/pci/devices/<bdf>/numa_node with the bus
    id torch reports); where it is unknown (-1, containers) the CPUs this process may use are split evenly by local rank instead.  Returns the CPUs kept (None = left alone)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    local_world = local_world or int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    cpus = None
    try:
        if torch.cuda.is_available():
            dev = torch.cuda.current_device() if device_index is None else device_index
            props = torch.cuda.get_device_properties(dev)
            bdf = f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
            with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
                node = int(f.read())
            if node >= 0:
                with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
                    near = set(_cpulist(f.read()))
                cpus = [c for c in allowed if c in near]
                sharers = max(1, local_world // max(1, len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")])))
                if cpus and sharers > 1:        # the ranks that share the node split its CPUs
                    k = local_rank % sharers
                    cpus = cpus[k * len(cpus) // sharers:(k + 1) * len(cpus) // sharers] or cpus
    except Exception:
        cpus = None
    if not cpus and local_world > 1 and len(allowed) >= local_world:
        cpus = allowed[local_rank * len(allowed) // local_world:(local_rank + 1) * len(allowed) // local_world]
    if not cpus:
        return None
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:                     # (a container that forbids it: the rank runs where the launcher put it)
        return None
    os.environ["POB_RANK_BOUND"] = "1"  # (read by the loader pool's width: this rank holds ITS share of the host, csrc/pack_json.hip default_threads)
    return cpus


def host_records(calc, n: int) -> torch.Tensor:
    """the calculator's result records as a CPU tensor (no copy) when its "device" memory is host memory: the CPU shim of the tests (tests/hostsim)"""
    import ctypes
    import numpy as np
    buf = (ctypes.c_uint8 * (RECORD_BYTES * n)).from_address(calc.records_device_ptr())
    return torch.from_numpy(np.ctypeslib.as_array(buf)).view(n, RECORD_BYTES)


def free_port() -> int:
    """a TCP port the kernel says is free right now (for a launcher that starts the ranks of ONE job itself: pick once, hand it to every rank)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def shard_bounds(total: int, rank: int, world: int):
    """contiguous slice [lo, hi) of rank `rank`; sizes differ by at most one"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_records(status: torch.Tensor, outputs: torch.Tensor, check_status: torch.Tensor | None = None, bad_wire: torch.Tensor | None = None) -> torch.Tensor:
    """status int32[n], outputs uint8[n, 32] (+ the evaluator's words) -> uint8[n, 44] records (host-side twin of the device packing, for tests)"""
    n = status.shape[0]
    rec = torch.empty((n, RECORD_BYTES), dtype=torch.uint8, device=status.device)
    words = torch.full((n, 3), NOT_EVALUATED - (1 << 32), dtype=torch.int32, device=status.device)
    words[:, 0] = status.to(torch.int32)
    if check_status is not None:
        words[:, 1] = check_status.to(torch.int32)
    if bad_wire is not None:
        words[:, 2] = bad_wire.to(torch.int32)
    rec[:, :12] = words.contiguous().view(torch.uint8).view(n, 12)
    rec[:, 12:] = outputs
    return rec


def unpack_records(rec: torch.Tensor):
    """uint8[n, 44] -> (status int32[n], outputs uint8[n, 32])"""
    n = rec.shape[0]
    return rec[:, :4].contiguous().view(torch.int32).view(n), rec[:, 12:].contiguous()


def unpack_verdicts(rec: torch.Tensor):
    """uint8[n, 44] -> (check_status int64[n], bad_wire int64[n]) as unsigned values (CLEAN = no finding, NOT_EVALUATED = not run)"""
    n = rec.shape[0]
    w = rec[:, 4:12].contiguous().view(torch.int32).view(n, 2).to(torch.int64) & 0xFFFFFFFF
    return w[:, 0], w[:, 1]


def gather_records(rec: torch.Tensor, total: int | None = None) -> torch.Tensor:
    """rec uint8[n_local, 44] of this rank -> uint8[total, 44] of the whole job, with ONE all_gather_into_tensor.
    Slices may differ by one witness (shard_bounds): every rank pads to the largest slice and the result is trimmed, so an
    uneven global batch neither hangs nor mis-aligns.  total=None: every rank holds the same number of witnesses."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rec
    world = dist.get_world_size()
    n_local = rec.shape[0]
    if total is None:
        counts = [n_local] * world
    else:
        counts = [hi - lo for lo, hi in (shard_bounds(total, r, world) for r in range(world))]
        assert counts[dist.get_rank()] == n_local, "slice size does not match shard_bounds"
    n_max = max(counts)
    if dist.get_backend() == "gloo" and rec.is_cuda:          # plumbing tests of the N>1 path on one GPU: gather through host memory
        rec = rec.cpu()
    if n_local < n_max:
        rec = torch.cat([rec, torch.zeros((n_max - n_local, RECORD_BYTES), dtype=torch.uint8, device=rec.device)])
    out = torch.empty((world * n_max, RECORD_BYTES), dtype=torch.uint8, device=rec.device)
    dist.all_gather_into_tensor(out, rec.contiguous())
    if all(c == n_max for c in counts):
        return out
    return torch.cat([out[r * n_max:r * n_max + counts[r]] for r in range(world)])


class _DevBuf:
    """wrap a raw device pointer (owned by libpob_hip.so) as a torch tensor without copying"""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_records(calc, n: int) -> torch.Tensor:
    """zero-copy uint8[n, 44] view of the calculator's device-resident result records"""
    ptr = calc.records_device_ptr()
    return torch.as_tensor(_DevBuf(ptr, RECORD_BYTES * n), device=f"cuda:{calc.device}").view(n, RECORD_BYTES)
