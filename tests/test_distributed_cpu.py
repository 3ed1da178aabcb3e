"""N>1 plumbing on CPU: world_size-2 gloo run of the slice-per-rank sharding and the single result all-gather
(proof_of_burn_amd/distributed.py).  The GPU path uses the same functions with backend "nccl" (= RCCL)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from proof_of_burn_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything_once():
    for total in (0, 1, 7, 64, 1000, 8192):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


WORKER = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from proof_of_burn_amd import distributed as D
    rank, local_rank, world = D.init("gloo")
    assert world == 2
    n = 5
    lo, hi = D.shard_bounds(2 * n, rank, world)
    status = torch.arange(lo, hi, dtype=torch.int32) * (rank + 1)
    outs = (torch.arange(n * 32, dtype=torch.int64).reshape(n, 32) + 100 * rank).to(torch.uint8)
    st, out = D.gather_results(status, outs)
    assert st.tolist() == [0, 1, 2, 3, 4, 10, 12, 14, 16, 18], st.tolist()
    assert out.shape == (10, 32) and out[0, 1].item() == 1 and out[5, 1].item() == 101
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


@pytest.mark.timeout(180)
def test_gather_results_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=150)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o
