"""CPU suite, world_size 2 over gloo: the multi-GPU path of bench.py shards
chunk indices with no data-path collective; only a barrier and a max-reduce of
the elapsed time cross ranks."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, chunks_per_rank, q, scaling="weak"):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from brotli_amd import synth
    from oracle_lib import OracleM4
    total = chunks_per_rank if scaling == "weak" else chunks_per_rank * world + 1  # strong: a ragged split
    own = bench.shard_chunks(total, rank, world, scaling)
    n = 1 << 14
    chunks_per_rank = len(own)
    host = np.empty(n * chunks_per_rank, dtype=np.uint8)
    synth.fill(host, n, own.start, "text", 2)
    # what each rank would produce for its shard (oracle stands in for the GPU here)
    counts = [len(OracleM4(**bench.PARAM_SETS["G"]).FindMatches(host[c * n:(c + 1) * n]))
              for c in range(chunks_per_rank)]
    # the only cross-rank traffic of the bench: barrier + max of elapsed time
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, list(own), counts, float(t.item()), host[:64].tobytes()))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_shard_disjointly(scaling):
    world, per = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, per, q, scaling)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    idx = [i for r in res for i in r[1]]
    assert idx == list(range(world * per + (scaling == "strong")))  # disjoint, ordered cover of every chunk
    assert all(r[3] == float(world) for r in res)   # max over ranks reached everyone
    # chunk content depends on the global chunk index only
    sys.path[:0] = [ROOT]
    from brotli_amd import synth
    for r in res:
        assert synth.text(1 << 14, r[1][0])[:64] == r[4]
    assert all(c > 0 for r in res for c in r[2])


def test_bench_gpus_n_refuses_a_node_with_fewer_devices():
    """`python bench.py --gpus 8` without a launcher used to measure
    one GPU silently.  It now starts the ranks itself, and exits non-zero before
    anything else when the node has fewer devices than asked for (this container
    has none)."""
    import subprocess
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node really has two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and "--gpus 2" in r.stderr, r.stderr
    env["WORLD_SIZE"] = "4"  # a launcher that started a different number of ranks than --gpus says
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)
