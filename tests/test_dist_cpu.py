"""world_size-2 gloo test of the cross-rank table reduction (the only exchange step)."""
import os
import random

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from trew_amd import capi
from trew_amd.dist import allreduce_rows, allreduce_rows_into_table, allreduce_tables, shard_range


def _make(rank):
    rnd = random.Random(100 + rank)
    t = {name: {} for name in capi.TABLE_NAMES}
    for name in capi.TABLE_NAMES:
        for _ in range(40):
            k = rnd.randint(5, 32)
            w = rnd.getrandbits(2 * k) if rnd.random() < 0.7 else (0x3FF if k >= 5 else 1)  # some keys shared by both ranks
            t[name][(k, w)] = t[name].get((k, w), 0) + rnd.randint(1, 10 ** 9)
    t["both_high"][(32, 2 ** 64 - 1)] = 7 + rank  # top bit set: exercises the u64 <-> i64 mapping
    return t


class _HostTable:
    """Stands in for the device count table of trew_amd.capi.TrewHip on a box without a GPU:
    add_rows() sums into a dict, collect_rows() returns the rows (what the HIP table does with atomics)."""

    def __init__(self, tables):
        self.t = {name: dict(d) for name, d in tables.items()}

    def add_rows(self, rows):
        for name, d in capi.rows_to_tables(rows).items():
            for key, c in d.items():
                self.t[name][key] = self.t[name].get(key, 0) + c

    def collect_rows(self):
        return capi.tables_to_rows(self.t)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = _make(rank)
    merged = allreduce_tables(mine)
    rows = allreduce_rows(capi.tables_to_rows(mine))
    via_table = allreduce_rows_into_table(_HostTable(mine), capi.tables_to_rows(mine))
    q.put((rank, merged, capi.rows_to_tables(rows), capi.rows_to_tables(via_table)))
    dist.destroy_process_group()


def test_allreduce_tables_gloo_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = {name: {} for name in capi.TABLE_NAMES}
    for r in range(world):
        for name, d in _make(r).items():
            for key, c in d.items():
                want[name][key] = want[name].get(key, 0) + c
    for rank, merged, merged_rows, merged_via_table in res:
        assert merged == want
        assert merged_rows == want
        assert merged_via_table == want


def test_single_process_passthrough_and_shards():
    t = _make(0)
    assert allreduce_tables(t) == t
    rows = capi.tables_to_rows(t)
    assert capi.rows_to_tables(allreduce_rows(rows)) == t
    assert capi.rows_to_tables(allreduce_rows_into_table(_HostTable(t), rows)) == t
    n, w = 1_000_000_007, 8
    cover = [shard_range(n, r, w) for r in range(w)]
    assert cover[0][0] == 0 and cover[-1][1] == n
    assert all(cover[i][1] == cover[i + 1][0] for i in range(w - 1))


def test_bench_gpus_n_starts_its_own_ranks_and_refuses_a_mismatch():
    """bench.py --gpus 2 started plainly (no torchrun, no WORLD_SIZE) launches two ranks itself -- before torch or HIP is
    imported -- and relays rank 0's line; with a WORLD_SIZE that contradicts --gpus it refuses instead of running a job of
    another size.  --rehearse-launch keeps the GPU out of it (this box has none): the ranks meet over gloo and are counted."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-launch"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    out = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["rehearsal"] is True
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-launch"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "refusing" in r.stderr
