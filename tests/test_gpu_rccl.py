"""The RCCL path of the cross-GPU table reduction on ONE GPU: backend "nccl" with world size 1, collectives really
issued (trew_amd.dist.allreduce_table_device(force_collectives=True)).  Runs in a child process because the nccl
process group has to be created before any other GPU work of the process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29800 + os.getpid() %% 150))
os.environ["RANK"] = "0"
os.environ["WORLD_SIZE"] = "1"
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))   # RCCL, before anything else touches the GPU
import numpy as np
import oracle as O
import trew_amd as T
from trew_amd import capi
from trew_amd.dist import allreduce_table_device

dev = torch.device("cuda", 0)
buf, st, nd = capi.synth_short_ascii(20250218, 0, 30000, 150)
reads = [buf[s:e + 1] for s, e in zip(st, nd)]
want = O.run_short(O.OracleParams(), reads)
with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=len(reads) + 8, max_batch_words=1 << 22) as t:
    t.submit_reads(reads)
    t.wait()
    merged = allreduce_table_device(t, dev, force_collectives=True)   # all_gather(sizes) + all_gather(rows) through RCCL
    assert capi.rows_to_tables(merged) == want, "tables changed by the RCCL exchange"
    # the gathered rows are usable as another rank's contribution: add them once more -> every count doubles
    n = len(merged)
    rows = torch.from_numpy(np.ascontiguousarray(merged).view(np.int64).reshape(n, 4)).to(dev)
    out = torch.empty((n, 4), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, rows)
    torch.cuda.synchronize()
    t.add_rows_device(out.data_ptr(), n)
    assert t.collect() == {name: {k: 2 * c for k, c in want[name].items()} for name in want}
    # an all_reduce on device memory too (the MAX over ranks of the timed region in bench.py)
    x = torch.tensor([3.5], dtype=torch.float64, device=dev)
    dist.all_reduce(x, op=dist.ReduceOp.MAX)
    assert float(x.item()) == 3.5
print("backend", dist.get_backend(), "rows", n)
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
'''


def test_rccl_collectives_on_one_gpu():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "backend nccl" in r.stdout


def test_bench_two_ranks_one_gpu_gloo_rehearsal():
    """bench.py's N > 1 code path (rank-sharded reads, barrier, reduction inside the timed region, MAX over ranks)
    with two ranks sharing this GPU over gloo -- RCCL itself refuses two ranks on one device."""
    import json

    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    port = 29950 + os.getpid() % 40
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--backend", "gloo", "--exchange", "device", "--reads", "2000000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["value"] > 0 and out["table_rows"] > 1000


def test_bench_gpus_2_launched_plainly_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (what a driver that forgets torchrun would do): bench.py starts the
    two ranks itself before it touches the GPU, relays rank 0's line, and the line says n_gpus 2 and carries exchange_ms."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo",
           "--exchange", "device", "--reads", "1500000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0 and out["table_rows"] > 1000 and out["exchange_ms"] > 0


def test_two_ranks_device_tensor_exchange_against_the_oracle():
    """allreduce_table_device (collect_device -> all_gather of device tensors -> add_rows_device), the reduction of the nccl
    runs, with two real ranks: gloo moves the device tensors, both ranks share this GPU.  Merged tables = oracle on all reads."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    port = 29900 + os.getpid() % 40
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "harness", "two_rank_exchange.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and "TWO_RANK_EXCHANGE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_two_ranks_at_config5_offsets():
    """The same two-rank device exchange with the ranks standing in for ranks 6 and 7 of config 5 (1 B reads over 8 GPUs:
    read ranges starting at 750 000 000 and 875 000 000), 40 000 reads each; merged tables = oracle on both ranges."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update(TREW_TEST_TOTAL="1000000000", TREW_TEST_WORLD="8", TREW_TEST_RANKS="6,7", TREW_TEST_TAKE="40000")
    port = 29860 + os.getpid() % 40
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "harness", "two_rank_exchange.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and "TWO_RANK_EXCHANGE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "(750000000, 750040000), (875000000, 875040000)" in r.stdout
