"""Run under torch.distributed.run with 2 ranks on ONE GPU (gloo carries the device tensors): every rank scans its shard,
the tables are reduced with trew_amd.dist.allreduce_table_device -- the exchange bench.py uses with RCCL -- and rank 0
checks the merged tables against the oracle on all the reads.  Prints TWO_RANK_EXCHANGE_OK."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (before the HIP library: one HIP runtime per process)
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()

import oracle as O  # noqa: E402
import trew_amd as T  # noqa: E402
from trew_amd import capi  # noqa: E402
from trew_amd.dist import allreduce_table_device, shard_range  # noqa: E402

n = 60000
lo, hi = shard_range(n, rank, world)
buf, st, nd = capi.synth_short_ascii(20250218, lo, hi - lo, 150)
reads = [buf[s:e + 1] for s, e in zip(st, nd)]
dev = torch.device("cuda", 0)
with T.TrewHip(mode=T.MODE_SHORT, max_batch_reads=hi - lo + 8, max_batch_words=1 << 22) as t:
    t.submit_reads(reads)
    t.wait()
    mine = t.collect()
    merged = allreduce_table_device(t, dev, rows_on_every_rank=True)
    got = capi.rows_to_tables(merged)
    on_device = t.collect()  # every rank's device table holds the global sums as well
buf, st, nd = capi.synth_short_ascii(20250218, 0, n, 150)
want = O.run_short(O.OracleParams(), [buf[s:e + 1] for s, e in zip(st, nd)])
assert got == want, "rank %d: merged tables differ from the oracle" % rank
assert on_device == want
assert mine != want  # the shard alone is not the whole
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("TWO_RANK_EXCHANGE_OK")
