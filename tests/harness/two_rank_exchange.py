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

# default: 60 000 reads split over the ranks.  TREW_TEST_TOTAL / TREW_TEST_WORLD / TREW_TEST_RANKS / TREW_TEST_TAKE: the ranks
# play ranks TREW_TEST_RANKS (comma list) of a TREW_TEST_WORLD-way split of TREW_TEST_TOTAL reads and scan the first
# TREW_TEST_TAKE reads of their range -- config 5's read offsets (1 B reads over 8 GPUs) without config 5's size.
if "TREW_TEST_TOTAL" in os.environ:
    total, vworld = int(os.environ["TREW_TEST_TOTAL"]), int(os.environ["TREW_TEST_WORLD"])
    vranks = [int(x) for x in os.environ["TREW_TEST_RANKS"].split(",")]
    take = int(os.environ["TREW_TEST_TAKE"])
    ranges = []
    for vr in vranks:
        a, b = shard_range(total, vr, vworld)
        ranges.append((a, min(b, a + take)))
    assert len(ranges) == world
else:
    ranges = [shard_range(60000, r, world) for r in range(world)]
lo, hi = ranges[rank]
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
all_reads = []
for a, b in ranges:
    buf, st, nd = capi.synth_short_ascii(20250218, a, b - a, 150)
    all_reads += [buf[s:e + 1] for s, e in zip(st, nd)]
want = O.run_short(O.OracleParams(), all_reads)
assert got == want, "rank %d: merged tables differ from the oracle" % rank
assert on_device == want
assert mine != want  # the shard alone is not the whole
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("TWO_RANK_EXCHANGE_OK ranges", ranges)
