"""Cross-GPU reduction of the motif-count tables.

Reads shard trivially (one process per GPU, contiguous read ranges, no data-path
collective).  The only exchange is at the end: the per-rank tables are sparse
hash tables whose slot layout differs per GPU, so they are made reducible first:

  1. all_gather of every rank's compacted key list (table, k, word);
  2. every rank builds the same sorted key dictionary;
  3. local counts are scattered into a dense int64 vector in dictionary order;
  4. ONE all_reduce(sum) of that vector (RCCL over xGMI with backend "nccl";
     gloo in the CPU tests).

allreduce_table_device() is what bench.py calls for N > 1: the rows are compacted on the device into a
fixed-capacity slice, ONE all_gather_into_tensor (RCCL) moves the slices, and one kernel merges them
into the device count table itself (the table IS a sum-merge structure) -- no host round trip between
compaction and merge, no dictionary.  allreduce_rows_into_table() is the same
exchange with host-side row arrays (gloo rehearsals and the CPU tests); allreduce_tables() /
allreduce_rows() are the dictionary + ONE all_reduce formulation of SURVEY 8(e), kept as the
reference the other two are tested against.

The payload is KB..MB, i.e. latency-bound; the per-link xGMI bandwidth does not bind.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .capi import TABLE_NAMES

_MASK63 = (1 << 63) - 1
# Rows per rank in the exchange buffer of allreduce_table_device.  A 125 M-read share of config 5 (1 B reads over 8 GPUs) leaves
# about 60 k rows in a rank's tables; twice that, so that a real 8-rank run does not sit at the retry boundary (32 B per row:
# 4 MiB per slice, 32 MiB gathered on 8 ranks).
DEFAULT_SLICE_ROWS = 1 << 17


def _to_i64(x: int) -> int:
    """uint64 -> the int64 with the same bits."""
    return x - (1 << 64) if x >= (1 << 63) else x


def _from_i64(x: int) -> int:
    return x + (1 << 64) if x < 0 else x


def tables_to_keys(tables):
    """{name: {(k, word): count}} -> (list of (t*128+k, word_lo_i64, word_hi_i64), list of counts), sorted."""
    items = []
    for t, name in enumerate(TABLE_NAMES):
        for (k, w), c in tables.get(name, {}).items():
            items.append(((t * 128 + k, _to_i64(w & 0xFFFFFFFFFFFFFFFF), _to_i64(w >> 64)), int(c)))
    items.sort()
    return [i[0] for i in items], [i[1] for i in items]


def keys_to_tables(keys, counts):
    out = {name: {} for name in TABLE_NAMES}
    for (tk, lo, hi), c in zip(keys, counts):
        if c == 0:
            continue
        t, k = divmod(int(tk), 128)
        out[TABLE_NAMES[t]][(k, (_from_i64(int(hi)) << 64) | _from_i64(int(lo)))] = int(c)
    return out


def allreduce_tables(tables, device="cpu", group=None):
    """Sum the tables of every rank; every rank returns the same merged tables."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return {name: dict(tables.get(name, {})) for name in TABLE_NAMES}
    world = dist.get_world_size(group)
    keys, counts = tables_to_keys(tables)
    n_local = torch.tensor([len(keys)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(max(sizes), 1)
    local = torch.zeros((n_max, 3), dtype=torch.int64, device=device)
    if keys:
        local[: len(keys)] = torch.tensor(keys, dtype=torch.int64, device=device)
    gathered = [torch.zeros((n_max, 3), dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(gathered, local, group=group)
    # identical sorted dictionary on every rank
    allk = set()
    for r in range(world):
        for row in gathered[r][: sizes[r]].cpu().tolist():
            allk.add(tuple(row))
    dictionary = sorted(allk)
    index = {key: i for i, key in enumerate(dictionary)}
    dense = torch.zeros(max(len(dictionary), 1), dtype=torch.int64, device=device)
    if keys:
        idx = torch.tensor([index[k] for k in keys], dtype=torch.int64, device=device)
        dense[idx] = torch.tensor(counts, dtype=torch.int64, device=device)
    dist.all_reduce(dense, op=dist.ReduceOp.SUM, group=group)
    return keys_to_tables(dictionary, dense.cpu().tolist()[: len(dictionary)])


def allreduce_rows(rows, device="cpu", group=None):
    """Array form of allreduce_tables: structured rows (capi.ROW_DTYPE) in, merged rows out.
    Same exchange (all_gather of keys -> common dictionary -> one all_reduce of the dense
    count vector), no per-row Python work."""
    import numpy as np

    from .capi import ROW_DTYPE

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rows
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    keys = np.stack([rows["table"].astype(np.int64) * 128 + rows["k"].astype(np.int64),
                     rows["word_lo"].view(np.int64), rows["word_hi"].view(np.int64)], axis=1) if len(rows) else np.zeros((0, 3), np.int64)
    n_local = torch.tensor([len(keys)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(x.item()) for x in sizes]
    n_max = max(max(sizes), 1)
    local = torch.zeros((n_max, 3), dtype=torch.int64, device=device)
    if len(keys):
        local[: len(keys)] = torch.from_numpy(keys).to(device)
    gathered = [torch.zeros((n_max, 3), dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(gathered, local, group=group)
    parts = [gathered[r][: sizes[r]].cpu().numpy() for r in range(world)]
    cat = np.concatenate(parts, axis=0) if sum(sizes) else np.zeros((0, 3), np.int64)
    if len(cat) == 0:
        return rows
    # common sorted dictionary: lexicographic sort + neighbour compare (np.unique(axis=0) is ~10x slower)
    order = np.lexsort((cat[:, 2], cat[:, 1], cat[:, 0]))
    srt = cat[order]
    new_key = np.ones(len(srt), dtype=bool)
    new_key[1:] = (srt[1:] != srt[:-1]).any(axis=1)
    dictionary = srt[new_key]
    inv = np.empty(len(srt), dtype=np.int64)
    inv[order] = np.cumsum(new_key) - 1
    off = sum(sizes[:rank])
    dense = torch.zeros(len(dictionary), dtype=torch.int64, device=device)
    if len(keys):
        dense[torch.from_numpy(inv[off: off + len(keys)]).to(device)] = torch.from_numpy(rows["count"].astype(np.int64)).to(device)
    dist.all_reduce(dense, op=dist.ReduceOp.SUM, group=group)
    out = np.zeros(len(dictionary), dtype=ROW_DTYPE)
    out["table"] = dictionary[:, 0] // 128
    out["k"] = dictionary[:, 0] % 128
    out["word_lo"] = dictionary[:, 1].view(np.uint64)
    out["word_hi"] = dictionary[:, 2].view(np.uint64)
    out["count"] = dense.cpu().numpy().view(np.uint64)
    return out


def allreduce_rows_into_table(ctx, rows, device="cpu", group=None):
    """The same reduction with the device count table as the merge structure: every rank all_gathers
    the compacted rows (32 B each: table, k, word, count), adds the OTHER ranks' rows into its own
    device table (trew_hip_add_rows: one kernel of atomic adds per rank) and compacts again.  Two
    collectives (sizes, rows) of KB..MB; no dictionary is built on the host.

    ctx: a trew_amd.capi.TrewHip (or anything with add_rows(rows) / collect_rows()) whose table holds
    exactly `rows`.  Returns the merged rows; the table of every rank then holds the global sums."""
    import numpy as np

    from .capi import ROW_DTYPE

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rows
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = torch.tensor([len(rows)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(x.item()) for x in sizes]
    n_max = max(max(sizes), 1)
    words = ROW_DTYPE.itemsize // 8
    local = torch.zeros((n_max, words), dtype=torch.int64, device=device)
    if len(rows):
        local[: len(rows)] = torch.from_numpy(np.ascontiguousarray(rows).view(np.int64).reshape(-1, words)).to(device)
    gathered = [torch.zeros((n_max, words), dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(gathered, local, group=group)
    others = [gathered[r][: sizes[r]].cpu().numpy().reshape(-1).view(ROW_DTYPE) for r in range(world) if r != rank and sizes[r]]
    if others:
        ctx.add_rows(np.concatenate(others))
    return ctx.collect_rows()


def allreduce_table_device(ctx, device, group=None, force_collectives=False, rows_on_every_rank=False):
    """The reduction bench.py uses for N > 1 with backend "nccl" (= RCCL): ONE collective, nothing leaves HBM until the
    final rows, no host round trip between the compaction and the merge.

      1. trew_hip_collect_slice_device compacts this rank's table straight into its slice of a fixed-capacity exchange buffer:
         1 + cap rows of 32 B (table, k, word, count), row 0 = header carrying the row count, written by a kernel behind the
         compaction (the collective's stream waits for it on the device: no host hop);
      2. ONE all_gather_into_tensor of the slices -- RCCL over xGMI, KB..MB, latency-bound;
      3. trew_hip_add_gathered_device: one kernel over the gathered buffer adds every OTHER rank's rows into this rank's
         device table (own slice skipped by index; ordered behind the collective on the device, by an event) -- the table
         is the sum-merge structure, no dictionary is built anywhere;
      4. rank 0 collects the merged rows; every rank's table then holds the global sums.

    Why an all_gather and not the single all_reduce the north star names: the tables are sparse hash tables whose slot
    layout differs per rank, so there is no common dense vector to reduce until the key sets have been exchanged --
    which is the all_gather; once the rows are there the sum is local (SURVEY.md section 8(e) needs two collectives for
    the same reason).  A rank with more rows than the slice holds shows in its header on every rank: nothing is added
    anywhere and all ranks repeat the exchange with larger slices (same decision everywhere, no extra collective).

    ctx: trew_amd.capi.TrewHip.  device: the torch device of ctx's GPU.  force_collectives issues the all_gather even
    at world size 1 (the one-GPU test of the RCCL path).  Returns the merged rows on rank 0 (on every rank with
    rows_on_every_rank), None elsewhere."""
    from .capi import ROW_DTYPE

    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    if not on or (world == 1 and not force_collectives):
        return ctx.collect_rows()
    rank = dist.get_rank(group)
    words = ROW_DTYPE.itemsize // 8
    cap = getattr(ctx, "_dev_rows_cap", DEFAULT_SLICE_ROWS)
    on_gpu = torch.device(device).type == "cuda"
    while True:
        bufs = getattr(ctx, "_exchange_bufs", None)
        if bufs is None or bufs[0] != (cap, world, str(device)):
            local = torch.zeros((1 + cap, words), dtype=torch.int64, device=device)
            gathered = torch.empty((world * (1 + cap), words), dtype=torch.int64, device=device)
            ctx._exchange_bufs = bufs = ((cap, world, str(device)), local, gathered)
        _, local, gathered = bufs
        stream = torch.cuda.current_stream(device).cuda_stream if on_gpu else None
        # rows 1.. and the header row (count where a row keeps its count) are written by kernels on the context's stream; the
        # collective's stream waits for them on the device -- the host neither reads the count nor writes the header
        ctx.collect_slice_device(local.data_ptr(), cap, stream)
        dist.all_gather_into_tensor(gathered, local, group=group)
        most = ctx.add_gathered_device(gathered.data_ptr(), world, rank, cap, stream)
        if most <= cap:
            break
        cap = ctx._dev_rows_cap = most + 4096  # some rank had more rows than a slice holds: nothing was added, go again
    if rows_on_every_rank or rank == 0:
        ctx._collect_cap = max(getattr(ctx, "_collect_cap", 0), world * most + 1024)  # the merged table holds at most this many rows: one compaction
        return ctx.collect_rows()
    return None  # this rank's device table holds the global sums as well; only rank 0 needs them on the host


def shard_range(n_total, rank, world):
    """Contiguous split of n_total reads over ranks: [lo, hi)."""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)
