#!/usr/bin/env python3
"""bench.py -- Gbases/s scanned, `short 5 32`, synthetic 150 bp reads, on N MI355X.

Contract (see the task prompt): `python bench.py --gpus N --steps K --warmup W`;
for N > 1 it is launched by torch.distributed.run, one rank per GPU.  A "step"
is one pass of the hot path (prefilter kernel + exact kernel + count-table
accumulation) over one batch of `--reads` synthetic reads that are already
resident in HBM when the timed region starts.  Reads are sharded contiguously
across ranks (weak scaling: every rank scans its own --reads reads); the only
exchange is the final reduction of the count tables, which is inside the timed
region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 20250218
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# integer-issue peak: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (one VALU lane-op per lane per clock)
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9
EVALS_PER_150BP_READ = 3220  # (window,k) evaluations per 150-bp read at 5 32


def usable_cores():
    """Cores this process may actually use: the affinity mask and the cgroup CPU quota, not the host's count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(round(quota / period))))
        except (OSError, ValueError, IndexError):
            continue
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step (config 2: 10M x 150 bp)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--min-mer", type=int, default=5)
    ap.add_argument("--max-mer", type=int, default=32)
    ap.add_argument("--cpu-reads", type=int, default=1_000_000, help="reads of the same workload timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="TREW_FLAG_* (debug experiments only)")
    ap.add_argument("--mode", default="short", choices=["short", "pair", "long"],
                    help="short = the BASELINE metric (config 2); pair / long = configs 3 / 4, reported for information")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import trew_amd as T
    from trew_amd import capi
    from trew_amd.dist import allreduce_rows_into_table

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = max(1, torch.cuda.device_count())
    dev_index = local_rank % n_dev if world > 1 else 0  # one GPU per rank; wraps only in a one-GPU rehearsal
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=args.backend)
    dev = torch.device("cuda", dev_index)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")
    if args.gpus != world:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    n, L = args.reads, args.read_len
    stride = 3 * ((L + 31) // 32)
    first_read = rank * n  # contiguous read-index ranges per rank
    dev_mode = {"short": T.MODE_SHORT, "pair": T.MODE_PAIR, "long": T.MODE_LONG}[args.mode]
    n_reads_dev = 2 * n if args.mode == "pair" else n  # --reads counts pairs in pair mode
    t = T.TrewHip(mode=dev_mode, min_mer=args.min_mer, max_mer=args.max_mer, device=dev.index, n_slots=1,
                  max_batch_words=16, max_batch_reads=n_reads_dev, table_log2_slots=20, flags=args.flags)
    to_free = []
    if args.mode == "short":
        d_words = t.malloc(n * stride * 4 + 64)
        t.synth_short_device(SEED, first_read, n, L, d_words)
        batch = t.device_uniform_batch(d_words, n, L)
        bases_per_step = n * L
    elif args.mode == "pair":
        d_words = t.malloc(2 * n * stride * 4 + 64)
        t.synth_pair_device(SEED, first_read, n, L, d_words)
        batch = t.device_uniform_batch(d_words, 2 * n, L)
        bases_per_step = 2 * n * L
    else:
        batch, to_free, bases_per_step = t.synth_long_device(SEED, first_read, n)
        d_words = to_free[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        t.submit(batch, 0)
        t.wait(0)

    for _ in range(args.warmup):
        step()
    if args.warmup:
        t.last_timing(0, want_flagged=False)  # drop the warm-up launches from the averages
        t.collect_rows()  # first collect allocates its device scratch: part of warm-up, not of the timed job
    t.reset_tables()
    filt_ms, exact_ms = [], []
    barrier()
    t0 = time.perf_counter()
    host_ms = []
    # the K passes are queued back to back on the slot's HIP stream (they serialise on the device,
    # launch latency hides behind the running kernels) and waited for once
    h0 = time.perf_counter()
    for _ in range(args.steps):
        t.submit(batch, 0)
    t.wait(0)
    host_ms.append((time.perf_counter() - h0) * 1e3 / args.steps)
    a, b, nflag = t.last_timing(0)  # mean over the K submits, HIP events on the kernels' own stream
    filt_ms.append(a)
    exact_ms.append(b)
    rows = t.collect_rows()
    merged = allreduce_rows_into_table(t, rows, device=comm_dev)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    bases = float(world) * bases_per_step * args.steps
    value = bases / dt / 1e9
    ms_per_step = dt / args.steps * 1e3

    out = None
    if rank == 0:
        f_avg = sum(filt_ms) / len(filt_ms)
        e_avg = sum(exact_ms) / len(exact_ms)
        dom, dom_ms = ("filter_kernel", f_avg) if f_avg >= e_avg else ("exact_kernel", e_avg)
        # algorithmic bytes per launch: 0.25 B per base (2-bit input) + 8 B per read (offset/length), SURVEY 8(d).
        # Long mode counts every base of a read although only the outer slices are touched.
        alg_bytes = bases_per_step * 0.25 + n_reads_dev * 8.0
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        evals = EVALS_PER_150BP_READ * (L / 150.0) * n if args.mode != "long" else 7420.0 * n
        # HBM bytes per launch of the dominant kernel from the committed PMC profile of this same command
        # (profiles/traffic.json, written by profiles/summarize.py); null when the configuration differs
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if args.mode == "short" and tj["config"] == {"reads_per_gpu": n, "read_len": L} and args.min_mer == 5 and args.max_mer == 32:
                traffic = round(tj["kernels"][dom]["hbm_bytes_per_launch"])
        except (OSError, KeyError, ValueError):
            traffic = None
        # VALU occupancy of the dominant kernel from the committed rocprofv3 run of this same command
        # (profiles/r01/pmc_summary_final.csv + kernel_stats_final.csv): rocprof's VALUBusy definition,
        # SQ_ACTIVE_INST_VALU * 4 / (SIMDs * kernel cycles), with the profiled average duration
        valu_busy = None
        try:
            if traffic is not None:
                import csv
                prof = os.path.join(ROOT, "profiles", "r01")
                active = dur_ns = None
                for r in csv.DictReader(open(os.path.join(prof, "pmc_summary_final.csv"))):
                    if dom in r["kernel"] and r["counter"] == "SQ_ACTIVE_INST_VALU":
                        active = float(r["avg_per_dispatch"])
                for r in csv.DictReader(open(os.path.join(prof, "kernel_stats_final.csv"))):
                    if dom in r["Name"]:
                        dur_ns = float(r["AverageNs"])
                if active and dur_ns:
                    valu_busy = round(active * 4.0 / (256 * 4 * dur_ns * 1e-9 * 2.4e9), 3)
        except (OSError, KeyError, ValueError):
            valu_busy = None
        out = {
            "metric": {"short": "Gbases/s scanned (short %d %d, %d bp reads)" % (args.min_mer, args.max_mer, L),
                       "pair": "Gbases/s scanned (short %d %d --paired_end, 2x%d bp)" % (args.min_mer, args.max_mer, L),
                       "long": "Gbases/s scanned (long %d %d, ONT-like reads, every base counted)" % (args.min_mer, args.max_mer)}[args.mode],
            "value": round(value, 3),
            "unit": "Gbases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": {"short": "short %d %d, %d synthetic %d bp reads per GPU (TTAGGG-seeded, seed %d)" % (args.min_mer, args.max_mer, n, L, SEED),
                             "pair": "short %d %d --paired_end, %d synthetic 2x%d bp pairs per GPU (seed %d)" % (args.min_mer, args.max_mer, n, L, SEED),
                             "long": "long %d %d, %d synthetic ONT-like reads per GPU (N50 ~20 kb, %.2f Gbases, seed %d)" % (args.min_mer, args.max_mer, n, bases_per_step / 1e9, SEED)}[args.mode],
                "reads_per_gpu": n,
                "read_len": L,
                "parallelism": "dp%d read-sharded, one table all-reduce" % world,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "avg_launch_ms": {"filter_kernel": round(f_avg, 4), "exact_kernel": round(e_avg, 4)},
                "valu_busy": valu_busy,
                "note": "integer-issue bound, not HBM bound (SURVEY 8(d)): valu_busy = share of SIMD cycles issuing VALU work in the committed rocprofv3 PMC run (profiles/r01); %.3g (window,k) evals/s = %.3f of the %.3g lane-op/s VALU peak at 1 lane-op per eval"
                        % (evals / ((f_avg + e_avg) * 1e-3), evals / ((f_avg + e_avg) * 1e-3) / VALU_PEAK_LANEOPS, VALU_PEAK_LANEOPS),
            },
            "flagged_reads_per_step": int(nflag),
            "host_ms_per_step": round(sum(host_ms) / len(host_ms), 4),
            "table_rows": int(len(merged)),
        }

    # CPU baseline + parity on a bounded sample of the same workload (rank 0, N = 1 only)
    if rank == 0 and world == 1 and not args.no_cpu:
        import oracle as O

        cores = usable_cores()
        op = O.OracleParams(min_mer=args.min_mer, max_mer=args.max_mer)
        t.reset_tables()
        if args.mode == "short":
            # bounded sample, ~10-30 s of CPU work: 250 k reads per usable core
            m = min(max(args.cpu_reads, cores * 250_000), n)
            buf, st, nd = capi.synth_short_ascii(SEED, 0, m, L)
            want, cpu_dt = O.run_short_mt_timed(op, buf, st, nd, cores)
            t.submit(t.device_uniform_batch(d_words, m, L), 0)
            sample_bases, used = m * L, cores
            what = "first %d reads" % m
        elif args.mode == "pair":
            m = min(args.cpu_reads // 10, n)
            b1, b2, st, nd = capi.synth_pair_ascii(SEED, 0, m, L)
            c0 = time.perf_counter()
            want = O.run_pair(op, [b1[s:e + 1] for s, e in zip(st, nd)], [b2[s:e + 1] for s, e in zip(st, nd)])
            cpu_dt = time.perf_counter() - c0
            t.submit(t.device_uniform_batch(d_words, 2 * m, L), 0)
            sample_bases, used = 2 * m * L, 1
            what = "first %d pairs" % m
        else:
            m = min(args.cpu_reads // 100, n)
            buf, st, nd = capi.synth_long_ascii(SEED, 0, m)
            c0 = time.perf_counter()
            want = O.run_long(op, [buf[s:e + 1] for s, e in zip(st, nd)])
            cpu_dt = time.perf_counter() - c0
            sub = capi.Batch(batch.words, batch.n_words, batch.offsets, batch.lengths, 0, 0, m, 1, batch.max_length)
            t.submit(sub, 0)
            sample_bases, used = int((nd - st + 1).sum()), 1
            what = "first %d reads" % m
        t.wait(0)
        got = t.collect()
        out["cpu_baseline"] = {
            "value": round(sample_bases / cpu_dt / 1e9, 6),
            "unit": "Gbases/s",
            "cores": used,
            "kind": "port",
            "sample": "%s of the same synthetic workload (%.1f Mbases), oracle/trew_oracle.c with %d thread(s), %.1f s" % (
                what, sample_bases / 1e6, used, cpu_dt),
        }
        out["parity"] = bool(got == want)
        out["parity_note"] = "GPU tables vs CPU oracle on the sample: %s" % ("bit-exact" if got == want else "MISMATCH")

    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    for ptr in (to_free or [d_words]):
        t.free(ptr)
    t.close()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and out.get("parity") is False:
        sys.exit(3)


if __name__ == "__main__":
    main()
