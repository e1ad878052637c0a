#!/usr/bin/env python3
"""bench.py -- Gbases/s scanned, `short 5 32`, synthetic 150 bp reads, on N MI355X.

Contract (see the task prompt): `python bench.py --gpus N --steps K --warmup W`;
for N > 1 it is launched by torch.distributed.run, one rank per GPU.  A "step"
is one pass of the hot path (prefilter kernel + exact kernel + count-table
accumulation) over one batch of synthetic reads that are already resident in HBM
when the timed region starts.  The K passes alternate between two batch slots
(two HIP streams), so the prefilter of pass i+1 runs beside the exact kernel of
pass i -- the same double buffering the `trew` host uses.  Reads are sharded
contiguously across ranks (weak scaling: every rank scans its own reads); the
only exchange is the final reduction of the count tables, inside the timed region.
Rank 0 prints ONE JSON line.

N = 1 (default): the BASELINE metric on config 2 (10 M x 150 bp).  The same run then
times config 3 (50 M pairs of 2 x 150 bp) and config 4 (1 M ONT-like reads) and
attaches them as `other_configs`, each with its own oracle check.
N > 1: every rank scans config 5's per-GPU share (1 B reads / 8 = 125 M reads).
"""
import argparse
import csv
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 20250218
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_SIMD = 256 * 4
CLOCK_HZ = 2.4e9
# integer-issue peak: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (one VALU lane-op per lane per clock)
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9
EVALS_PER_150BP_READ = 3220  # (window,k) evaluations per 150-bp read at 5 32
CONFIG5_READS_PER_GPU = 1_000_000_000 // 8  # BASELINE config 5: 1 B reads over 8 GPUs


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


def committed_profile(n, L):
    """HBM traffic and VALU issue accounting of both kernels from the committed rocprofv3 PMC runs of this same command
    (profiles/<round>/pmc_summary_final.csv + kernel_stats_final_serial.csv, written by profiles/summarize.py), the static
    instruction mix of the hot loops (profiles/<round>/hot_loops.json, tools/isa_loops.py on the ISA of the build) and the
    measured issue cost of the two VALU classes (profiles/valu_rate.json, tools/valu_rate.hip on the box).
    Returns (traffic_by_kernel, valu_by_kernel, round) or (None, None, None) when the configuration differs.

    Per kernel:
      issue_frac_2cyc   SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x kernel cycles): the share of the naive peak of one wave64
                        VALU instruction per SIMD-32 every 2 cycles (MI355X_MICROARCH.md)
      class_weighted    the same with what the instructions really cost: the hot loops' static split into full-rate ops
                        (xor / and / add / sub / mov / bitop3: 2.26 cycles per wave-instruction measured) and half-rate ops
                        (shifts, v_bcnt, v_alignbit, compares, v_max3, 64-bit shifts: 4.15 cycles) applied to the dynamic
                        instruction count: valu_issue = the share of the kernel's SIMD cycles in which a VALU instruction
                        occupies the pipe (half_rate / full_rate: its two parts), not_valu_issue = the rest (scalar
                        instructions of the same waves, waits, barriers, the tail of the launch)
    Kernel cycles use the clock the chip held in the profiled run (GRBM_GUI_ACTIVE / 8 XCDs / kernel time)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if tj["config"] != {"reads_per_gpu": n, "read_len": L}:
            return None, None, None
    except (OSError, KeyError, ValueError):
        return None, None, None
    traffic = {k: round(v["hbm_bytes_per_launch"]) for k, v in tj["kernels"].items()}
    rnd = os.path.basename(os.path.dirname(tj.get("source", "profiles/r01/x")))
    prof = os.path.join(ROOT, "profiles", rnd)
    valu = {}
    try:
        cnt, dur = {}, {}
        for r in csv.DictReader(open(os.path.join(prof, "pmc_summary_final.csv"))):
            for kern in ("filter_kernel", "exact_kernel"):
                if kern in r["kernel"]:
                    cnt.setdefault(kern, {})[r["counter"]] = float(r["avg_per_dispatch"])
        # durations of the kernels ALONE on the device (the PMC passes serialise dispatches): the one-stream trace
        stats = os.path.join(prof, "kernel_stats_final_serial.csv")
        if not os.path.exists(stats):
            stats = os.path.join(prof, "kernel_stats_serial.csv")
        if not os.path.exists(stats):
            stats = os.path.join(prof, "kernel_stats_final.csv")
        for r in csv.DictReader(open(stats)):
            for kern in ("filter_kernel", "exact_kernel"):
                if kern in r["Name"]:
                    dur[kern] = float(r["AverageNs"])
        rates = json.load(open(os.path.join(ROOT, "profiles", "valu_rate.json")))
        c_full, c_half = rates["classes"]["full_rate"], rates["classes"]["half_rate"]
        try:
            hot = json.load(open(os.path.join(prof, "hot_loops.json")))
        except (OSError, ValueError):
            hot = {}
        for kern in cnt:
            c = cnt[kern]
            if "SQ_INSTS_VALU" not in c or kern not in dur:
                continue
            clock = c["GRBM_GUI_ACTIVE"] / 8.0 / (dur[kern] * 1e-9) if "GRBM_GUI_ACTIVE" in c else CLOCK_HZ
            simd_cycles = N_SIMD * dur[kern] * 1e-9 * clock
            entry = {
                "insts_valu": round(c["SQ_INSTS_VALU"]),
                "insts_salu": round(c.get("SQ_INSTS_SALU", 0)),
                "profiled_launch_ms": round(dur[kern] * 1e-6, 4),
                "clock_ghz": round(clock * 1e-9, 3),
                "issue_frac_2cyc": round(c["SQ_INSTS_VALU"] * 2.0 / simd_cycles, 3),
                "waves_resident_avg_per_simd": round(c["SQ_WAVE_CYCLES"] * 4.0 / simd_cycles, 2) if "SQ_WAVE_CYCLES" in c else None,
            }
            h = hot.get(kern)
            dyn = None
            if kern == "exact_kernel":  # the exact kernel has no dominant loop: its mix is the DYNAMIC one of tools/region_mix.py
                try:
                    dyn = json.load(open(os.path.join(prof, "exact_dynamic_mix.json")))
                except (OSError, ValueError):
                    dyn = None
            if dyn:
                share = dyn["dynamic"]["half_rate_share_of_valu_insts"]
                mix_from = ("dynamic: static class counts of every TREW_MARK region x the region's trips on the GPU (tools/region_mix.py, "
                            "profiles/%s/exact_dynamic_mix.json: the regions' straight-line counts x trips model %.0f M VALU instructions against the %.0f M "
                            "the counter saw -- both sides of wave-uniform branches are counted -- so only the class SHARE is taken from it), every opcode "
                            "priced by tools/valu_rate.hip's measurement" % (rnd, dyn["dynamic"]["valu_insts_in_regions"] / 1e6, c["SQ_INSTS_VALU"] / 1e6))
            elif h:
                share, mix_from = h["half_rate_share_of_valu_insts"], "static: " + h["from"]
            else:
                share = None
            if share is not None:
                half = c["SQ_INSTS_VALU"] * share * c_half / simd_cycles
                full = c["SQ_INSTS_VALU"] * (1.0 - share) * c_full / simd_cycles
                entry["class_weighted"] = {
                    "half_rate_share_of_valu_insts": round(share, 3),
                    "half_rate_issue": round(half, 3),
                    "full_rate_issue": round(full, 3),
                    "valu_issue": round(half + full, 3),
                    "not_valu_issue": round(1.0 - half - full, 3),
                    "salu_per_valu_inst": round(c.get("SQ_INSTS_SALU", 0) / c["SQ_INSTS_VALU"], 3),
                    "lanes_per_valu_inst": round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"], 1) if "SQ_THREAD_CYCLES_VALU" in c else None,
                    "mix_from": mix_from,
                }
            valu[kern] = entry
        valu["cycles_per_wave_inst"] = {"full_rate": c_full, "half_rate": c_half}
        valu["source"] = "profiles/%s/pmc_summary_final.csv (counters), %s (durations alone on the device), profiles/%s/hot_loops.json (tools/isa_loops.py), profiles/valu_rate.json (tools/valu_rate.hip on the box)" % (
            rnd, os.path.basename(stats), rnd)
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        valu = {}
    return traffic, (valu or None), rnd


class Workload:
    """One BASELINE configuration resident in HBM: context, device batch, bookkeeping."""

    def __init__(self, T, capi, mode, n, L, args, dev_index, first_read):
        self.T, self.capi, self.mode, self.n, self.L = T, capi, mode, n, L
        dev_mode = {"short": T.MODE_SHORT, "pair": T.MODE_PAIR, "long": T.MODE_LONG}[mode]
        self.n_reads_dev = 2 * n if mode == "pair" else n  # n counts pairs in pair mode
        self.t = t = T.TrewHip(mode=dev_mode, min_mer=args.min_mer, max_mer=args.max_mer, device=dev_index, n_slots=max(1, args.streams),
                               max_batch_words=16, max_batch_reads=self.n_reads_dev, table_log2_slots=22, flags=args.flags)
        stride = 3 * ((L + 31) // 32)
        self.to_free = []
        if mode == "short":
            self.d_words = t.malloc(n * stride * 4 + 64)
            t.synth_short_device(SEED, first_read, n, L, self.d_words)
            self.batch = t.device_uniform_batch(self.d_words, n, L)
            self.bases_per_step = n * L
            self.to_free = [self.d_words]
        elif mode == "pair":
            self.d_words = t.malloc(2 * n * stride * 4 + 64)
            t.synth_pair_device(SEED, first_read, n, L, self.d_words)
            self.batch = t.device_uniform_batch(self.d_words, 2 * n, L)
            self.bases_per_step = 2 * n * L
            self.to_free = [self.d_words]
        else:
            self.batch, self.to_free, self.bases_per_step = t.synth_long_device(SEED, first_read, n)
            self.d_words = self.to_free[0]

    def close(self):
        for ptr in self.to_free:
            self.t.free(ptr)
        self.t.close()

    def sub_batch(self, m):
        """The first m units of the resident batch."""
        if self.mode == "short":
            return self.t.device_uniform_batch(self.d_words, m, self.L)
        if self.mode == "pair":
            return self.t.device_uniform_batch(self.d_words, 2 * m, self.L)
        b = self.batch
        return self.capi.Batch(b.words, b.n_words, b.offsets, b.lengths, 0, 0, m, 1, b.max_length)


def timed_run(w, args, steps, warmup, world, barrier, reduce_fn):
    """W warm-up passes, then exactly `steps` passes between barriers.  Returns a dict of measurements."""
    t, nslots = w.t, max(1, args.streams)
    for i in range(warmup):
        t.submit(w.batch, i % nslots)
    for s in range(nslots):
        t.wait(s)
    serial = None
    if warmup:
        for s in range(nslots):
            try:
                t.last_timing(s, want_flagged=False)  # drop the warm-up launches from the averages
            except w.T.TrewHipError:
                pass
        t.collect_rows()  # first collect allocates its device scratch: part of warm-up, not of the timed job
    # per-kernel durations with the device to themselves: two passes on ONE stream (not part of the timed region)
    t.submit(w.batch, 0)
    t.submit(w.batch, 0)
    t.wait(0)
    serial = t.last_timing(0, want_flagged=False)[:2]
    t.reset_tables()
    barrier()
    t0 = time.perf_counter()
    # the K passes are queued on the slots' HIP streams round-robin (each slot serialises its own passes, the
    # slots overlap each other; launch latency hides behind the running kernels) and waited for once
    for i in range(steps):
        t.submit(w.batch, i % nslots)
    for s in range(nslots):
        t.wait(s)
    host_ms = (time.perf_counter() - t0) * 1e3 / steps
    rows = t.collect_rows() if world == 1 else None
    t1 = time.perf_counter()
    merged = reduce_fn(t, rows)  # N > 1: the table exchange, inside the timed region
    exchange_ms = (time.perf_counter() - t1) * 1e3
    barrier()
    dt = time.perf_counter() - t0
    # mean HIP-event durations of the timed passes, on the kernels' own streams
    fs, es, cnt, nflag = 0.0, 0.0, 0, 0
    for s in range(min(nslots, steps)):
        k = len(range(s, steps, nslots))
        a, b, nf = t.last_timing(s)
        fs, es, cnt, nflag = fs + a * k, es + b * k, cnt + k, nf
    return {"dt": dt, "host_ms": host_ms, "filter_ms": fs / cnt, "exact_ms": es / cnt, "serial_ms": serial, "nflag": int(nflag), "rows": merged, "exchange_ms": exchange_ms}


def oracle_check(w, args, O, cores):
    """GPU tables vs the CPU oracle on a bounded prefix of the workload; returns (ok, sample text, bases, seconds, threads)."""
    t, capi = w.t, w.capi
    op = O.OracleParams(min_mer=args.min_mer, max_mer=args.max_mer)
    t.reset_tables()
    if w.mode == "short":
        m = min(max(args.cpu_reads, cores * 250_000), w.n)  # ~10-30 s of CPU work: 250 k reads per usable core
        buf, st, nd = capi.synth_short_ascii(SEED, 0, m, w.L)
        want, cpu_dt = O.run_short_mt_timed(op, buf, st, nd, cores)
        bases, used, what = m * w.L, cores, "first %d reads" % m
    elif w.mode == "pair":
        m = min(args.cpu_pairs, w.n)
        b1, b2, st, nd = capi.synth_pair_ascii(SEED, 0, m, w.L)
        c0 = time.perf_counter()
        want = O.run_pair(op, [b1[s:e + 1] for s, e in zip(st, nd)], [b2[s:e + 1] for s, e in zip(st, nd)])
        cpu_dt = time.perf_counter() - c0
        bases, used, what = 2 * m * w.L, 1, "first %d pairs" % m
    else:
        m = min(args.cpu_long_reads, w.n)
        buf, st, nd = capi.synth_long_ascii(SEED, 0, m)
        c0 = time.perf_counter()
        want = O.run_long(op, [buf[s:e + 1] for s, e in zip(st, nd)])
        cpu_dt = time.perf_counter() - c0
        bases, used, what = int((nd - st + 1).sum()), 1, "first %d reads" % m
    t.submit(w.sub_batch(m), 0)
    t.wait(0)
    got = t.collect()
    return got == want, what, bases, cpu_dt, used


def write_fastq(path, capi, n, L, chunk=1_000_000):
    """n synthetic reads of the bench workload as a plain FASTQ file (records "@r / seq / + / quality", large writes)."""
    import numpy as np

    with open(path, "wb") as f:
        for lo in range(0, n, chunk):
            m = min(chunk, n - lo)
            buf, _, _ = capi.synth_short_ascii(SEED, lo, m, L)
            b = np.frombuffer(buf, dtype=np.uint8).reshape(m, L + 1)
            rec = np.zeros((m, 3 + (L + 1) + 2 + (L + 1)), dtype=np.uint8)
            rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
            rec[:, 3:3 + L + 1] = b
            rec[:, 4 + L:6 + L] = np.frombuffer(b"+\n", dtype=np.uint8)
            rec[:, 6 + L:6 + 2 * L] = ord("I")
            rec[:, 6 + 2 * L] = ord("\n")
            rec.tofile(f)


def end_to_end(capi, args, cores):
    """The `trew` binary on a FASTQ file of the same workload in page cache: FASTQ text -> CSV, PCIe-inclusive.  Never the
    bench `value`; reported beside it.  Returns a dict or None (no writable scratch directory)."""
    import re
    import shutil
    import subprocess
    import tempfile

    trew = os.path.join(ROOT, "trew_amd", "bin", "trew")
    if not os.path.exists(trew):
        return None
    n = args.e2e_reads
    need = n * (2 * args.read_len + 8) * 1.05
    base = None
    for cand in ("/dev/shm", tempfile.gettempdir()):
        try:
            if os.path.isdir(cand) and shutil.disk_usage(cand).free > need * 1.2:
                base = cand
                break
        except OSError:
            continue
    if base is None:
        return None
    d = tempfile.mkdtemp(prefix="trew_e2e_", dir=base)
    try:
        path = os.path.join(d, "e2e.fastq")
        write_fastq(path, capi, n, args.read_len)
        threads = max(2, cores)
        best = None
        for rep in range(2):  # the first run also pays for the module load and the page-cache warm-up of the mapping
            r = subprocess.run([trew, "short", str(args.min_mer), str(args.max_mer), path, "-t", str(threads), "--stats"], capture_output=True, text=True, timeout=600)
            m = re.search(r"([0-9.]+) s, ([0-9.]+) Gbases/s end-to-end \(decode \+ pack \+ scan; ([^,]+),", r.stderr)
            if r.returncode != 0 or not m:
                return {"error": (r.stderr or "")[-300:]}
            if best is None or float(m.group(2)) > best["gbases_s"]:
                pw = re.search(r"newline scan \+ line chain ([0-9.]+), wait for the slot ([0-9.]+), pack ([0-9.]+), submit ([0-9.]+)", r.stderr)
                phases = dict(zip(("scan", "wait_slot", "pack_or_copy", "submit"), (float(x) for x in pw.groups()))) if pw else {}
                best = {"gbases_s": float(m.group(2)), "seconds": float(m.group(1)), "threads": threads, "reads": n, "reader": m.group(3).strip(),
                        "worker_seconds": phases, "bound_by": ("host: " + max(phases, key=phases.get)) if phases else None,
                        "input": "plain FASTQ, %.1f GB of text in page cache (%s), CSV to a pipe" % (os.path.getsize(path) / 1e9, base),
                        "note": "PCIe-inclusive, FASTQ text to CSV; never the bench value"}
        return best
    finally:
        shutil.rmtree(d, ignore_errors=True)


def launch_ranks(n):
    """Start `n` ranks of this very command under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) as a
    child process, relay its output, return its exit status.  Nothing here imports torch or touches the GPU."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:  # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def rehearse_launch(args):
    """--rehearse-launch: the ranks meet, agree on the world size and leave; no GPU, no scan (see the option's help)."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        dist.barrier()
        seen = int(t.item())
        dist.destroy_process_group()
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": seen, "steps": args.steps, "warmup": args.warmup, "value": None,
                          "note": "launch rehearsal only: the ranks were started and counted, nothing was scanned"}))
        sys.stdout.flush()
    return 0 if seen == args.gpus else 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=0, help="reads (pairs) per GPU per step; 0 = the BASELINE size of the mode: "
                    "short 10 M at N = 1 (config 2) and 125 M at N > 1 (config 5's share), pair 50 M (config 3), long 1 M (config 4)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--min-mer", type=int, default=5)
    ap.add_argument("--max-mer", type=int, default=32)
    ap.add_argument("--cpu-reads", type=int, default=1_000_000, help="reads of the same workload timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--cpu-pairs", type=int, default=30_000, help="pairs of config 3 checked against the oracle")
    ap.add_argument("--cpu-long-reads", type=int, default=5_000, help="reads of config 4 checked against the oracle")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="N = 1: do not run the `trew` binary on a FASTQ file of the workload")
    ap.add_argument("--e2e-reads", type=int, default=48_000_000, help="reads of the end-to-end FASTQ file (14.7 GB of text at 48 M x 150 bp: long enough that start-up does not show)")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1, --mode short only: do not time configs 3 and 4")
    ap.add_argument("--other-steps", type=int, default=0, help="passes per other config (default: min(steps, 10))")
    ap.add_argument("--streams", type=int, default=2, help="batch slots (HIP streams) the passes alternate between")
    ap.add_argument("--flags", type=int, default=0, help="TREW_FLAG_* (debug experiments only)")
    ap.add_argument("--mode", default="short", choices=["short", "pair", "long"],
                    help="short = the BASELINE metric (config 2); pair / long = configs 3 / 4 as the primary line")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "device", "host"],
                    help="table reduction: device = collect_device -> all_gather of device tensors -> add_rows_device (what nccl runs use); "
                         "host = the same exchange with host-side row arrays; auto = device with nccl, host otherwise")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="test aid (no GPU needed): every rank joins the process group, the ranks agree on the world size with one all_reduce, "
                         "rank 0 prints {n_gpus, rehearsal: true} and nothing is scanned -- exercises the self-launch of --gpus N on a CPU box")
    args = ap.parse_args()

    # `python bench.py --gpus N` started plainly: start the N ranks ourselves, BEFORE torch or HIP is imported or any GPU call is
    # made in this process (a process that has touched the GPU must not be replaced or forked into ranks).  The children run
    # under torch.distributed.run exactly as the driver would start them; rank 0's JSON line is relayed, the exit status is theirs.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world_env:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to run a job of another size than the one asked for" % (args.gpus, world_env))
    if args.rehearse_launch:
        sys.exit(rehearse_launch(args))

    import torch
    import torch.distributed as dist

    import trew_amd as T
    from trew_amd import capi
    from trew_amd.dist import allreduce_rows_into_table, allreduce_table_device

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        sys.exit("bench.py: no GPU visible (the HIP path has no CPU fallback)")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world > 1 and args.backend == "nccl" and local_world > n_dev:
        sys.exit("bench.py: %d ranks on this node but %d GPU(s) visible -- RCCL needs one GPU per rank "
                 "(use --backend gloo to rehearse several ranks on one GPU)" % (local_world, n_dev))
    dev_index = local_rank % n_dev if world > 1 else 0  # one GPU per rank; wraps only in a one-GPU gloo rehearsal
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=args.backend)
    dev = torch.device("cuda", dev_index)
    L = args.read_len
    default_reads = {"short": 10_000_000 if world == 1 else CONFIG5_READS_PER_GPU, "pair": 50_000_000, "long": 1_000_000}
    n = args.reads or default_reads[args.mode]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_fn(t, rows):
        if world == 1:
            return rows
        if args.exchange == "device" or (args.exchange == "auto" and args.backend == "nccl"):
            return allreduce_table_device(t, dev)  # compaction -> all_gather of device tensors (RCCL) -> add into the device table, all in HBM
        return allreduce_rows_into_table(t, t.collect_rows(), device=torch.device("cpu"))

    w = Workload(T, capi, args.mode, n, L, args, dev.index, rank * n)  # contiguous read-index ranges per rank
    m = timed_run(w, args, args.steps, args.warmup, world, barrier, reduce_fn)
    dt = m["dt"]
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    bases = float(world) * w.bases_per_step * args.steps
    value = bases / dt / 1e9
    ms_per_step = dt / args.steps * 1e3

    def workload_name(mode, cnt, bases_per_step):
        return {"short": "short %d %d, %d synthetic %d bp reads per GPU (TTAGGG-seeded, seed %d)" % (args.min_mer, args.max_mer, cnt, L, SEED),
                "pair": "short %d %d --paired_end, %d synthetic 2x%d bp pairs per GPU (seed %d)" % (args.min_mer, args.max_mer, cnt, L, SEED),
                "long": "long %d %d, %d synthetic ONT-like reads per GPU (N50 ~20 kb, %.2f Gbases, seed %d)" % (args.min_mer, args.max_mer, cnt, bases_per_step / 1e9, SEED)}[mode]

    out = None
    if rank == 0:
        f_avg, e_avg = m["filter_ms"], m["exact_ms"]
        dom, dom_ms = ("filter_kernel", f_avg) if f_avg >= e_avg else ("exact_kernel", e_avg)
        # algorithmic bytes per launch: 0.25 B per base (2-bit input) + 8 B per read (offset/length), SURVEY 8(d).
        # Long mode counts every base of a read although only the outer slices are touched.
        alg_bytes = w.bases_per_step * 0.25 + w.n_reads_dev * 8.0
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        evals = EVALS_PER_150BP_READ * (L / 150.0) * n if args.mode != "long" else 7420.0 * n
        traffic_all, valu, prof_round = (None, None, None)
        if args.mode == "short" and args.min_mer == 5 and args.max_mer == 32:
            traffic_all, valu, prof_round = committed_profile(n, L)
        # counter traffic of one STEP (both kernels of a slot; FETCH_SIZE x 2 + WRITE_SIZE per MI355X_MICROARCH.md), with the split
        traffic = round(sum(traffic_all.values())) if traffic_all else None
        s_f, s_e = m["serial_ms"]
        s_dom = s_f if dom == "filter_kernel" else s_e
        workload = workload_name(args.mode, n, w.bases_per_step)
        if world > 1 and args.mode == "short" and n == CONFIG5_READS_PER_GPU:
            workload += "; config 5's per-GPU share (1 B reads / 8)"
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
                "workload": workload,
                "reads_per_gpu": n,
                "read_len": L,
                "streams": max(1, args.streams),
                "parallelism": "dp%d read-sharded, tables reduced once at the end%s" % (
                    world, "" if world == 1 else (" (RCCL all_gather of compacted rows, merged by the device table)" if args.backend == "nccl" else " (gloo rehearsal)")),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_by_kernel": traffic_all,
                "traffic_over_algorithmic": round(traffic / alg_bytes, 3) if traffic else None,
                "avg_launch_ms": {"filter_kernel": round(f_avg, 4), "exact_kernel": round(e_avg, 4)},
                "slot_cycle_ms": round(f_avg + e_avg, 4),  # one slot runs its prefilter then its exact kernel: with S slots a step takes >= slot_cycle_ms / S
                "serial_launch_ms": {"filter_kernel": round(s_f, 4), "exact_kernel": round(s_e, 4)},
                "frac_serial": round(alg_bytes / (s_dom * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "valu": valu,
                # MODELLED, not a counter: SQ_INSTS_VALU of the committed profile x the measured cost of its instruction classes
                "valu_issue_modelled": ((valu or {}).get(dom, {}).get("class_weighted") or {}).get("valu_issue"),
                "note": "integer-issue bound, not HBM bound (SURVEY 8(d)). avg_launch_ms: HIP events over the timed region, where the %d slots overlap "
                        "(a kernel shares the SIMDs with the other slot's kernel, so its own launch is longer than alone); serial_launch_ms: the same "
                        "kernels alone on one stream, measured in this run outside the timed region.  traffic: HBM bytes of one step from the committed PMC runs "
                        "(both kernels; achieved / frac use the ALGORITHMIC bytes).  valu: SQ counters of the committed rocprofv3 PMC "
                        "runs of this command (profiles/%s): issue_frac_2cyc = SQ_INSTS_VALU x 2 / SIMD cycles, class_weighted = the same with the measured cost of full-rate and half-rate instructions (profiles/valu_rate.json) -- the prefilter's mix is the static one of its dominant loop, the exact kernel's the dynamic one of tools/region_mix.py; see bench.py::committed_profile.  %.3g (window,k) "
                        "evals/s = %.3f of the %.3g lane-op/s VALU peak at 1 lane-op per eval"
                        % (max(1, args.streams), prof_round or "none for this configuration", evals / (ms_per_step * 1e-3),
                           evals / (ms_per_step * 1e-3) / VALU_PEAK_LANEOPS, VALU_PEAK_LANEOPS),
            },
            "flagged_reads_per_step": m["nflag"],
            "reads_per_s": round(world * n * args.steps / dt, 1),
            "host_ms_per_step": round(m["host_ms"], 4),
            "table_rows": int(len(m["rows"])) if m["rows"] is not None else None,
        }
        if world > 1:
            # rank 0's wall time of the one exchange of the job (compaction -> all_gather -> merge kernel -> final rows), part of dt
            out["exchange_ms"] = round(m["exchange_ms"], 3)

    # CPU baseline + parity on a bounded sample of the same workload (rank 0, N = 1 only)
    do_cpu = rank == 0 and world == 1 and not args.no_cpu
    O = None
    if do_cpu:
        import oracle as O

        cores = usable_cores()
        ok, what, sample_bases, cpu_dt, used = oracle_check(w, args, O, cores)
        out["cpu_baseline"] = {
            "value": round(sample_bases / cpu_dt / 1e9, 6),
            "unit": "Gbases/s",
            "cores": used,
            "kind": "port",
            "sample": "%s of the same synthetic workload (%.1f Mbases), oracle/trew_oracle.c with %d thread(s), %.1f s" % (
                what, sample_bases / 1e6, used, cpu_dt),
        }
        out["parity"] = bool(ok)
        out["parity_note"] = "GPU tables vs CPU oracle on the sample: %s" % ("bit-exact" if ok else "MISMATCH")
    w.close()

    # configs 3 and 4 at BASELINE size in the same run (N = 1, default invocation), each checked against the oracle
    if rank == 0 and world == 1 and args.mode == "short" and not args.no_other_configs:
        others = []
        osteps = args.other_steps or max(1, min(args.steps, 10))
        for mode in ("pair", "long"):
            cnt = default_reads[mode]
            ow = Workload(T, capi, mode, cnt, L, args, dev.index, 0)
            om = timed_run(ow, args, osteps, min(args.warmup, 2), 1, barrier, reduce_fn)
            entry = {
                "workload": workload_name(mode, cnt, ow.bases_per_step),
                "value": round(ow.bases_per_step * osteps / om["dt"] / 1e9, 3),
                "unit": "Gbases/s",
                "steps": osteps,
                "ms_per_step": round(om["dt"] / osteps * 1e3, 4),
                "reads_per_s": round(cnt * osteps / om["dt"], 1),
                "avg_launch_ms": {"filter_kernel": round(om["filter_ms"], 4), "exact_kernel": round(om["exact_ms"], 4)},
                "serial_launch_ms": {"filter_kernel": round(om["serial_ms"][0], 4), "exact_kernel": round(om["serial_ms"][1], 4)},
                "flagged_per_step": om["nflag"],
                "table_rows": int(len(om["rows"])),
            }
            if mode == "long":
                entry["note"] = "every base of a read is counted although only the chained outer slices are loaded (SURVEY 8(d)): reads_per_s is the size-independent figure"
            if do_cpu:
                ok, what, sample_bases, cpu_dt, used = oracle_check(ow, args, O, 1)
                entry["parity"] = bool(ok)
                entry["parity_sample"] = "%s (%.1f Mbases) vs oracle/trew_oracle.c, %.1f s on 1 thread" % (what, sample_bases / 1e6, cpu_dt)
                if not ok:
                    out["parity"] = False
                    out["parity_note"] += "; %s MISMATCH" % mode
            ow.close()
            others.append(entry)
        out["other_configs"] = others

    if rank == 0 and world == 1 and args.mode == "short" and not args.no_e2e and not args.no_cpu:
        try:
            out["e2e"] = end_to_end(capi, args, usable_cores())
        except Exception as ex:  # the end-to-end leg must never take the bench line down
            out["e2e"] = {"error": repr(ex)[:300]}
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and out.get("parity") is False:
        sys.exit(3)


if __name__ == "__main__":
    main()
