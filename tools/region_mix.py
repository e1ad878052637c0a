#!/usr/bin/env python3
"""Dynamic instruction mix of exact_kernel<3, short, u64> = static class counts of every TREW_MARK region (from the ISA of the
build, --save-temps) x the trips of that region counted on the GPU by a -DTREW_REGION_COUNTS build of the library.

  1. here (no GPU):   tools/region_mix.py static profiles/r04/region_static.json      # compiles with --save-temps, classifies
  2. on the GPU box:  TREW_HIP_LIB=tools/proflib/regions/libtrew_hip.so tools/region_mix.py trips gpurun_out/region_trips.json
  3. here:            tools/region_mix.py mix profiles/r04/region_static.json gpurun_out/region_trips.json profiles/valu_rate.json profiles/r04/exact_dynamic_mix.json

Classes: VALU instructions are priced by tools/valu_rate.hip's measurement of that very opcode where it has one (profiles/
valu_rate.json, cycles_per_wave_inst), else by the class of the nearest measured relative (listed in the output as `unmeasured`,
so that nothing is priced silently).  A region that the ISA holds several times (eval_row is inlined twice) contributes the mean
of its copies.  Callees reached through s_swappc (the wave-per-segment fall-back code) are not in any region: their share is
what SQ_INSTS_VALU of the profiled run has beyond the regions' sum (`outside_regions`)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN4trew12exact_kernelILi3ELi0EmEEv"


SCALAR_SRC = re.compile(r"^(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|ttmp\d+)$")


def scalar_source(line):
    """Does a VALU instruction read a scalar register (operands after the destination; carry-out / compare destinations are
    not sources)?  Inline constants and literals are not scalar registers (measured full rate)."""
    body = line.split(";")[0].strip()
    t = body.split(None, 1)
    if len(t) < 2 or not t[0].startswith("v_"):
        return False
    operands = [o.strip() for o in t[1].split(",")]
    srcs = operands[1:]
    if re.match(r"v_(add|sub|subrev)_co_u32|v_(addc|subb|subbrev)_co_u32|v_mad_u64_u32|v_div_scale", t[0]):
        srcs = operands[2:]  # second operand is the carry-out
    return any(SCALAR_SRC.match(o.split()[0]) for o in srcs if o)


def static(out):
    d = tempfile.mkdtemp(prefix="trew_isa_")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-Wno-pass-failed", "-save-temps",
                    "-c", os.path.join(ROOT, "trew_amd", "csrc", "trew_kernels.hip"), "-o", os.path.join(d, "k.o")], cwd=d, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    path = os.path.join(d, "trew_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
    cur, inside = None, False
    copies = collections.Counter()
    ops = collections.defaultdict(collections.Counter)
    for line in open(path, errors="replace"):
        if line.startswith(KERNEL):
            inside, cur = True, "head"
            copies[cur] += 1
            continue
        if not inside:
            continue
        if "s_endpgm" in line:
            break
        m = re.search(r"; MARK (\w+)", line)
        if m:
            cur = m.group(1)
            copies[cur] += 1
            continue
        t = line.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", t[0])
        if t[0].endswith("_dpp"):
            op += "_dpp"
        elif scalar_source(line):
            op += "+s"  # a scalar source operand: half rate whatever the opcode (tools/valu_rate.hip, "(sgpr)" entries)
        ops[cur][op] += 1
    json.dump({"kernel": KERNEL, "copies": copies, "ops": ops}, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, {r: sum(c.values()) for r, c in ops.items()})


def trips(out):
    import ctypes as C
    sys.path.insert(0, ROOT)
    import trew_amd as T
    from trew_amd import capi

    lib = capi.load()
    lib.trew_debug_region_names.restype = C.c_char_p
    names = lib.trew_debug_region_names().decode().split()
    n, L = 10_000_000, 150
    t = T.TrewHip(mode=T.MODE_SHORT, n_slots=1, max_batch_words=16, max_batch_reads=n, table_log2_slots=20)
    d = t.malloc(n * 60 + 64)
    t.synth_short_device(20250218, 0, n, L, d)
    b = t.device_uniform_batch(d, n, L)
    buf = (C.c_ulonglong * 64)()
    t.submit(b, 0)
    t.wait(0)
    lib.trew_debug_regions(buf, 64, 1)  # warm-up launch dropped
    t.submit(b, 0)
    t.wait(0)
    lib.trew_debug_regions(buf, 64, 1)
    res = {"workload": "short 5 32, 10 M synthetic 150 bp reads, one launch", "flagged": int(t.last_timing(0)[2]),
           "trips": {nm: int(buf[i]) for i, nm in enumerate(names)}}
    json.dump(res, open(out, "w"), indent=1)
    print(res)


def price(op, rates, unmeasured):
    """SIMD cycles per wave-instruction of a VALU opcode."""
    full, half = rates["classes"]["full_rate"], rates["classes"]["half_rate"]
    table = rates["cycles_per_wave_inst"]
    if op.endswith("+s"):  # scalar source operand: never faster than half rate
        return max(half, price(op[:-2], rates, unmeasured))
    base = op.replace("_dpp", "")
    if op.endswith("_dpp") and "v_mov_b32_dpp" in table:
        return table["v_or_b32_dpp" if base != "v_mov_b32" and "v_or_b32_dpp" in table else "v_mov_b32_dpp"]
    if base in table:
        return table[base]
    rel = {"v_subrev_u32": "v_sub_u32", "v_not_b32": "v_xor_b32", "v_or3_b32": "v_bitop3_b32", "v_and_or_b32": "v_bitop3_b32", "v_xad_u32": "v_bitop3_b32",
           "v_add3_u32": "v_max3_u32", "v_lshl_add_u32": "v_lshl_or_b32", "v_add_lshl_u32": "v_lshl_or_b32", "v_cmp_ne_u32": "v_cmp_eq_u32", "v_cmp_lt_u32": "v_cmp_le_u32",
           "v_cmp_gt_u32": "v_cmp_le_u32", "v_cmp_ge_u32": "v_cmp_le_u32", "v_cmp_lt_i32": "v_cmp_le_u32", "v_cmp_gt_i32": "v_cmp_le_u32", "v_cmp_ge_i32": "v_cmp_le_u32",
           "v_cmp_le_i32": "v_cmp_le_u32", "v_cmp_eq_u64": "v_cmp_lt_u64", "v_cmp_ne_u64": "v_cmp_lt_u64", "v_cmp_gt_u64": "v_cmp_lt_u64", "v_ashrrev_i32": "v_lshrrev_b32",
           "v_mov_b64": "v_mov_b32", "v_sub_co_u32": "v_add_co_u32", "v_addc_co_u32": "v_add_co_u32", "v_subb_co_u32": "v_add_co_u32", "v_ffbh_u32": "v_ffbl_b32",
           "v_mbcnt_hi_u32_b32": "v_mbcnt_lo_u32_b32", "v_mul_hi_u32": "v_mul_lo_u32", "v_min3_u32": "v_max3_u32", "v_max_i32": "v_max_u32", "v_min_i32": "v_min_u32"}
    if base in rel and rel[base] in table:
        unmeasured[base] = "priced as " + rel[base]
        return table[rel[base]]
    unmeasured[base] = "no relative measured: priced half rate"
    return half


def mix(static_path, trips_path, rates_path, out):
    st, tr, rates = json.load(open(static_path)), json.load(open(trips_path)), json.load(open(rates_path))
    unmeasured = {}
    regions, tot = {}, collections.Counter()
    for r, ops in st["ops"].items():
        n = tr["trips"].get(r, None)
        if r == "head":
            n = None  # prologue + loop control: once per wave, negligible
        if n is None:
            continue
        c = max(1, st["copies"].get(r, 1))
        valu = {o: k / c for o, k in ops.items() if o.startswith("v_")}
        cyc = sum(price(o, rates, unmeasured) * k for o, k in valu.items())
        ent = {"trips": n, "copies_in_isa": c, "valu_per_trip": round(sum(valu.values()), 1), "valu_cycles_per_trip": round(cyc, 1),
               "salu_per_trip": round(sum(k for o, k in ops.items() if o.startswith("s_")) / c, 1),
               "lds_per_trip": round(sum(k for o, k in ops.items() if o.startswith("ds_")) / c, 1)}
        regions[r] = ent
        tot["valu"] += n * sum(valu.values())
        tot["valu_cycles"] += n * cyc
        tot["salu"] += n * ent["salu_per_trip"]
        full = rates["classes"]["full_rate"]
        tot["valu_full"] += n * sum(k for o, k in valu.items() if price(o, rates, {}) < 1.5 * full)
    res = {"kernel": "exact_kernel<3, short, u64>", "workload": tr["workload"], "flagged": tr["flagged"], "regions": regions,
           "dynamic": {"valu_insts_in_regions": round(tot["valu"]), "valu_simd_cycles_in_regions": round(tot["valu_cycles"]),
                       "mean_cycles_per_valu_inst": round(tot["valu_cycles"] / tot["valu"], 3),
                       "half_rate_share_of_valu_insts": round(1.0 - tot["valu_full"] / tot["valu"], 3), "salu_insts_in_regions": round(tot["salu"])},
           "unmeasured": unmeasured,
           "note": "static class counts per TREW_MARK region (straight-line: a loop inside a region is counted once per trip of the region, its own trips where it has a mark) x trips; callees behind s_swappc are outside"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["dynamic"]), "unmeasured:", unmeasured)


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "static":
        static(sys.argv[2])
    elif cmd == "trips":
        trips(sys.argv[2])
    else:
        mix(*sys.argv[2:6])
