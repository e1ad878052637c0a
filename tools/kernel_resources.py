#!/usr/bin/env python3
"""Compile trew_kernels.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
VGPRs, SGPRs, spills, scratch bytes, occupancy (waves per SIMD), LDS.  No GPU needed (hipcc cross-compiles).

  tools/kernel_resources.py [--csv out.csv] [--filter exact_kernel] [extra hipcc flags...]

With --isa <dir> the per-kernel instruction mix of the generated ISA (--save-temps) is tallied as well: static counts of
the half-rate VALU ops (v_alignbit, v_bcnt, 64-bit shifts), other VALU, SALU, LDS, VMEM, v_readlane / v_writelane
(SGPR spill traffic) and scratch accesses."""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "trew_amd", "csrc")
KEYS = ["TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o).replace("void ", "").replace("unsigned long", "u64").replace("unsigned __int128", "u128") for o in out]


HALF = ("v_alignbit_b32", "v_bcnt_u32_b32", "v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32")


def isa_mix(path):
    """static instruction tally per kernel symbol of one .s file"""
    mix = {}
    cur = None
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = mix.setdefault(m.group(1), collections.Counter())
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        t = line.strip().split()
        if not t or t[0].startswith((".", ";", "/")) or t[0].endswith(":"):
            continue
        op = t[0]
        if op.startswith("v_"):
            base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
            if base in ("v_readlane_b32", "v_readfirstlane_b32"):
                cur["v_readlane"] += 1
            elif base == "v_writelane_b32":
                cur["v_writelane"] += 1
            elif base in HALF:
                cur["valu_half"] += 1
            else:
                cur["valu_full"] += 1
            cur["valu"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            cur["vmem"] += 1
        elif op.startswith("scratch_"):
            cur["scratch"] += 1
    return mix


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--csv")
    ap.add_argument("--filter", default="")
    ap.add_argument("--isa", help="directory for --save-temps output; adds the static instruction mix")
    ap.add_argument("extra", nargs="*")
    a, unknown = ap.parse_known_args()
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage",
           "-c", "trew_kernels.hip", "-o", "/tmp/trew_kernels_res.o"] + a.extra + unknown
    cwd = SRC
    if a.isa:
        os.makedirs(a.isa, exist_ok=True)
        cmd += ["--save-temps=obj"]
        cmd[cmd.index("/tmp/trew_kernels_res.o")] = os.path.join(os.path.abspath(a.isa), "trew_kernels.o")
    err = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.split("\n"):
        # "<file>:<l>:<c>: remark: Key: value [-Rpass...]" or, with --save-temps, "remark: <file>:<l>:<c>: Key: value [-Rpass...]"
        m = re.search(r"remark: (?:\S+:\d+:\d+: )?\s*([A-Za-z][^:]*): (.*) \[-Rpass-analysis", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"sym": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    if not rows:
        sys.stderr.write(err[-4000:])
        sys.exit("no kernel-resource-usage remarks found")
    for r, n in zip(rows, demangle([r["sym"] for r in rows])):
        r["name"] = n
    mix = {}
    if a.isa:
        for f in os.listdir(a.isa):
            if f.endswith(".s") and "gfx950" in f:
                mix = isa_mix(os.path.join(a.isa, f))
    hdr = ["kernel", "vgpr", "sgpr", "vgpr_spill", "sgpr_spill", "scratch_B", "waves_per_simd", "lds_B"]
    mixkeys = ["valu", "valu_half", "valu_full", "v_readlane", "v_writelane", "salu", "lds", "vmem", "scratch"]
    if mix:
        hdr += mixkeys
    lines = [hdr]
    for r in rows:
        if a.filter and a.filter not in r["name"]:
            continue
        line = [r["name"], r.get("VGPRs", "?"), r.get("TotalSGPRs", "?"), r.get("VGPRs Spill", "?"), r.get("SGPRs Spill", "?"),
                r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")]
        if mix:
            c = mix.get(r["sym"], {})
            line += [str(c.get(k, 0)) for k in mixkeys]
        lines.append(line)
    w = [max(len(l[i]) for l in lines) for i in range(len(hdr))]
    for l in lines:
        print("  ".join(x.ljust(w[i]) for i, x in enumerate(l)))
    if a.csv:
        with open(a.csv, "w") as f:
            for l in lines:
                f.write(",".join(x.replace(",", ";") for x in l) + "\n")


if __name__ == "__main__":
    main()
