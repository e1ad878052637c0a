#!/usr/bin/env python3
"""Static instruction mix of the hot loops of a kernel, from the gfx950 ISA hipcc writes with --save-temps
(tools/kernel_resources.py --isa DIR).

  tools/isa_loops.py DIR/trew_kernels-hip-amdgcn-amd-amdhsa-gfx950.s 'filter_kernelILi3E' [--min-bcnt 6] [--json out.json]

A loop is a backward branch: the lines from the branch target to the branch.  Innermost loops with at least --min-bcnt
v_bcnt instructions are the prefilter's k loops.  Every VALU instruction is put in the full-rate or the half-rate class of
profiles/valu_rate.json (tools/valu_rate.hip measured them on the MI355X); SALU, SMEM, LDS, VMEM and branches are counted
beside them.  The functions a kernel calls (noinline device functions) are tallied as whole bodies with --functions."""
import argparse
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_classes():
    r = json.load(open(os.path.join(ROOT, "profiles", "valu_rate.json")))
    return set(r["classes"]["full_rate_ops"]), r["classes"]["full_rate"], r["classes"]["half_rate"]


SCALAR_SRC = re.compile(r"^(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|ttmp\d+)$")


def scalar_source(line):
    """A VALU instruction that reads a scalar register occupies the SIMD like a half-rate one whatever its opcode
    (tools/valu_rate.hip, the "(sgpr)" entries); inline constants and literals do not."""
    t = line.split(";")[0].strip().split(None, 1)
    if len(t) < 2:
        return False
    operands = [o.strip() for o in t[1].split(",")]
    srcs = operands[2:] if re.match(r"v_(add|sub|subrev|addc|subb|subbrev)_co_u32|v_mad_u64_u32", t[0]) else operands[1:]
    return any(SCALAR_SRC.match(o.split()[0]) for o in srcs if o)


def classify(op, full, line=""):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith("v_"):
        if base in ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32"):
            return "valu_lane"
        return "valu_full" if base in full and not scalar_source(line) else "valu_half"
    if base.startswith("s_load") or base.startswith("s_buffer_load"):
        return "smem"
    if base.startswith("s_cbranch") or base == "s_branch":
        return "branch"
    if base in ("s_waitcnt", "s_nop", "s_barrier"):
        return "wait_nop"
    if base.startswith("s_"):
        return "salu"
    if base.startswith("ds_"):
        return "lds"
    if base.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def function_lines(path, pattern):
    out, name, on = [], None, False
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", line)
        if m and not on:
            if pattern in m.group(1):
                on, name = True, m.group(1)
            continue
        if on:
            if line.startswith(".Lfunc_end"):
                break
            out.append(line.rstrip("\n"))
    return name, out


def tally(lines, full):
    c = collections.Counter()
    ops = collections.Counter()
    for ln in lines:
        t = ln.strip().split()
        if not t or t[0].startswith((".", ";", "/")) or t[0].endswith(":"):
            continue
        cls = classify(t[0], full, ln)
        c[cls] += 1
        if cls.startswith("valu"):
            ops[re.sub(r"_(e32|e64|dpp|sdwa)$", "", t[0])] += 1
    return c, ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("isa")
    ap.add_argument("kernel")
    ap.add_argument("--min-bcnt", type=int, default=6)
    ap.add_argument("--json")
    ap.add_argument("--functions", nargs="*", default=[], help="substrings of device functions to tally as whole bodies")
    a = ap.parse_args()
    full, c_full, c_half = load_classes()
    name, lines = function_lines(a.isa, a.kernel)
    if not lines:
        sys.exit("kernel not found")
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, ln in enumerate(lines):
        t = ln.strip().split()
        if t and (t[0].startswith("s_cbranch") or t[0] == "s_branch") and t[-1] in labels and labels[t[-1]] < i:
            loops.append((labels[t[-1]], i))
    # innermost: no other loop strictly inside
    inner = [lp for lp in loops if not any(o != lp and o[0] >= lp[0] and o[1] <= lp[1] for o in loops)]
    report = {"kernel": name, "cycles_full_rate": c_full, "cycles_half_rate": c_half, "loops": [], "functions": {}}
    for a0, b0 in inner:
        c, ops = tally(lines[a0:b0 + 1], full)
        if ops.get("v_bcnt_u32_b32", 0) < a.min_bcnt:
            continue
        valu = c["valu_full"] + c["valu_half"] + c["valu_lane"]
        cyc = c["valu_full"] * c_full + (c["valu_half"] + c["valu_lane"]) * c_half
        entry = {"label": lines[a0].split(":")[0], "lines": [a0, b0], "valu": valu, "valu_full_rate": c["valu_full"], "valu_half_rate": c["valu_half"] + c["valu_lane"],
                 "salu": c["salu"], "smem": c["smem"], "branch": c["branch"], "wait_nop": c["wait_nop"], "lds": c["lds"], "vmem": c["vmem"],
                 "valu_issue_cycles": round(cyc, 1), "half_rate_share_of_valu_cycles": round((c["valu_half"] + c["valu_lane"]) * c_half / cyc, 3) if cyc else 0,
                 "ops": dict(ops.most_common())}
        report["loops"].append(entry)
        print("loop %-12s valu %3d (full %3d, half %3d)  salu %3d  smem %d  branch %d  -> %.0f VALU issue cycles per trip, %.0f %% of them half-rate ops" % (
            entry["label"], valu, entry["valu_full_rate"], entry["valu_half_rate"], c["salu"], c["smem"], c["branch"], cyc, 100 * entry["half_rate_share_of_valu_cycles"]))
    for fn in [a.kernel] + a.functions:
        n2, l2 = function_lines(a.isa, fn)
        if not l2:
            continue
        c, ops = tally(l2, full)
        cyc = c["valu_full"] * c_full + (c["valu_half"] + c["valu_lane"]) * c_half
        report["functions"][n2] = {"valu_full_rate": c["valu_full"], "valu_half_rate": c["valu_half"], "valu_lane_ops": c["valu_lane"], "salu": c["salu"], "smem": c["smem"],
                                   "branch": c["branch"], "lds": c["lds"], "vmem": c["vmem"],
                                   "half_rate_share_of_valu_cycles": round((c["valu_half"] + c["valu_lane"]) * c_half / cyc, 3) if cyc else 0}
        print("whole %-60s full %5d half %5d lane %4d salu %5d smem %3d branch %4d lds %3d vmem %3d" % (n2[:60], c["valu_full"], c["valu_half"], c["valu_lane"], c["salu"], c["smem"], c["branch"], c["lds"], c["vmem"]))
    if a.json:
        json.dump(report, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
