#!/usr/bin/env python3
"""profiles/make_hot_loops.py ROUND -- writes profiles/ROUND/hot_loops.json (read by bench.py) and the kernel resource table
profiles/ROUND/kernel_resources.csv from the ISA of the current sources (hipcc --save-temps, no GPU needed).

filter_kernel: the static mix of its dominant loop, the two-halves container loop of the 3-word kernel (the innermost loop
with the most v_lshrrev_b64).  exact_kernel: no single loop dominates, so the static mix of the whole short-read kernel
body and of the device functions it calls is used."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
out = os.path.join(ROOT, "profiles", rnd)
os.makedirs(out, exist_ok=True)
isa_dir = "/tmp/trew_isa_%s" % rnd
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "--isa", isa_dir, "--csv", os.path.join(out, "kernel_resources.csv")],
               check=True, stdout=subprocess.DEVNULL)
isa = os.path.join(isa_dir, "trew_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
tmp = os.path.join(isa_dir, "loops.json")
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_loops.py"), isa, "filter_kernelILi3E", "--json", tmp], check=True, stdout=subprocess.DEVNULL)
f = json.load(open(tmp))
loop = max(f["loops"], key=lambda l: l["ops"].get("v_lshrrev_b64", 0))
res = {"filter_kernel": {
    "half_rate_share_of_valu_insts": round(loop["valu_half_rate"] / loop["valu"], 3),
    "from": "innermost loop %s of filter_kernel<3> (both halves of a read, two k per trip; an instruction with a scalar source counts as half rate): %d VALU = %d full-rate + %d half-rate, %d SALU, %d SMEM, %d branches" % (
        loop["label"], loop["valu"], loop["valu_full_rate"], loop["valu_half_rate"], loop["salu"], loop["smem"], loop["branch"]),
    "loop": loop}}
fns = ["eval_runsImEE", "eval_kImEE", "emit_kImEE", "stage_basesE", "9table_addENS", "eval_k_windowsImEE"]
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_loops.py"), isa, "exact_kernelILi3ELi0EmEE", "--min-bcnt", "1000", "--json", tmp, "--functions"] + fns,
               check=True, stdout=subprocess.DEVNULL)
e = json.load(open(tmp))
full = sum(v["valu_full_rate"] for v in e["functions"].values())
half = sum(v["valu_half_rate"] + v["valu_lane_ops"] for v in e["functions"].values())
res["exact_kernel"] = {
    "half_rate_share_of_valu_insts": round(half / (full + half), 3),
    "from": "static mix of exact_kernel<3, short, u64> and the device functions it calls (%s): %d full-rate + %d half-rate VALU instructions (v_readlane / v_writelane counted half rate)" % (
        ", ".join(sorted(k[:28] for k in e["functions"])), full, half),
    "functions": e["functions"]}
json.dump(res, open(os.path.join(out, "hot_loops.json"), "w"), indent=1)
print(json.dumps({k: v["half_rate_share_of_valu_insts"] for k, v in res.items()}))
