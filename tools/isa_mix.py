#!/usr/bin/env python3
"""Static instruction mix of the shipped k_icp launch shapes: which share of the vector instructions are float64 (4 issue
cycles per wave on gfx950's SIMD-32) and which are not (2 cycles).  bench.py prices SQ_INSTS_VALU with it ("issue_mix")
next to the all-f64 upper bound.  Runs hipcc -S on csrc/icp_kernels.hip (cross-compiles without a GPU).

usage: isa_mix.py <out.json>"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "a-2d-lidar-based-slam-system-for-wheeled-mobile-robots_amd", "csrc")


def main():
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "icp.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                               "-I" + CSRC, "-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, "icp_kernels.hip")], stderr=subprocess.DEVNULL)
        text = open(asm).read()
    out = {}
    for name, sym in (("qpt2", "_ZN4slam5k_icpIdLi2ELi4EEEvNS_7IcpArgsE"), ("qpt3", "_ZN4slam5k_icpIdLi3ELi4EEEvNS_7IcpArgsE")):
        m = re.search(r"^%s:.*?^\s*\.amdhsa_kernel %s" % (re.escape(sym), re.escape(sym)), text, re.S | re.M)
        body = m.group(0) if m else ""
        ops = re.findall(r"^\s+(v_\w+)([^\n]*)", body, re.M)
        if not ops:
            sys.exit("isa_mix.py: kernel %s not found in the ISA of icp_kernels.hip (template arguments changed?)" % sym)
        c = {"f64_arith": 0, "f64_cmp_minmax": 0, "f64_other": 0, "dpp_lane": 0, "cvt_mulint": 0, "trans_f32": 0, "cndmask": 0, "simple": 0}
        for op, rest in ops:
            if "dpp" in op or "quad_perm" in rest or "row_" in rest or re.search(r"permlane|readlane|readfirstlane|writelane", op):
                c["dpp_lane"] += 1
            elif re.search(r"v_(add|mul|fma|fmac)_f64", op):
                c["f64_arith"] += 1
            elif re.search(r"v_cmpx?_\w+_f64|v_(min|max)_f64", op):
                c["f64_cmp_minmax"] += 1
            elif "f64" in op:
                c["f64_other"] += 1
            elif re.search(r"v_cvt|v_mul_lo|v_mul_hi|v_mad_u64", op):
                c["cvt_mulint"] += 1
            elif re.search(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f32", op):
                c["trans_f32"] += 1
            elif op.startswith("v_cndmask"):
                c["cndmask"] += 1
            else:
                c["simple"] += 1
        out[name] = {"valu": len(ops), "f64": c["f64_arith"] + c["f64_cmp_minmax"] + c["f64_other"], "classes": c}
    tot = out["qpt2"]
    res = {"f64_share": tot["f64"] / float(max(tot["valu"], 1)), "static_valu_instructions": tot["valu"], "static_f64_instructions": tot["f64"],
           "static_classes": tot["classes"],
           "kernel": "k_icp<double, 2, 4> (both passes, every search path)", "qpt3": out["qpt3"],
           "note": "static count over the kernel's ISA (hipcc -S): f64_arith = v_add / mul / fma / fmac_f64 (what the hardware's SQ_INSTS_VALU_ADD / MUL / FMA_F64 counters count), "
                   "f64_cmp_minmax = v_cmp*_f64, v_min / max_f64 (float64 issue rate, no counter of their own), dpp_lane = DPP moves, permlane, readlane (quarter rate), "
                   "cvt_mulint = conversions and 32-bit integer multiplies (quarter rate), cndmask = selects (4.3 cycles in their VOP3 encoding), trans_f32 = v_rcp / sqrt / ..._f32; issue costs: profiles/r04_ubench_issue.txt"}
    json.dump(res, open(sys.argv[1], "w"), indent=1)
    print(res)


if __name__ == "__main__":
    main()
