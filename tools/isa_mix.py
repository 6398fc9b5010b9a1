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
        valu = re.findall(r"^\s+(v_\w+)", body, re.M)
        f64 = [i for i in valu if re.search(r"_f64|f64_", i)]
        if not valu:
            sys.exit("isa_mix.py: kernel %s not found in the ISA of icp_kernels.hip (template arguments changed?)" % sym)
        out[name] = {"valu": len(valu), "f64": len(f64)}
    tot = out["qpt2"]
    res = {"f64_share": tot["f64"] / float(max(tot["valu"], 1)), "static_valu_instructions": tot["valu"], "static_f64_instructions": tot["f64"],
           "kernel": "k_icp<double, 2, 4> (both passes, every search path)", "qpt3": out["qpt3"],
           "note": "static count over the kernel's ISA (hipcc -S); f64 = mnemonics with an f64 operand type (4 issue cycles), the rest 2"}
    json.dump(res, open(sys.argv[1], "w"), indent=1)
    print(res)


if __name__ == "__main__":
    main()
