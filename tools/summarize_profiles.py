#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_gpu.sh into the files kept under profiles/:

  <tag>_kernel_stats_default.csv   --kernel-trace --stats of `python3 bench.py` (4 lanes)
  <tag>_kernel_stats_1lane.csv     the same with --lanes 1 (kernels back to back)
  <tag>_pmc_summary.txt            per-kernel, per-launch averages of every --pmc pass
  <tag>_bench*.json                the bench lines of those runs
  pmc_traffic.json                 HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB
                                   (gfx950: FETCH_SIZE counts wide coalesced reads at half
                                   their bytes, MI355X_MICROARCH.md HBM section) and the VALU
                                   busy share, read by bench.py for roofline.traffic

usage: summarize_profiles.py <gpurun_out/prof_tag_config> <tag> [config]

pmc_traffic.json is keyed by bench.py --config ({"replay": {kernel: ...}, "particles": ...});
the entry of the given config is replaced, the others are kept."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def short(name):
    n = name.replace("void ", "")
    n = n.split("(")[0]
    n = n.replace("slam::", "")
    base = n.split("<")[0]
    return base


def newest(pattern):
    """Files of the LATEST run only: gpurun_out/ accumulates the output of earlier calls (rocprofv3 names its files
    by process id), and kernels renamed since would be averaged into today's."""
    files = glob.glob(pattern, recursive=True)
    if not files:
        return []
    latest = max(files, key=os.path.getmtime)
    pid = os.path.basename(latest).split("_")[0]
    return [f for f in files if os.path.basename(f).split("_")[0] == pid and os.path.dirname(f) == os.path.dirname(latest)]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    cfg = sys.argv[3] if len(sys.argv) > 3 else "replay"
    tag = tag if cfg == "replay" else "%s_%s" % (tag, cfg)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(src, "profiles")
    os.makedirs(dst, exist_ok=True)
    for which in ("default", "1lane"):
        f = newest(os.path.join(src, "trace_" + which, "**", "*kernel_stats.csv"))
        if f:
            shutil.copy(f[0], os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, which)))
        b = os.path.join(src, "bench_%s_under_rocprof.json" % which)
        if os.path.exists(b):
            shutil.copy(b, os.path.join(dst, "%s_bench_%s_under_rocprof.json" % (tag, which)))
    if os.path.exists(os.path.join(src, "bench.json")):
        shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "%s_bench.json" % tag))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    lines = ["# rocprofv3 --pmc passes (separate runs, --pmc only) of: python3 bench.py --config %s --no-cpu-baseline --lanes 1 (few steps%s)" % (cfg, "; the later half of every kernel's launches: maps that have seen a few scans" if cfg == "particles" else ""),
             "# per-launch averages; FETCH_SIZE / WRITE_SIZE in KB as reported (gfx950: FETCH_SIZE counts wide coalesced reads at half their bytes)", ""]
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        grp = open(os.path.join(d, "counters.txt")).read().strip() if os.path.exists(os.path.join(d, "counters.txt")) else d
        lines.append("## --pmc " + grp)
        local = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in newest(os.path.join(d, "**", "*counter_collection.csv")):
            rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id") or 0))
            for row in rows:
                k = short(row["Kernel_Name"])
                if not k.startswith("k_"):
                    continue
                local[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if cfg == "particles":
            # the particle maps persist across steps: a map's first scans find every cell fresh (pmap written everywhere);
            # the benchmark's hundreds of steps run in the settled state, which the LATER HALF of a pass's launches show
            for k in local:
                for c in local[k]:
                    v = local[k][c]
                    local[k][c] = v[len(v) // 2:]
        for k in sorted(local):
            lines.append("%s %s" % (k, {c: "%.4g (n=%d)" % (sum(v) / len(v), len(v)) for c, v in sorted(local[k].items())}))
            for c, v in local[k].items():
                if "icp_qpt=3" in grp:                # the extra pass in the overlapped run's launch shape: kept apart
                    per[k][c + "@qpt3"] = v
                else:
                    per[k][c] = v
        lines.append("")
    traffic = {}
    lines.append("# derived per launch -> profiles/pmc_traffic.json")
    for k, c in sorted(per.items()):
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        avg = lambda name: sum(c[name]) / len(c[name]) if c.get(name) else None
        fetch, write = avg("FETCH_SIZE"), avg("WRITE_SIZE")
        hbm = (2.0 * fetch + write) * 1024.0
        busy = None
        if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_BUSY_CYCLES"):
            # SQ_ACTIVE_INST_VALU counts issue slots of 4 cycles (it tracks SQ_INSTS_VALU to 2 %),
            # summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the
            # kernel had (GRBM/8) x 1024 SIMD-cycles: busy = 4 A / ((G / 8) x 1024) = A / (32 G)
            busy = avg("SQ_ACTIVE_INST_VALU") / (32.0 * avg("GRBM_GUI_ACTIVE")) if c.get("GRBM_GUI_ACTIVE") else None
        traffic[k] = {"hbm_bytes_per_launch": hbm, "fetch_kb_raw": fetch, "write_kb_raw": write,
                      "source": "profiles/%s_pmc_summary.txt: (2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 FETCH_SIZE x2 correction (MI355X_MICROARCH.md HBM section)" % tag,
                      "valu_busy_frac": busy, "valu_insts_per_launch": avg("SQ_INSTS_VALU"),
                      "lds_insts_per_launch": avg("SQ_INSTS_LDS")}
        if c.get("SQ_INSTS_VALU@qpt3"):
            traffic[k]["valu_insts_per_launch_qpt3"] = avg("SQ_INSTS_VALU@qpt3")
        if c.get("SQ_INSTS_VALU_ADD_F64") is not None and c.get("SQ_INSTS_VALU_FMA_F64") is not None:
            # the hardware's own instruction classes (per launch): float64 add / mul / fma, float64 and float32
            # transcendentals, conversions; SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = lanes active per instruction
            traffic[k]["issue_mix_hw"] = {"f64_arith": (avg("SQ_INSTS_VALU_ADD_F64") or 0.0) + (avg("SQ_INSTS_VALU_MUL_F64") or 0.0) + (avg("SQ_INSTS_VALU_FMA_F64") or 0.0),
                                          "f64_trans": avg("SQ_INSTS_VALU_TRANS_F64") or 0.0, "cvt": avg("SQ_INSTS_VALU_CVT") or 0.0,
                                          "trans_f32": avg("SQ_INSTS_VALU_TRANS_F32") or 0.0, "int32": avg("SQ_INSTS_VALU_INT32") or 0.0,
                                          "active_lanes_per_inst": (avg("SQ_THREAD_CYCLES_VALU") or 0.0) / max(avg("SQ_INSTS_VALU") or 1.0, 1.0),
                                          "source": "profiles/%s_pmc_summary.txt: SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64, _CVT, _TRANS_F32, _INT32 per launch" % tag}
        lds_c, lds_a = avg("SQ_LDS_BANK_CONFLICT"), avg("SQ_LDS_IDX_ACTIVE")
        if lds_a:
            traffic[k]["lds_conflict_cycle_share"] = lds_c / lds_a
        lines.append("%s HBM %.3f MB VALU busy %s" % (k, hbm / 1e6, "n/a" if busy is None else "%.1f %%" % (100 * busy)))
    open(os.path.join(dst, "%s_pmc_summary.txt" % tag), "w").write("\n".join(lines) + "\n")
    allcfg = {}
    try:
        allcfg = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
        if "k_icp" in allcfg:                     # round-1 layout (replay only, flat)
            allcfg = {"replay": allcfg}
    except Exception:
        allcfg = {}
    # the kernel sources these figures belong to (bench.py marks a roofline built on figures of other sources stale_pmc)
    sys.path.insert(0, root)
    try:
        import bench
        traffic["_source_hash"] = bench.source_hash()
    except Exception as e:
        traffic["_source_hash"] = "unknown (%s)" % e
    mix = os.path.join(src, "isa_mix.json")
    if os.path.exists(mix) and "k_icp" in traffic:
        traffic["k_icp"]["issue_mix"] = json.load(open(mix))
    allcfg[cfg] = traffic
    json.dump(allcfg, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    if os.path.isdir(os.path.join(root, "profiles")) and os.access(os.path.join(root, "profiles"), os.W_OK):
        for f in os.listdir(dst):
            shutil.copy(os.path.join(dst, f), os.path.join(root, "profiles", f))
    print("\n".join(lines[-8:]))


if __name__ == "__main__":
    main()
