#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of scripts/profile_round.sh into profiles/ (per-launch means of the dominant
comparison kernel; every counter group was collected in its own pass, as /opt/skills/guides/MI355X_MICROARCH.md
prescribes).

usage: scripts/pmc_summary.py <tag> gpurun_out/prof_<tag>
Writes   profiles/<tag>_pmc_summary.json         everything below, with the kernel name, the workload (bench line of
                                                 the stats pass) and the comparisons per launch actually profiled
         profiles/<tag>_kernel_stats.csv         rocprofv3 --stats summary of the un-perturbed pass
         profiles/<tag>_pmc_<pass>_sample.csv    first rows of each counter file (the raw evidence)
         profiles/<tag>_bench_under_rocprof.log  bench.py output of the stats pass
and, for the default (BASELINE config 2) workload, refreshes profiles/pmc_current.json, which bench.py reads for
the instruction and traffic figures of `roofline` / `hbm` (it refuses a summary whose kernel or shape differs).

Units and corrections: SQ_INSTS_* count wave-instructions; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves/SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (clock = value / 8 / duration);
FETCH_SIZE (KB) reports 1/2 of the bytes of 16-B/lane coalesced streams on gfx950 -> doubled; WRITE_SIZE (KB) is exact.
"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

VALU_PEAK = 1024 * 2.4e9 / 2   # wave-instructions/s: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles at 2.4 GHz
HBM_PEAK = 8.0e12


def dominant_kernel(rows):
    tot = {}
    for r in rows:
        k = r["Kernel_Name"]
        if "k_compare" in k:
            tot[k] = tot.get(k, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return max(tot, key=tot.get)


def read_pass(d):
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv")) + glob.glob(os.path.join(d, "*counter_collection.csv"))
    if not files:
        return None, None, None
    files.sort(key=os.path.getmtime)  # gpurun merges every call's files into gpurun_out/: take the latest run's
    rows = list(csv.DictReader(open(files[-1])))
    kname = dominant_kernel(rows)
    vals, durs = {}, []
    seen = set()
    for r in rows:
        if r["Kernel_Name"] != kname:
            continue
        vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    meta = {"kernel": kname, "launches": len(durs), "mean_launch_ms_under_pmc": statistics.mean(durs),
            "grid_size": int(rows[0]["Grid_Size"]) if rows else None}
    for r in rows:
        if r["Kernel_Name"] == kname:
            meta.update(grid_size=int(r["Grid_Size"]), workgroup=int(r["Workgroup_Size"]), lds_bytes=int(r["LDS_Block_Size"]),
                        scratch_bytes=int(r["Scratch_Size"]), vgpr_count_field=int(r["VGPR_Count"]),
                        sgpr_count_field=int(r["SGPR_Count"]))
            break
    return {k: statistics.mean(v) for k, v in vals.items()}, meta, files[0]


def main():
    tag, base = sys.argv[1:3]
    os.makedirs("profiles", exist_ok=True)
    out = {"tag": tag, "passes": {}}
    # un-perturbed stats pass: bench line + kernel stats
    log = os.path.join(base, "bench_under_rocprof.log")
    bench = None
    for ln in open(log):
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            bench = json.loads(ln)
    out["bench_under_rocprof"] = bench
    shutil.copy(log, "profiles/%s_bench_under_rocprof.log" % tag)
    st = glob.glob(os.path.join(base, "stats", "*", "*kernel_stats.csv")) + glob.glob(os.path.join(base, "stats", "*kernel_stats.csv"))
    kstats = None
    if st:
        st.sort(key=os.path.getmtime)
        st = st[-1:]
        shutil.copy(st[0], "profiles/%s_kernel_stats.csv" % tag)
        rows = list(csv.DictReader(open(st[0])))
        cmp_rows = [r for r in rows if "k_compare" in r["Name"]]
        if cmp_rows:
            r = max(cmp_rows, key=lambda r: float(r["TotalDurationNs"]))
            kstats = {"kernel": r["Name"], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                      "share_of_gpu_time_pct": float(r["Percentage"])}
    out["kernel_stats"] = kstats
    c = {}
    for name in ("sqa", "sqb", "tc", "fetch", "write"):
        vals, meta, f = read_pass(os.path.join(base, name))
        if vals is None:
            continue
        out["passes"][name] = {"counters_per_launch": vals, **meta}
        c.update(vals)
        with open(f) as fi, open("profiles/%s_pmc_%s_sample.csv" % (tag, name), "w") as fo:
            n = 0
            for ln in fi:
                if n == 0 or meta["kernel"] in ln:
                    fo.write(ln)
                    n += 1
                if n > 40:
                    break
    rl = (bench or {}).get("roofline", {})
    cpl = rl.get("comparisons_per_launch")
    ms = kstats["avg_ms"] if kstats else None
    out["comparisons_per_launch"] = cpl
    out["kernel"] = kstats["kernel"] if kstats else None
    out["config"] = (bench or {}).get("config")
    d = {}
    if cpl and ms:
        sec = ms / 1e3
        if "SQ_INSTS_VALU" in c:
            d["valu_wave_instr_per_comparison"] = c["SQ_INSTS_VALU"] / cpl
            d["valu_wave_instr_per_s"] = c["SQ_INSTS_VALU"] / sec
            d["valu_issue_frac_of_spec_peak"] = d["valu_wave_instr_per_s"] / VALU_PEAK
        for k in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"):
            if k in c:
                d[k.lower() + "_per_comparison"] = c[k] / cpl
        if "GRBM_GUI_ACTIVE" in c:
            ms_pmc = out["passes"]["sqa"]["mean_launch_ms_under_pmc"]
            d["clock_mhz_under_pmc"] = c["GRBM_GUI_ACTIVE"] / 8 / (ms_pmc / 1e3) / 1e6
            if "SQ_INSTS_VALU" in c:
                d["valu_issue_frac_at_sustained_clock"] = (c["SQ_INSTS_VALU"] / (ms_pmc / 1e3)) / (
                    1024 * d["clock_mhz_under_pmc"] * 1e6 / 2)
        if "SQ_WAVE_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c:
            d["active_inst_valu_over_wave_cycles"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
        if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
            d["wait_any_over_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_INST_ANY" in c:
            d["wait_inst_any_over_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "TCP_TCC_READ_REQ_sum" in c:
            d["l1_hit_rate_reads"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
        if "FETCH_SIZE" in c:
            d["fetch_bytes_per_launch_corrected_x2"] = c["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in c:
            d["write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            t = d["fetch_bytes_per_launch_corrected_x2"] + d["write_bytes_per_launch"]
            d["traffic_bytes_per_launch"] = t
            d["traffic_bytes_per_comparison"] = t / cpl
            d["traffic_GBps"] = t / sec / 1e9
            d["traffic_frac_of_hbm_peak"] = t / sec / HBM_PEAK
    out["derived"] = d
    # identity of the device sources these counters belong to (bench.py refuses them on any other tree)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bioem_amd.buildinfo import source_blobs
    out["source_blobs"] = source_blobs(out["kernel"])   # the kernel's own translation unit (kernels_*.hip + headers)
    out["notes"] = ("per-launch means over the launches of the dominant comparison kernel; durations for the rates are the "
                    "un-perturbed kernel-trace pass (kernel_stats.avg_ms); VGPR_Count field of rocprofv3 is in allocation "
                    "granules as reported, see the code object for the exact register count")
    json.dump(out, open("profiles/%s_pmc_summary.json" % tag, "w"), indent=1)
    flags = open(os.path.join(base, "bench_flags.txt")).read().strip() if os.path.exists(os.path.join(base, "bench_flags.txt")) else ""
    if not flags:
        json.dump(out, open("profiles/pmc_current.json", "w"), indent=1)
    print(json.dumps({"kernel": out["kernel"], "cpl": cpl, "avg_ms": ms, **d}, indent=1))


if __name__ == "__main__":
    main()
