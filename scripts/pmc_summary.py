#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) for the comparison kernel into profiles/.

usage: scripts/pmc_summary.py <round-tag> <fetch_dir> <write_dir>
Writes profiles/<tag>_pmc_summary.json and refreshes profiles/pmc_traffic.json (read by bench.py for the
`roofline.traffic` field).  gfx950 correction: FETCH_SIZE reports 1/2 of the bytes of 16-B/lane coalesced
streams -> doubled; WRITE_SIZE is exact for 16-B streaming stores."""
import csv
import glob
import json
import statistics
import sys

tag, fdir, wdir = sys.argv[1:4]
out = {}
for name, d in (("fetch", fdir), ("write", wdir)):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    vals, durs, kname = [], [], None
    for r in csv.DictReader(open(f)):
        if "k_compare" in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
            durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            kname = "k_compare_fast" if "k_compare_fast" in r["Kernel_Name"] else "k_compare_generic"
    out[name] = {"launches": len(vals), "mean_counter_KB": statistics.mean(vals),
                 "mean_launch_ms": statistics.mean(durs), "kernel": kname}
fetch_b = out["fetch"]["mean_counter_KB"] * 1024 * 2
write_b = out["write"]["mean_counter_KB"] * 1024
summary = {"tag": tag, "raw": out, "fetch_bytes_per_launch_corrected_x2": fetch_b,
           "write_bytes_per_launch": write_b, "traffic_bytes_per_launch": fetch_b + write_b,
           "workload": "224^2, 1000 particles, 64 orientations x 5 CTF per launch (320000 comparisons)",
           "comparisons_per_launch": 320000,
           "note": "FETCH_SIZE doubled (gfx950 16-B/lane stream correction); Infinity-Cache hits may be counted."}
json.dump(summary, open("profiles/%s_pmc_summary.json" % tag, "w"), indent=1)
json.dump(summary, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
