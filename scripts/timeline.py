#!/usr/bin/env python3
"""Timeline of one bench workload from a rocprofv3 kernel trace (scripts/kstats.sh TAG ... leaves it under
gpurun_out/kstats_TAG/): for the LAST pass of the run, busy time of the comparison stream and of the preparation
stream, their overlap, the idle gaps and the kernels in launch order.
usage: scripts/timeline.py gpurun_out/kstats_TAG [passes]"""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# passes are separated by the fold's successor gap: take the last pass = kernels after the last k_init / D2H gap of > 200 us
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
gaps = [i for i in range(1, len(ks)) if ks[i][0] - max(k[1] for k in ks[:i][-8:]) > 150000]
start = gaps[-1] if gaps else 0
sel = ks[start:]
t0 = sel[0][0]
span = max(k[1] for k in sel) - t0


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


cmp_iv = [(s, e) for s, e, n, q in sel if "k_compare" in n or "k_fold" in n or "k_nyquist" in n or "k_posterior" in n]
prep_iv = [(s, e) for s, e, n, q in sel if (s, e) not in cmp_iv]
print("last pass: %d kernels, span %.1f us; comparison stream busy %.1f us, preparation busy %.1f us, any busy %.1f us"
      % (len(sel), span / 1e3, union(cmp_iv) / 1e3, union(prep_iv) / 1e3, union(cmp_iv + prep_iv) / 1e3))
for s, e, n, q in sel[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print("%9.1f %9.1f  %7.1f us  q%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short(n)))
