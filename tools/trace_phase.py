#!/usr/bin/env python3
"""Kernel sequence of ONE witness phase from a rocprofv3 kernel trace of tools/phase_times.py (run_prof_rank.sh):
python tools/trace_phase.py <kernel_trace.csv> [k]  -- the k-th (default 2: the third, blocking) launch of the group / G1
accumulation, from the histogram of its schedule to the last copy of its results, with start (ms) and duration (ms)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
names = [r["Kernel_Name"] for r in rows]
acc = [i for i, n in enumerate(names) if "segment_accumulate_group" in n]
at = acc[k]
i0 = at
while "radix_hist" not in names[i0]:
    i0 -= 1
i1 = at + 1
while i1 < len(rows) and "radix_hist" not in names[i1] and "h_final" not in names[i1]:
    i1 += 1
t0 = int(rows[i0]["Start_Timestamp"])
tot = {}
for r in rows[i0:i1]:
    n = r["Kernel_Name"].replace("ug::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:56]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot[n] = tot.get(n, 0.0) + d
    print("%9.3f %8.3f  grid %-10s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, d, r["Grid_Size_X"], n))
print("span %.3f ms, kernels %.3f ms" % ((int(rows[i1 - 1]["End_Timestamp"]) - t0) / 1e6, sum(tot.values())))
for n, d in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("  %8.3f  %s" % (d, n))
