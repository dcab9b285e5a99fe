#!/usr/bin/env python3
"""Per-kernel durations and the gaps between consecutive dispatches from a rocprofv3 --kernel-trace csv.
usage: gap_from_trace.py <kernel_trace.csv> [n_last_dispatches]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rows = rows[-n:]
def short(nm):
    m = re.search(r"k_\w+", nm)
    s = m.group(0) if m else nm[:30]
    t = re.search(r"<([^>]*)>", nm)
    return s + ("<" + t.group(1)[:40] + ">" if t else "")
dur = collections.defaultdict(list); gap_before = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    k = short(b["Kernel_Name"])
    dur[k].append((int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1000.0)
    gap_before[k].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1000.0)
print("%-70s %6s %9s %9s %9s" % ("kernel", "n", "dur_us", "gap_pre", "dur+gap"))
import statistics as st
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d, g = st.median(dur[k]), st.median(gap_before[k])
    print("%-70s %6d %9.2f %9.2f %9.2f" % (k, len(dur[k]), d, g, d + g))
