#!/usr/bin/env python3
"""Per-kernel SQ counter shares from a rocprofv3 --pmc counter_collection.csv:  sq_summary.py <csv> [kernel substring]"""
import collections, csv, sys
rows = csv.DictReader(open(sys.argv[1]))
want = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    if want in n:
        agg[(n[:48], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    d = {c: sum(x) / len(x) for c, x in v.items()}
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    print(k[0], "grid", k[1], "launches", len(v.get("SQ_WAVE_CYCLES", [])), "wave quad-cycles %.0f |" % wc,
          " ".join("%s %.0f%%" % (c.replace("SQ_", "").lower(), 100 * d[c] / wc) for c in sorted(d) if c != "SQ_WAVE_CYCLES"))
