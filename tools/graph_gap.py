#!/usr/bin/env python3
"""Gap between the last kernel of a decode step (k_dec_lmhead_tail) and the first of the next one, against the gaps inside a
step, from a rocprofv3 --kernel-trace csv:  graph_gap.py <kernel_trace.csv>"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
inter, intra = [], []
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1000.0
    if "k_dec_lmhead_tail" in a["Kernel_Name"] and "k_dec_qkv_attn" in b["Kernel_Name"]: inter.append(g)
    elif "k_dec_" in a["Kernel_Name"] and "k_dec_" in b["Kernel_Name"] and g < 50: intra.append(g)
print("between steps (tail -> next qkv+attn): n %d median %.2f us min %.2f max %.2f" % (len(inter), st.median(inter), min(inter), max(inter)))
print("inside a step: n %d median %.2f us p90 %.2f" % (len(intra), st.median(intra), sorted(intra)[int(len(intra) * 0.9)]))
