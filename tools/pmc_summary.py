#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc FETCH_SIZE counter CSV (bench.py run) into profiles/:
per decode kernel the average FETCH_SIZE per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE is in KiB and counts 64 B per 128-B request of a wide streaming read: bytes = value * 1024 * 2;
calibrated here on the F16 lm_head kernel whose byte count is known exactly).

    cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -o pmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline
    python tools/pmc_summary.py OUT/pmc_counter_collection.csv profiles/r02_pmc_traffic.json
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inferflow_amd.build import source_hash  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(src)):
    n = r["Kernel_Name"]
    if "k_dec" in n and r["Counter_Name"] == "FETCH_SIZE":
        agg[n.replace("void ", "").split("(")[0]].append(float(r["Counter_Value"]))
out = {"source_hash": source_hash(), "counter": "FETCH_SIZE (KiB, rocprofv3 --pmc, own pass)", "correction": "bytes = KiB * 1024 * 2 (gfx950: 128-B requests tallied at 64 B)",
       "kernels": {}}
for k, v in sorted(agg.items()):
    out["kernels"][k] = {"launches": len(v), "fetch_kib_avg": sum(v) / len(v), "hbm_bytes_per_launch": sum(v) / len(v) * 1024 * 2}
json.dump(out, open(dst, "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-70s %6d launches  %12.0f B/launch" % (k[:70], v["launches"], v["hbm_bytes_per_launch"]))
