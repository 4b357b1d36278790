#!/usr/bin/env python3
"""Aggregates a rocprofv3 --pmc counter_collection CSV by kernel name (and optionally per dispatch).
usage: pmc_summary.py <counter_collection.csv> [--per-dispatch substring]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    per = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--per-dispatch" else None
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    rows = defaultdict(lambda: defaultdict(float))
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0][:70]
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[name].add(r["Dispatch_Id"])
            if per and per in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
                rows[int(r["Dispatch_Id"])]["_grid"] = float(r.get("Grid_Size", 0) or 0)
    for name, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
        n = len(calls[name])
        print("%s  calls=%d" % (name, n))
        for c, v in sorted(cs.items()):
            print("    %-28s total=%.4g  per_call=%.4g" % (c, v, v / n))
    if per:
        for d, cs in sorted(rows.items()):
            print(d, {k: round(v, 1) for k, v in cs.items()})


if __name__ == "__main__":
    main()
