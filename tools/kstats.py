#!/usr/bin/env python3
"""Prints the top rows of a rocprofv3 kernel_stats CSV. usage: kstats.py file.csv [n]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.2f over %d kernels, %d launches" % (tot / 1e6, len(rows), sum(int(r["Calls"]) for r in rows)))
for r in rows[:n]:
    print("%-86s calls=%6s tot=%8.2fms avg=%8.1fus %5.1f%%" % (r["Name"][:86], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
