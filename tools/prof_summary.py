#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output per (kernel, grid size): kernel-trace durations and PMC counters.

usage: prof_summary.py <rocprof output dir> [substring of kernel name]
Prints a markdown table; used to produce the files under profiles/.
"""
import collections
import csv
import glob
import sys


def main():
    root = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "hive"
    for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if filt not in name:
                continue
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            key = (name.split("(")[0][:60], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"),
                   r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"))
            agg[key].append(dur)
        print("| kernel | grid (threads) | wg | vgpr | lds | calls | avg us | min us | max us |")
        print("|---|---|---|---|---|---|---|---|---|")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            print(f"| {k[0]} | {k[1]} | {k[2]} | {k[3]} | {k[4]} | {len(v)} | {sum(v) / len(v):.2f} | {min(v):.2f} | {max(v):.2f} |")
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if filt not in name:
                continue
            key = (name.split("(")[0][:60], r.get("Grid_Size", "?"), r["Counter_Name"])
            agg[key].append(float(r["Counter_Value"]))
        print("| kernel | grid (threads) | counter | dispatches | mean per dispatch |")
        print("|---|---|---|---|---|")
        for k, v in sorted(agg.items()):
            print(f"| {k[0]} | {k[1]} | {k[2]} | {len(v)} | {sum(v) / len(v):.2f} |")


if __name__ == "__main__":
    main()
