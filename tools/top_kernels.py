#!/usr/bin/env python3
"""Top kernels by total time from a rocprofv3 kernel_trace CSV (names truncated)."""
import collections, csv, glob, sys
root = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    t0, t1 = None, None
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        k = r["Kernel_Name"][:70]
        agg[k][0] += 1
        agg[k][1] += (e - s) / 1e3
        t0 = s if t0 is None else min(t0, s)
        t1 = e if t1 is None else max(t1, e)
    tot = sum(v[1] for v in agg.values())
    print(f"total kernel time {tot/1e3:.1f} ms over span {(t1-t0)/1e6:.1f} ms")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
        print(f"| {k} | {v[0]} | {v[1]/1e3:.2f} | {v[1]/v[0]:.1f} | {100*v[1]/tot:.1f} |")
