#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs:  tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else "k_trace"
for f in sorted(glob.glob(d + "/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    n = len(dur)
    if not n: continue
    print(f"# {f}: {n} dispatches of *{want}*, mean duration {sum(dur.values())/n/1e3:.1f} us, VGPR {r['VGPR_Count']}")
    for k, v in sorted(acc.items()):
        print(f"  {k:<40} mean/dispatch {sum(v)/len(v):>18.1f}   total {sum(v):>18.0f}")
