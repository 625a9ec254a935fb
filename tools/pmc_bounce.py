#!/usr/bin/env python3
"""Per-dispatch view of one tools/pmc.sh pass: the dispatches of a kernel in order (a batch enqueues its bounces back
to back, so dispatch i of k_traverse4 is bounce i % bounces of batch i // bounces).
usage: pmc_bounce.py <dir> <pass, e.g. rdreq|tcc> <kernel-substring> [bounces=8]"""
import collections, csv, glob, sys
d, name, want = sys.argv[1], sys.argv[2], sys.argv[3]
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 8
f = (glob.glob(f"{d}/{name}/*/*_counter_collection.csv") + glob.glob(f"{d}/{name}/*_counter_collection.csv"))[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if want in r["Kernel_Name"]:
        e = disp.setdefault(int(r["Dispatch_Id"]), {"us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = [disp[k] for k in sorted(disp)]
keys = [k for k in rows[0] if k != "us"]
print("dispatches", len(rows))
for i, e in enumerate(rows):
    extra = ""
    if "TCC_EA0_RDREQ_sum" in e:
        r128, r64, r32 = e.get("TCC_EA0_RDREQ_128B_sum", 0), e.get("TCC_EA0_RDREQ_64B_sum", 0), e.get("TCC_EA0_RDREQ_32B_sum", 0)
        other = max(e["TCC_EA0_RDREQ_sum"] - r128 - r64 - r32, 0)
        extra = " read_MB %.1f" % ((r128 * 128 + r64 * 64 + r32 * 32 + other * 64) / 1e6)
    if "TCC_REQ_sum" in e:
        extra = " hit %.3f" % (e["TCC_HIT_sum"] / e["TCC_REQ_sum"])
    print("batch %d bounce %d  %8.1f us " % (i // nb, i % nb, e["us"]) + " ".join("%s=%.0f" % (k.replace("_sum", ""), e[k]) for k in keys) + extra)
