#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc passes of tools/pmc.sh into pmc_traffic.json and pmc_sq.json (what bench.py reads
from profiles/): per-launch means of the dominant kernel, the gfx950 read-size correction (MI355X_MICROARCH.md, HBM
section: FETCH_SIZE tallies 128-byte requests at 64 B) and the derived VALU figures.
usage: pmc_json.py <dir> <kernel-substring> <frames_per_launch>"""
import collections, csv, glob, json, os, sys

d, want, fpl = sys.argv[1], sys.argv[2], int(sys.argv[3])
means, durs, ndisp = {}, {}, {}
for f in sorted(glob.glob(d + "/*/*/*_counter_collection.csv") + glob.glob(d + "/*/*_counter_collection.csv")):
    name = os.path.basename(f).split("_counter_collection")[0]
    acc = collections.defaultdict(float); dur = {}
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if not dur:
        continue
    n = len(dur)
    for k, v in acc.items():
        means[k] = v / n
    durs[name] = sum(dur.values()) / n / 1e3
    ndisp[name] = n
rays = None
for lg in glob.glob(d + "/*.log"):
    for line in open(lg):
        if line.startswith('{"scene"'):
            rays = json.loads(line)
launches = max(ndisp.values()) if ndisp else 0
rays_per_launch = rays["rays_total"] / launches if rays and launches else None
src = "tools/pmc.sh: separate rocprofv3 --pmc passes over tools/run_frames.py %s %s %d (kernels serialised by the profiler)" % (
    rays["scene"] if rays else "?", rays["frames"] if rays else "?", fpl)
if "FETCH_SIZE" in means:
    r128, r64, r32 = means.get("TCC_EA0_RDREQ_128B_sum", 0.0), means.get("TCC_EA0_RDREQ_64B_sum", 0.0), means.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    rd_all = means.get("TCC_EA0_RDREQ_sum", 0.0)
    other = max(rd_all - r128 - r64 - r32, 0.0)
    read_bytes = r128 * 128 + r64 * 64 + r32 * 32 + other * 64 if rd_all else 2 * means["FETCH_SIZE"] * 1024
    write_bytes = means.get("WRITE_SIZE", 0.0) * 1024
    json.dump({"kernel": want, "frames_per_launch": fpl, "source": src, "launches_sampled": launches,
               "FETCH_SIZE_KB_per_launch": means["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": means.get("WRITE_SIZE"),
               "TCC_EA0_RDREQ_per_launch": rd_all, "TCC_EA0_RDREQ_128B_per_launch": r128, "TCC_EA0_RDREQ_64B_per_launch": r64,
               "TCC_EA0_RDREQ_32B_per_launch": r32,
               "correction": "gfx950: read bytes = sum of read requests by size (128/64/32 B); 2 x FETCH_SIZE = %.1f MB agrees when "
                             "nearly all requests are 128-byte; WRITE_SIZE as is; Infinity-Cache hits are included in these "
                             "fabric-side counters (upper bound of HBM traffic)" % (2 * means["FETCH_SIZE"] * 1024 / 1e6),
               "read_bytes_per_launch": read_bytes, "write_bytes_per_launch": write_bytes,
               "trace_kernel_hbm_bytes_per_launch": round(read_bytes + write_bytes),
               "mean_launch_us_serialised_by_the_profiler": durs.get("fetch"), "rays_per_launch": rays_per_launch},
              open(d + "/pmc_traffic.json", "w"), indent=2)
    print("traffic bytes/launch", round(read_bytes + write_bytes), "rays/launch", rays_per_launch)
if "SQ_INSTS_VALU" in means:
    dur_us = durs.get("sq1")
    clk = means["GRBM_GUI_ACTIVE"] / 8.0 / (dur_us * 1e-6) if "GRBM_GUI_ACTIVE" in means and dur_us else 2.2e9
    avail_quad = dur_us * 1e-6 * clk * 1024 / 4.0   # 256 CUs x 4 SIMDs, counters in quad-cycles
    out = {"kernel": want, "frames_per_launch": fpl, "source": src, "launches_sampled": launches,
           "mean_launch_us_serialised_by_the_profiler": dur_us, "shader_clock_ghz": round(clk / 1e9, 3),
           # a wave64 VALU instruction is "active" for one quad-cycle; a SIMD (32 lanes per cycle) can run two
           # wavefronts' instructions overlapped, i.e. issue one every 2 cycles (MI355X_MICROARCH.md): busy > 1 is possible
           "valu_pipe_frac": round(means["SQ_INSTS_VALU"] * 2.0 / (dur_us * 1e-6 * clk * 1024), 4),
           "lanes_per_valu_inst": round(means["SQ_THREAD_CYCLES_VALU"] / means["SQ_INSTS_VALU"], 2),
           "valu_inst_per_ray": round(means["SQ_INSTS_VALU"] / rays_per_launch, 2) if rays_per_launch else None,
           "vmem_inst_per_ray": round((means.get("SQ_INSTS_VMEM_RD", 0) + means.get("SQ_INSTS_VMEM_WR", 0)) / rays_per_launch, 3) if rays_per_launch else None,
           "salu_inst_per_ray": round(means.get("SQ_INSTS_SALU", 0) / rays_per_launch, 2) if rays_per_launch else None,
           "lds_inst_per_ray": round(means.get("SQ_INSTS_LDS", 0) / rays_per_launch, 3) if rays_per_launch else None,
           "wave_occupancy_frac": round(means["SQ_WAVE_CYCLES"] / (avail_quad * 8), 4) if "SQ_WAVE_CYCLES" in means else None,
           "l2_hit_frac": round(means["TCC_HIT_sum"] / means["TCC_REQ_sum"], 4) if "TCC_REQ_sum" in means else None,
           "raw_means_per_launch": {k: v for k, v in sorted(means.items())}}
    json.dump(out, open(d + "/pmc_sq.json", "w"), indent=2)
    print({k: v for k, v in out.items() if k not in ("raw_means_per_launch", "source")})
