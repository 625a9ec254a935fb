#!/usr/bin/env python3
"""tools/ab.py -- one parametrised same-box A/B runner (replaces ab.sh, sweep.sh and round 4's 21 r4_*.sh one-offs).

    tools/ab.py <outdir> [--reps N] [--libs default,w_x,... | all] [--sweep name=v1,v2,...]... [--case "label: bench args"]...

Every combination of library x sweep value x case is one `bench.py` run (alternating, so that box drift hits all alike), with
bench.py's parity leg ON (--cpu-frames 1: a variant that corrupts its results prints PARITY FAILED instead of a number) and
--no-extras unless a case asks otherwise; logs under gpurun_out/<outdir>/, one summary line per run on stdout.

  --libs    default = cuda-path-tracer_amd/libptcore.so; w_<tag> = libptcore_w_<tag>.so (make -C cuda-path-tracer_amd/csrc
            variant TAG=<tag> EXTRA="-D..."); all = the default and every libptcore_w_*.so
  --sweep   a run-time parameter (ptc_set_param through bench.py --param); several --sweep: the cartesian product
  --case    named bench.py argument lists; built-ins: s20 (--steps 20 --warmup 5: the driver's command), def (default run),
            sh8 / sh4 / sh2 (--share-of N --steps 20 --warmup 5), c2 (--config 2), c5 (--config 5),
            lat (--steps 20 --warmup 5 with extras: latency and steady state)
Examples (round 4's scripts as calls of this one):
  r4_ab.sh        tools/ab.py r4ab --reps 2 --libs all --case s20 --case sh8 --case def --case lat
  r4_sched.sh     tools/ab.py r4sched --reps 2 --case "1x20: --steps 20 --warmup 5 --streams 1 --batch-frames 20" --case "2x10: ... --streams 2 --batch-frames 10"
  r4_feedparams   tools/ab.py r4feed --sweep refill_lanes=20,32 --sweep static_eighths=3,4 --case s20 --case c2 --case sh8 --case c5
  sweep.sh        tools/ab.py sw --libs w_x --sweep split_idle=4,8,16 --case s20
"""
import argparse
import glob
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILTIN = {"s20": "--steps 20 --warmup 5", "def": "", "sh8": "--share-of 8 --steps 20 --warmup 5", "sh4": "--share-of 4 --steps 20 --warmup 5",
           "sh2": "--share-of 2 --steps 20 --warmup 5", "c2": "--config 2", "c5": "--config 5 --steps 64 --warmup 16",
           "lat": "--steps 20 --warmup 5 --extras"}


def summary(d):
    p = d.get("parity")
    if p is not None and not (p.get("bit_exact") and p.get("live_equal", True) and p.get("rays_equal", True)):
        return "PARITY FAILED -- number void: %s" % {k: p.get(k) for k in ("bit_exact", "live_equal", "rays_equal", "mse")}
    r = d.get("roofline") or {}
    out = "%s %10.1f %s  %.4f ms/step" % ("parity ok " if p else "parity n/a", d["value"], d["unit"], d["ms_per_step"])
    if "avg_launch_us" in r:
        out += "  launch %8.1f us  frac %.4f" % (r["avg_launch_us"], r.get("frac") or 0)
    if r.get("frame_level_frac") is not None:
        out += "  frame-level %.3f" % r["frame_level_frac"]
    if r.get("per_bounce"):
        out += "  trace_ms " + " ".join("b%d:%.2f" % (b["bounce"], b["trace_ms"]) for b in r["per_bounce"])
    if d.get("latency"):
        out += "  latency %s" % d["latency"]
    if d.get("steady_state"):
        out += "  steady %.1f" % d["steady_state"]["value"]
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("outdir")
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--libs", default="default")
    ap.add_argument("--sweep", action="append", default=[])
    ap.add_argument("--case", action="append", default=[])
    ap.add_argument("--timeout", type=int, default=300)
    a = ap.parse_args()
    out = os.path.join(ROOT, "gpurun_out", a.outdir)
    os.makedirs(out, exist_ok=True)
    pkg = os.path.join(ROOT, "cuda-path-tracer_amd")
    if a.libs == "all":
        libs = [("default", os.path.join(pkg, "libptcore.so"))] + [(os.path.basename(f)[len("libptcore_"):-3], f)
                                                                   for f in sorted(glob.glob(os.path.join(pkg, "libptcore_w_*.so")))]
    else:
        libs = [(t, os.path.join(pkg, "libptcore.so" if t == "default" else "libptcore_%s.so" % t)) for t in a.libs.split(",")]
    sweeps = []
    for s in a.sweep:
        name, values = s.split("=")
        sweeps.append([(name, v) for v in values.split(",")])
    cases = []
    for c in a.case or ["s20"]:
        label, _, args = c.partition(":")
        cases.append((label.strip(), (args if _ else BUILTIN[label.strip()]).split()))
    for rep in range(a.reps):
        for (label, args), combo, (tag, lib) in itertools.product(cases, itertools.product(*sweeps) if sweeps else [()], libs):
            extras = "--extras" in args
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-frames", "1"] + ([] if extras else ["--no-extras"])
            cmd += [x for x in args if x != "--extras"]
            for name, v in combo:
                cmd += ["--param", "%s=%s" % (name, v)]
            name = "_".join([label, tag] + ["%s%s" % nv for nv in combo] + [str(rep)])
            log = os.path.join(out, name + ".log")
            env = dict(os.environ, PTCORE_LIB=lib)
            try:
                with open(log, "w") as f:
                    subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, env=env, timeout=a.timeout, cwd=ROOT)
                lines = [l for l in open(log) if l.startswith('{"metric"')]
                text = summary(json.loads(lines[-1])) if lines else "FAILED (see %s)" % log
            except subprocess.TimeoutExpired:
                text = "TIMEOUT"
            print("%-44s %s" % (name, text), flush=True)


if __name__ == "__main__":
    main()
