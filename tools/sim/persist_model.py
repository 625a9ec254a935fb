#!/usr/bin/env python3
"""persist_model.py -- what a bounce-spanning persistent schedule is worth, BEFORE it is built (review of round 4, item 1).

Today a batch of F frames is eight launch sequences T(b) -> S(b): every traversal launch T(b) carries bounce b of all frames
and ends on its longest rays (the tail: from the first wavefront that finds the feed dry to the last exit, 170 us), and the
streaming kernels (k_shade_fused S(b), k_raygen, k_accumulate) run with no traversal beside them.  The schedule modelled here:
bounce 0's traversal stays a launch of its own (entry points, work list), then ONE persistent launch runs the rest --
traversal wavefronts (4 of every 5) take rays of (frame f, bounce b) as soon as S(f, b-1) is done, frames in order; service
wavefronts (1 of every 5) shade tiles of (f, b) as soon as T(f, b) is complete.  Only the chain of the last frames' last
bounces is exposed.

Inputs are measurements of round 4 / 5's code on the driver's command (20 frames) and on one rank of eight (10 + 10 frames):
  * per-bounce launch spans, first exhaustion and exit times: profiles/r04_tailprof_two_loops_steps20.txt,
    profiles/r04_tailprof_before_share8.txt (PT_TAILPROF builds);
  * k_shade_fused / k_raygen / k_accumulate durations: profiles/r03_kernel_stats_steps20.txt, r04_kernel_stats_steps20.txt;
  * one wavefront fewer per SIMD costs 4-6 % of traversal throughput (occupancy sweep);
  * two 32-frame batches in flight (the hardware interleaves their kernels) run 7073 against 6130 Mrays/s.
The model is a fluid one (work in chip-microseconds, capacities as fractions of the chip), frame by frame:
  work_T(f, b)  = (first exhaustion of launch b, the part of a launch in which every wavefront is busy, + a quarter of the
                   170-us tail: the tail's iterations are real work at 45 of 64 lanes) / F
  latency_T     = a frame-bounce cannot complete sooner than `tail_us` after its last ray was handed out (a wavefront's own
                   tail: 50 us median, 110 us maximum)
                   (the rays of a small frame-bounce are handed out at once and then simply take their ~25 dependent steps:
                   that latency is this term, not a throughput limit)
  cap_T         = 0.95 with the service wavefronts idle, 0.80 with all of them busy (they contend like traversal wavefronts)
  work_S(f, b)  = S(b) / F at the full-chip kernel's rate; the service wavefronts reach `svc_rate` of that rate
Output: milliseconds for the batch and the ratio to today's, for a range of the two uncertain parameters (svc_rate, tail_us).
No GPU, no parity claim: a planning tool.  usage: python3 tools/sim/persist_model.py"""
import itertools

CASES = {
    # name: (frames, [span_us], [first_exh_us], [S_us for all frames], raygen_us, accumulate_us, measured_ms_today, wavefronts, rays per frame and bounce)
    "driver's command, whole frame, 1 x 20 frames": dict(
        F=20, span=[3223.8, 2381.6, 1458.2, 1013.2, 726.0, 549.6, 428.6, 353.4],
        exh=[2705.7, 2209.3, 1290.0, 844.9, 557.5, 379.6, 261.6, 230.2],
        # bench line (no profiler): trace_ms per bounce
        trace=[2988, 2549, 1568, 1098, 721, 537, 415, 399],
        S=[540, 580, 360, 250, 170, 130, 100, 100], raygen=490, accum=300, today_ms=13.10, waves=5120,
        rays=[924521, 769641, 425930, 281238, 183097, 126766, 89135, 65010]),
    "one rank of eight, 2 x 10 frames today": dict(
        F=20, span=[475.2, 322.4, 229.3, 177.0, 153.8, 109.9, 99.5, 97.2],
        exh=[220.7, 175.5, 124.6, 76.9, 42.4, 15.3, 16.4, 14.8],
        trace=[x * 1000 / 2 for x in [1.015, 0.76, 0.479, 0.4, 0.338, 0.225, 0.209, 0.188]],   # per 10-frame launch
        S=[2 * x for x in [40, 42, 27, 20, 14, 11, 9, 9]], raygen=80, accum=45, today_ms=2.68, waves=2560,
        rays=[119138, 97970, 53403, 34662, 22601, 15664, 11023, 8026]),
}
TAIL_FULL = 170.0     # first exhaustion -> last exit of a launch
LANES = 64


def simulate(c, svc_rate, tail_us, svc_share=0.2, dt=2.0):
    F, MB = c["F"], 8
    launches_per_bounce = 1 if c["waves"] == 5120 else 2     # today's schedule of the rank share: two 10-frame sequences
    per_launch_frames = F // launches_per_bounce
    # chip-us of traversal work per frame and bounce: the busy part of a launch + a quarter of its tail, per frame
    wT = [(c["exh"][b] + 0.25 * (c["span"][b] - c["exh"][b])) / per_launch_frames for b in range(MB)]
    # the profiler's launches are 5-8 % longer than the bench line's: scale to the bench line
    scale = sum(c["trace"]) / (sum(c["span"]) * (1 if launches_per_bounce == 1 else 1))
    wT = [w * scale for w in wT]
    wS = [c["S"][b] / F for b in range(MB)]
    # state per frame: bounce, remaining T work, remaining S work, phase
    # bounce 0's traversal is today's launch (with its tail); the persistent launch starts with every S(f, 0) ready
    # (a rank's share today: two 10-frame launches side by side; as ONE 20-frame launch: twice the busy part + one tail)
    t0 = c["raygen"] + (c["trace"][0] if launches_per_bounce == 1 else 2.0 * c["exh"][0] + (c["span"][0] - c["exh"][0])) + 10.0
    phase = ["S"] * F
    bounce = [0] * F
    remT = [0.0] * F
    remS = [wS[0]] * F
    handed_out_at = [None] * F
    t = 0.0
    done = 0
    while done < F and t < 1e6:
        # service: frames in order, whole service capacity to the first ready one (tiles of one frame keep 1024 wavefronts busy)
        cap_s = svc_rate * dt
        busy_s = 0.0
        for f in range(F):
            if phase[f] == "S" and cap_s > 0:
                use = min(cap_s, remS[f])
                remS[f] -= use
                cap_s -= use
                busy_s += use
                if remS[f] <= 1e-9:
                    if bounce[f] == MB - 1:
                        phase[f] = "done"
                        done += 1
                    else:
                        bounce[f] += 1
                        phase[f] = "T"
                        remT[f] = wT[bounce[f]]
                        handed_out_at[f] = None
        u_s = busy_s / (svc_rate * dt)
        cap_t = (0.95 - 0.15 * u_s) * dt
        for f in range(F):
            if phase[f] == "T":
                if remT[f] > 0:
                    use = min(cap_t, remT[f])
                    remT[f] -= use
                    cap_t -= use
                    if remT[f] <= 1e-9:
                        handed_out_at[f] = t
                elif t - handed_out_at[f] >= tail_us:
                    phase[f] = "S"
                    remS[f] = wS[bounce[f]]
        t += dt
    return (t0 + t + c["accum"]) / 1000.0


def main():
    for name, c in CASES.items():
        print(name)
        print("  today (measured) %.2f ms" % c["today_ms"])
        rows = []
        for svc_rate, tail_us in itertools.product((0.5, 0.65, 0.8, 1.0), (50.0, 80.0, 110.0)):
            ms = simulate(c, svc_rate, tail_us)
            rows.append((svc_rate, tail_us, ms))
            print("  service rate %.2f of the full-chip shade kernel, frame-bounce tail %3.0f us: %6.2f ms  (x%.3f)"
                  % (svc_rate, tail_us, ms, c["today_ms"] / ms))
        mid = [r for r in rows if r[0] == 0.65 and r[1] == 80.0][0]
        print("  central estimate (0.65, 80 us): %.2f ms = x%.3f" % (mid[2], c["today_ms"] / mid[2]))


if __name__ == "__main__":
    main()
