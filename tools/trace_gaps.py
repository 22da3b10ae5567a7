"""Busy / idle accounting of a rocprofv3 --kernel-trace CSV over whole training steps.

    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [steps]

Step boundaries are read off `k_crop_fwd` (three launches per step of the default recipe: the object discriminator's
generator pass and its two discriminator passes) or, without the object discriminator, off `k_layout_fwd` bursts.
Prints wall time per step, the union of kernel intervals (busy), the sum of kernel durations (> busy where the two
PatchGAN scales overlap), the idle gaps by size class and the kernels by time."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    per_step_marks = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [s for s, e, n in rows if n.startswith("csg::k_crop_fwd") or n.startswith("k_crop_fwd") or "k_crop_fwd" in n]
    bounds = marks[::per_step_marks]
    if len(bounds) < nsteps + 2:
        raise SystemExit("only %d step marks" % len(bounds))
    t0, t1 = bounds[-nsteps - 1], bounds[-1]
    win = [(s, e, n) for s, e, n in rows if s >= t0 and s < t1]
    wall = (t1 - t0) / nsteps / 1e6
    # union of intervals
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    for s, e, n in win:
        if cur_e is None:
            cur_s, cur_e = s, e
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
    busy += cur_e - cur_s
    ksum = defaultdict(lambda: [0, 0])
    for s, e, n in win:
        k = n.replace("(anonymous namespace)::", "").split("(")[0]
        ksum[k][0] += e - s
        ksum[k][1] += 1
    total = sum(v[0] for v in ksum.values())
    print("steps %d  wall %.2f ms/step  busy %.2f ms/step  sum of kernel durations %.2f ms/step  launches %.0f/step"
          % (nsteps, wall, busy / nsteps / 1e6, total / nsteps / 1e6, len(win) / nsteps))
    cls = [(2e3, "<2us"), (5e3, "2-5us"), (10e3, "5-10us"), (20e3, "10-20us"), (50e3, "20-50us"), (200e3, "50-200us"),
           (1e12, ">200us")]
    acc = defaultdict(lambda: [0, 0])
    for g in gaps:
        for lim, name in cls:
            if g < lim:
                acc[name][0] += g
                acc[name][1] += 1
                break
    print("idle %.2f ms/step in %d gaps/step:" % (sum(gaps) / nsteps / 1e6, len(gaps) / nsteps))
    for lim, name in cls:
        if acc[name][1]:
            print("   %-9s %7.1f gaps/step  %6.2f ms/step" % (name, acc[name][1] / nsteps, acc[name][0] / nsteps / 1e6))
    print("kernels by time (ms/step, launches/step, avg us):")
    for k, (t, c) in sorted(ksum.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("ROWS", "40"))]:
        print("   %8.3f %7.1f %8.1f  %s" % (t / nsteps / 1e6, c / nsteps, t / c / 1e3, k[:170]))


if __name__ == "__main__":
    main()
