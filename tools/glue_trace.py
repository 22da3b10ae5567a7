"""The torch-native (glue) kernels of one training step out of a rocprofv3 --kernel-trace CSV: time by kernel family, and the
longest instances with the library kernels launched around them (which op they belong to).

    python tools/glue_trace.py <dir with *_kernel_trace.csv> [how many to list]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0)))
rows.sort()
marks = [i for i, r in enumerate(rows) if "k_crop_fwd" in r[2]]
if len(marks) < 7:
    raise SystemExit("only %d step marks" % len(marks))
i0, i1 = marks[-7], marks[-4]                    # one whole step (three crop launches per step), not the last one
win = rows[i0:i1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("at::native::", "")
    for a, b in (("vectorized_elementwise_kernel<4, ", "vec<"), ("elementwise_kernel_manual_unroll<128, 4, ", "unroll<"),
                 ("gpu_kernel_impl_nocast<", ""), ("std::array<char*, ", "arr")):
        n = n.replace(a, b)
    return n[:90]


fam = defaultdict(lambda: [0, 0])
glue = []
for k, (s, e, n, g) in enumerate(win):
    if "at::native" in n or "rocclr" in n or n.startswith("void at::"):
        fam[short(n).split("(")[0][:70]][0] += e - s
        fam[short(n).split("(")[0][:70]][1] += 1
        glue.append((e - s, k))
print("step: %d launches, %.2f ms of kernel time; glue: %d launches, %.3f ms" % (
    len(win), sum(e - s for s, e, _, _ in win) / 1e6, len(glue), sum(t for t, _ in glue) / 1e6))
for name, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:16]:
    print("  %8.3f ms %5d x %7.1f us  %s" % (t / 1e6, c, t / c / 1e3, name))
print("longest glue launches (us, grid) and the non-glue kernels around them:")


def neighbour(k, step):
    k += step
    while 0 <= k < len(win):
        if "at::native" not in win[k][2] and "rocclr" not in win[k][2]:
            return short(win[k][2]).split("(")[0][:40]
        k += step
    return "-"


for t, k in sorted(glue, reverse=True)[:top]:
    print("  %7.1f %9d  %-58s after %-40s before %s" % (t / 1e3, win[k][3], short(win[k][2]).split("(")[0][:58], neighbour(k, -1),
                                                     neighbour(k, 1)))
