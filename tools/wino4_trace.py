#!/usr/bin/env python3
"""Phase timeline of k_wino4_conv_v blocks (developer build with -DW4_TRACE: thread 0 of each block leaves shader-clock
timestamps at the phase boundaries).
    python tools/wino4_trace.py --build                 # here (cross-compiles canonicalsg2im_amd/csrc/build/libcsg_hip_trace.so)
    python tools/wino4_trace.py B Cin Cout H W          # on the GPU box
Markers: 0 entry | 1 stage 0 in LDS | 2 pipeline primed (V[0], raw[1]) | 3 main loop done | 4 epilogue R written |
5 first two output columns stored | 6 end."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "canonicalsg2im_amd", "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.environ.get("W4_TRACE_LIB") or os.path.join(OBJ, "libcsg_hip_trace.so")
sys.path.insert(0, ROOT)

if sys.argv[1:] == ["--build"]:
    import __graft_entry__ as g
    g.build()
    obj = os.path.join(OBJ, "wino4_trace.o")
    subprocess.run([g._hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-value",
                    "-fno-slp-vectorize", "-DW4_TRACE"] + os.environ.get("W4_DEFS", "").split() + ["-c", os.path.join(CSRC, "wino4.hip"), "-o", obj], check=True, cwd=CSRC)
    objs = [obj if s == "wino4.hip" else g._obj(s) for s in g.SOURCES]
    subprocess.run([g._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, check=True, cwd=CSRC)
    print("built", LIB)
    sys.exit(0)

os.environ["CSG_HIP_LIB"] = LIB
import numpy as np  # noqa: E402
import torch  # noqa: E402
from canonicalsg2im_amd import ops  # noqa: E402
from canonicalsg2im_amd._lib import WinoDesc, check, lib, ptr, stream  # noqa: E402

B, Cin, Cout, H, W = [int(a) for a in sys.argv[1:6]]
x = ops.nhwc(torch.randn(B, Cin, H, W, device="cuda").clamp_min(0))
w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)
up = ops.wino_pack(w, False, None, 4)
y = ops.empty_nhwc(B, Cout, H, W, x.device)
d = WinoDesc()
d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = B, H, W, Cin, Cin, Cout, Cout, 0, 0.0
for _ in range(3):
    check(lib.csg_wino4_conv(d, ptr(x), ptr(up), None, None, None, 0.0, ptr(y), None, 0, stream()), "conv")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
check(lib.csg_wino4_conv(d, ptr(x), ptr(up), None, None, None, 0.0, ptr(y), None, 0, stream()), "conv")
e1.record()
torch.cuda.synchronize()
nblk = B * ((H + 15) // 16) * ((W + 31) // 32) * ((Cout + 63) // 64)
n = min(nblk, 8192)
buf = (ctypes.c_ulonglong * (n * 8))()
raw = ctypes.CDLL(LIB)
assert raw.csg_wino4_trace_read(buf, n * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
print("B %d Cin %d Cout %d %dx%d: %d blocks (%d traced), kernel %.3f ms" % (B, Cin, Cout, H, W, nblk, n, e0.elapsed_time(e1)))
names = ["0-1 first stage lands", "1-2 prime (V[0], raw[1])", "2-3 main loop", "3-4 epilogue: R to LDS",
         "4-5 epilogue: columns 0,1", "5-6 epilogue: columns 2,3", "0-6 whole block"]
for lo, hi, tag in ((0, min(256, n), "first wave of blocks"), (min(256, n), n, "later blocks")):
    if hi <= lo:
        continue
    tt = t[lo:hi]
    print("%s [%d, %d):" % (tag, lo, hi))
    for i, nm in enumerate(names):
        dlt = (tt[:, 6] - tt[:, 0]) if i == 6 else (tt[:, i + 1] - tt[:, i])
        print("  %-28s median %8.0f cycles   p10 %8.0f   p90 %8.0f" % (nm, np.median(dlt), np.percentile(dlt, 10),
                                                                      np.percentile(dlt, 90)))
buf2 = (ctypes.c_ulonglong * (n * 8))()
assert raw.csg_wino4_trace2_read(buf2, n * 8) == 0
t2 = np.frombuffer(buf2, dtype=np.uint64).reshape(n, 8).astype(np.int64)
persistent = bool((t2[:, 7] > 0).any())
# stage-level markers per ITEM (persistent form) or per block: 0 top | 1 main loop starts | 2 stages 0, 1 done | 4 all but the
# last two stages done | 3 main loop done | 6 epilogue done | 7 (persistent) V[0] of the next item formed
tt = t2[(t2[:, 7] > 0) if persistent else (t2[:, 3] > 0)]
if len(tt) == 0:
    tt = t2[t2[:, 3] > 0]
    persistent = False
print("%s, %d rows:" % ("persistent form, items with a successor" if persistent else "one block per item", len(tt)))
rows = [("0-1 zero (+ next plan)", tt[:, 1] - tt[:, 0]), ("1-2 stages 0, 1", tt[:, 2] - tt[:, 1]),
        ("2-4 middle stages", tt[:, 4] - tt[:, 2]), ("4-3 last two stages", tt[:, 3] - tt[:, 4]),
        ("0-3 plan + main loop", tt[:, 3] - tt[:, 0])]
if persistent:
    rows += [("3-6 epilogue", tt[:, 6] - tt[:, 3]), ("6-7 next ring + V[0]", tt[:, 7] - tt[:, 6]), ("0-7 whole item", tt[:, 7] - tt[:, 0])]
for nm, dlt in rows:
    print("  %-28s median %8.0f cycles   p10 %8.0f   p90 %8.0f" % (nm, np.median(dlt), np.percentile(dlt, 10), np.percentile(dlt, 90)))
span = (t[:, 6].max() - t[:, 0].min())
print("span first entry -> last exit (traced blocks): %d cycles; kernel %.3f ms => %.0f MHz if they cover the launch" %
      (span, e0.elapsed_time(e1), span / e0.elapsed_time(e1) / 1e3))
