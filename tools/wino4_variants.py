"""Developer A/B of compile-time variants of k_wino4_conv_v (csrc/wino4.hip: -DW4_SPLIT, the ablations).

    python tools/wino4_variants.py --build      HERE (no GPU): one libcsg_hip_<tag>.so per variant under csrc/build/variants/
                                                (only wino4.o is recompiled; the .so files travel with gpurun)
    python tools/wino4_variants.py              on the GPU box: every variant in its own process (CSG_HIP_LIB), alternating,
                                                two rounds; ms per launch on the generator's dominant shapes, ratio to `base`
    python tools/wino4_variants.py --one        (internal) time the library CSG_HIP_LIB names, print one JSON line
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "canonicalsg2im_amd", "csrc", "build", "variants")
# tag -> extra flags for wino4.hip ("valid": results are right; the ablations compute garbage and only time the loop)
VARIANTS = {
    "base": [],
    "abl_half_reads": ["-DW4_HALF_READS"],
    "abl_half_produce": ["-DW4_HALF_PRODUCE"],
    "abl_no_produce": ["-DW4_NO_PRODUCE"],
    "abl_no_barrier": ["-DW4_NO_BARRIER"],
    "abl_no_uload": ["-DW4_NO_ULOAD"],
    "abl_no_staging": ["-DW4_NO_STAGING"],
    "abl_no_staging_load": ["-DW4_NO_STAGING_LOAD"],
    "abl_mfma_only": ["-DW4_NO_PRODUCE", "-DW4_NO_ULOAD", "-DW4_NO_STAGING", "-DW4_NO_BARRIER"],
}
SHAPES = [(16, 128, 256, 256), (16, 128, 512, 128), (16, 128, 1024, 64), (16, 256, 128, 128), (16, 1024, 512, 32),
          (16, 32, 128, 256), (4, 128, 256, 256)]


def build():
    import __graft_entry__ as ge
    ge.build()
    os.makedirs(VDIR, exist_ok=True)
    objs = [ge._obj(s) for s in ge.SOURCES if s != "wino4.hip"]
    for tag, flags in VARIANTS.items():
        o = os.path.join(VDIR, "wino4_%s.o" % tag)
        so = os.path.join(VDIR, "libcsg_hip_%s.so" % tag)
        cmd = [ge._hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-value"] + \
            ge.EXTRA_FLAGS.get("wino4.hip", []) + flags + ["-c", os.path.join(ge.CSRC, "wino4.hip"), "-o", o]
        subprocess.run(cmd, check=True, cwd=ge.CSRC)
        subprocess.run([ge._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [o], check=True, cwd=ge.CSRC)
        print("built", so, flush=True)


def one():
    import torch
    from canonicalsg2im_amd import ops
    from canonicalsg2im_amd._lib import WinoDesc, check, lib, ptr, stream
    out = {}
    torch.manual_seed(0)
    for (B, Cin, Cout, H) in SHAPES:
        x = ops.nhwc(torch.randn(B, Cin, H, H, device="cuda").clamp_min(0))
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)
        up = ops.wino_pack(w, False, None, 4)
        y = ops.empty_nhwc(B, Cout, H, H, x.device)
        d = WinoDesc()
        d.B, d.H, d.W, d.Cin, d.x_cs, d.Cout, d.y_cs, d.act, d.slope = B, H, H, Cin, Cin, Cout, Cout, 0, 0.0
        nws = lib.csg_wino4_conv_workspace(d)
        ws = torch.empty(max(nws, 4) // 4, device="cuda")

        def call():
            check(lib.csg_wino4_conv(d, ptr(x), ptr(up), None, None, None, 0.0, ptr(y), ptr(ws), nws, stream()), "conv")

        for _ in range(5):
            call()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        # reference for validity: the direct sum on a small crop would cost more than the timing; a checksum is enough to see
        # that the valid variants agree with `base` bit for bit
        out["B%d %d->%d %d" % (B, Cin, Cout, H)] = (best, float(y.double().sum()))
    print(json.dumps(out), flush=True)


def main():
    tags = [t for t in VARIANTS if os.path.exists(os.path.join(VDIR, "libcsg_hip_%s.so" % t))]
    res = {t: [] for t in tags}
    for rnd in range(2):
        for t in tags:
            env = dict(os.environ, CSG_HIP_LIB=os.path.join(VDIR, "libcsg_hip_%s.so" % t))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                print(t, "FAILED", p.stderr[-500:], flush=True)
                continue
            res[t].append(json.loads(line[0]))
    peak = 157.3
    shapes = list(res["base"][0].keys())
    print("%-18s" % "variant" + "".join("%22s" % s for s in shapes) + "   (ms | executed frac | x base | same bits)")
    for t in tags:
        if not res[t]:
            continue
        row = "%-18s" % t
        for s in shapes:
            ms = min(r[s][0] for r in res[t])
            base = min(r[s][0] for r in res["base"])
            B, rest = s.split(" ", 1)
            cio, H = rest.rsplit(" ", 1)
            Cin, Cout = cio.split("->")
            ex = 2.0 * int(B[1:]) * int(H) ** 2 * 9 * int(Cin) * int(Cout) / 4 / 1e9
            same = res[t][0][s][1] == res["base"][0][s][1]
            row += "%8.3f %5.3f %5.3f %s" % (ms, ex / ms / peak, base / ms, "=" if same else "x")
        print(row, flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    elif "--one" in sys.argv:
        one()
    else:
        main()
