#!/usr/bin/env python3
"""What does a HIP event pair around ONE launch time?  ("What's weak" 4: bench.py's per-launch events read
k_norm_apply_fwd at 100.6 us where rocprofv3 reads 52.8 us.)

Run phase (under rocprofv3 --kernel-trace --output-format csv):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/evr -- python3 tools/event_vs_rocprof.py --run
  2 warm-up steps, then 3 steps with the library's profiling OFF, then 3 steps with an event pair on EVERY launch
  (csg_prof_enable(1)), then 3 steps with events on the HBM kernels only (mode 3).  Prints the event-side per-kernel
  averages of the two profiled phases as JSON.
Analysis phase:
    python3 tools/event_vs_rocprof.py --analyze gpurun_out/evr/**/*kernel_trace.csv --events gpurun_out/evr_events.json
  splits rocprofv3's per-dispatch durations of the same kernels by phase (dispatch order) and prints both side by side.
"""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KERNELS = {            # library table name -> substring of the rocprofv3 kernel name
    "norm_apply_fwd": "k_norm_apply_fwd", "norm_bwd_reduce": "k_norm_bwd_reduce", "norm_bwd_dx": "k_norm_bwd_dx",
    "splitk_epilogue": "k_splitk_epilogue", "wino_conv": "k_wino_conv2", "wino_wgrad": "k_wino_wgrad",
    "igemm_wgrad": "k_igemm_wgrad", "act_bwd": "k_act_bwd", "upsample2x_fwd": "k_upsample2x_fwd",
}
PHASES = (("warmup", 2, 0), ("events off", 3, 0), ("events on every launch", 3, 1), ("events on HBM kernels only", 3, 3))


def run():
    import torch
    from canonicalsg2im_amd import _lib, train as T
    from canonicalsg2im_amd.synth import BASELINE_CONFIGS, BatchConfig, make_batch, make_vocab
    dev = torch.device("cuda:0")
    vocab = make_vocab("coco")
    cfg = BASELINE_CONFIGS["C3"]["cfg"]
    opt = T.make_opt(vocab, ["--image_size", "256,256", "--no_vgg_loss", "--batch_size", "16"])
    torch.manual_seed(0)
    tr = T.Trainer(opt, dev)
    batch = [None if t is None else t.to(dev) for t in make_batch(vocab, BatchConfig(16, 256, cfg.min_objects,
                                                                                       cfg.max_objects, cfg.graph), 0)]
    out = {}
    for name, steps, mode in PHASES:
        _lib.prof_reset()
        _lib.prof_enable(mode)
        for _ in range(steps):
            tr.step(batch)
        torch.cuda.synchronize()
        if mode:
            prof = _lib.prof_read()
            out[name] = {k: {"launches_per_step": v[1] / steps, "avg_us": 1000.0 * v[0] / max(v[1], 1)}
                         for k, v in prof.items() if k in KERNELS}
        _lib.prof_enable(0)
    print(json.dumps(out))


def analyze(pattern, events):
    files = sorted(glob.glob(pattern, recursive=True))
    assert files, "no kernel trace matches " + pattern
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ev = json.load(open(events)) if events else {}
    total_steps = sum(p[1] for p in PHASES)
    print("%-18s %-30s %9s %12s %12s" % ("kernel", "phase", "launches", "rocprof us", "events us"))
    for kname, sub in KERNELS.items():
        durs = [(e - s) / 1000.0 for s, e, n in rows if sub in n]
        if not durs or len(durs) % total_steps:
            print("%-18s launches %d not a multiple of %d steps: skipped" % (kname, len(durs), total_steps))
            continue
        per = len(durs) // total_steps
        off = 0
        for pname, steps, mode in PHASES:
            d = durs[off:off + per * steps]
            off += per * steps
            e = ev.get(pname, {}).get(kname, {}).get("avg_us")
            print("%-18s %-30s %9d %12.2f %12s" % (kname, pname, per, sum(d) / len(d), "%.2f" % e if e else "-"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--analyze", default=None)
    ap.add_argument("--events", default=None)
    a = ap.parse_args()
    if a.run:
        run()
    elif a.analyze:
        analyze(a.analyze, a.events)
