"""HBM traffic of the F(4x4,3x3) convolution launches of one C3 step: ALGORITHMIC bytes (every operand of a launch once —
input, output, residual / gate / modulation operands, the packed weights; ops.WINO4_AUDIT) against the PMC capture
(profiles/pmc_traffic.json: FETCH_SIZE x2 + WRITE_SIZE per launch, launch-weighted over both forms of k_wino4_conv_v<4, *>).

    python tools/wino4_traffic_model.py            CSG_SPADE_JOINT=1 (default) and 0, one eager step each (GPU box)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import torch
    import __graft_entry__ as ge
    ge.build()
    from canonicalsg2im_amd import ops, train as T
    from canonicalsg2im_amd.synth import BASELINE_CONFIGS, BatchConfig, make_batch, make_vocab
    base = BASELINE_CONFIGS["C3"]
    vocab, cfg = make_vocab(base["vocab"]), base["cfg"]
    opt = T.make_opt(vocab, ["--image_size", "256,256", "--no_vgg_loss", "--batch_size", "16"])
    torch.manual_seed(0)
    tr = T.Trainer(opt, torch.device("cuda:0"))
    tr.use_graphs = False
    batch = [None if t is None else t.cuda() for t in make_batch(vocab, BatchConfig(16, 256, cfg.min_objects, cfg.max_objects, cfg.graph), seed=1)]
    tr.step(batch)
    ops.WINO4_AUDIT = {}
    tr.step(batch)
    torch.cuda.synchronize()
    print(json.dumps({k: v for k, v in ops.WINO4_AUDIT.items()}))


def main():
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    hits = [v for k, v in pmc["kernels"].items() if k.startswith("k_wino4_conv_v<4")]
    n = sum(v["launches_profiled"] for v in hits)
    measured = sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in hits) / n
    print("PMC capture %s: %d launches of k_wino4_conv_v<4,*>, %.1f MB per launch (launch-weighted)" % (pmc.get("capture"), n, measured / 1e6))
    for joint in ("1", "0"):
        env = dict(os.environ, CSG_SPADE_JOINT=joint, CSG_GRAPHS="0")
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("FAILED", p.stderr[-800:])
            continue
        a = json.loads(line[0])
        launches = sum(v[0] for v in a.values())
        total = sum(v[1] for v in a.values())
        print("CSG_SPADE_JOINT=%s: %d launches per step, %.1f MB algorithmic per launch, %.2f GB per step" %
              (joint, launches, total / launches / 1e6, total / 1e9))
        for k, v in sorted(a.items(), key=lambda kv: -kv[1][1]):
            print("    %-28s %3d launches  %8.1f MB each" % (k, v[0], v[1] / v[0] / 1e6))
        if joint == "1":
            print("    measured / algorithmic = %.2f" % (measured / (total / launches)))


if __name__ == "__main__":
    one() if "--one" in sys.argv else main()
