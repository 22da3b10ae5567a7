"""Developer probe: does the HIP-graph path (canonicalsg2im_amd/graphs.py) capture and replay on this stack?
Each variant runs in a child process (a crash in one does not hide the others); stage markers go to stderr."""
import faulthandler
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(use_img_disc, size, ngf, B):
    faulthandler.enable()
    import torch
    from canonicalsg2im_amd import graphs, train as T
    from canonicalsg2im_amd.synth import BatchConfig, make_batch, make_vocab
    dev = torch.device("cuda:0")
    vocab = make_vocab("tiny")
    opt = T.make_opt(vocab, ["--image_size", "%d,%d" % (size, size), "--ngf", str(ngf), "--ndf", str(ngf), "--gconv_dim", "32",
                             "--gconv_hidden_dim", "64", "--gconv_num_layers", "2", "--embedding_dim", "8",
                             "--no_vgg_loss", "--batch_size", str(B), "--use_img_disc", str(use_img_disc)])
    torch.manual_seed(0)
    tr = T.Trainer(opt, dev)
    print("graphs:", tr.graphs is not None, file=sys.stderr, flush=True)
    bs = [[None if t is None else t.to(dev) for t in make_batch(vocab, BatchConfig(B, size, 2, 6, "packed"), seed=i)]
          for i in range(3)]
    orig = graphs._GraphSet.run

    def run(self, name, fn):
        print("  graph", name, "capture" if name not in self.graphs else "replay", file=sys.stderr, flush=True)
        orig(self, name, fn)
        torch.cuda.synchronize()
        print("  graph", name, "done", file=sys.stderr, flush=True)
    graphs._GraphSet.run = run
    for it in range(5):
        print("step", it, file=sys.stderr, flush=True)
        G, D = tr.step(bs[it % 3])
        torch.cuda.synchronize()
        print("step", it, "ok", float(G["total_loss"]), float(D["total_img_loss"]), file=sys.stderr, flush=True)
    graphs._GraphSet.run = orig
    t0 = time.perf_counter()
    for it in range(10):
        tr.step(bs[it % 3])
    torch.cuda.synchronize()
    print("replay ms/step", 100.0 * (time.perf_counter() - t0), file=sys.stderr, flush=True)
    tr.use_graphs = False
    tr.step(bs[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(10):
        tr.step(bs[it % 3])
    torch.cuda.synchronize()
    print("eager ms/step", 100.0 * (time.perf_counter() - t0), file=sys.stderr, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(*[int(a) for a in sys.argv[2:6]])
        sys.exit(0)
    variants = [({"CSG_D_SCALE_STREAMS": "0"}, (1, 64, 8, 4)), ({"CSG_D_SCALE_STREAMS": "1"}, (1, 64, 8, 4)),
                ({"CSG_D_SCALE_STREAMS": "0"}, (0, 64, 8, 4)), ({"CSG_D_SCALE_STREAMS": "1"}, (0, 64, 8, 4))]
    for env, args in variants:
        print("==== variant", env, args, flush=True)
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"] + [str(a) for a in args], env=e,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        lines = r.stdout.splitlines()
        print("\n".join(lines[:80]))
        print("rc", r.returncode, flush=True)
