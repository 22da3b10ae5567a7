"""Developer aid: the generated image of a C5 step at the bench's batch (6 scenes of 64-128 objects) — HIP vs the fp32 oracle vs the
fp64 oracle, distribution of the differences."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

torch.set_num_threads(min(32, torch.get_num_threads()))
import oracle
from canonicalsg2im_amd import train as T
from canonicalsg2im_amd.synth import BatchConfig, make_batch, make_vocab
from fp64_band import batch_to64, trainstate_to64

B = int(os.environ.get("B", "6"))
vocab = make_vocab("clevr")
opt = T.make_opt(vocab, ["--image_size", "256,256", "--no_vgg_loss", "--batch_size", str(B)])
torch.manual_seed(4)
tr = T.Trainer(opt, torch.device("cuda:0"))
ts = T.oracle_state_from(tr, oracle)
ts64 = trainstate_to64(ts, oracle)
batch = make_batch(vocab, BatchConfig(B, 256, 64, 128, "closure"), seed=7)
G, D = tr.step([None if t is None else t.cuda() for t in batch])
torch.cuda.synchronize()
img = tr.last_model_out[0].detach().double().cpu()
img32 = oracle.train_step(ts, batch)[2].detach().double()
img64 = oracle.train_step(ts64, batch_to64(batch))[2].detach().double()
for name, a, b in (("hip - fp32 oracle", img, img32), ("hip - fp64 oracle", img, img64), ("fp32 oracle - fp64 oracle", img32, img64)):
    d = (a - b).abs()
    print("%-28s max %.3e  rel L2 %.3e  pixels over rtol 1e-4 + 1e-4: %d   99.99th pct %.3e" % (
        name, float(d.max()), float(d.norm() / b.norm()), int((d > 1e-4 * b.abs() + 1e-4).sum()), float(d.flatten().kthvalue(int(d.numel() * 0.9999))[0])))
print("image: max |pixel| %.3f  mean |pixel| %.4f  elements %d" % (float(img64.abs().max()), float(img64.abs().mean()), img64.numel()))
