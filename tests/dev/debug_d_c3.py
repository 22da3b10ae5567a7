"""Where do the scale-0 discriminator gradients of the C3 full-width test leave the fp32 noise band?
D forward (training mode, one call) on the REAL image: HIP vs oracle fp32 vs oracle fp64, every feature map; then the
gradients of sum(all prediction maps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
import oracle
from canonicalsg2im_amd import train as T
from canonicalsg2im_amd.synth import BatchConfig, make_batch, make_vocab
from fp64_band import state_to64, errors

kind, seed, bseed, lo = (sys.argv[1:] + ["coco", "0", "3", "1"])[:4]
vocab = make_vocab(kind)
opt = T.make_opt(vocab, ["--image_size", "256,256", "--no_vgg_loss", "--batch_size", "2"])
torch.manual_seed(int(seed))
tr = T.Trainer(opt, torch.device("cuda:0"))
ts = T.oracle_state_from(tr, oracle)
d32, d64 = ts.d, state_to64(ts.d)
batch = make_batch(vocab, BatchConfig(2, 256, int(lo), 30, "random"), seed=int(bseed))
imgs, objs, boxes = batch[0], batch[1], batch[2]
print("objects per image:", (objs[..., 0] != 0).sum(1).tolist())
D = tr.discriminator.img_discriminator
fh = D(imgs.cuda(), objs.cuda(), boxes.cuda())
f32 = oracle.multiscale_discriminator(d32, vocab, 256, imgs, objs, boxes, True)
f64 = oracle.multiscale_discriminator(d64, vocab, 256, imgs.double(), objs, boxes.double(), True)
for i in range(2):
    for j in range(5):
        eh, e3 = errors(fh[i][j], f64[i][j]), errors(f32[i][j], f64[i][j])
        print("feat scale %d layer %d  hip l2 %.2e max %.2e | fp32 l2 %.2e max %.2e | shape %s" % (i, j, eh[0], eh[1], e3[0], e3[1], tuple(f64[i][j].shape)))
loss_h = sum(f[-1].sum() for f in fh)
loss_h.backward()
sum(f[-1].sum() for f in f32).backward()
sum(f[-1].sum() for f in f64).backward()
for k, p in D.named_parameters():
    if p.grad is not None and k in d32 and d32[k].grad is not None:
        eh, e3 = errors(p.grad, d64[k].grad), errors(d32[k].grad, d64[k].grad)
        print("grad %-50s hip l2 %.2e | fp32 l2 %.2e" % (k, eh[0], e3[0]))
