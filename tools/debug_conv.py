import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalsg2im_amd import ops
torch.manual_seed(0)
def run(B, Cin, Cout, H, K, s, p):
    x = torch.randn(B, Cin, H, H); w = torch.randn(Cout, Cin, K, K) / (Cin*K*K) ** 0.5
    ref = F.conv2d(x, w, None, stride=s, padding=p)
    y = ops.conv2d(x.cuda(), w.cuda(), None, s, p).cpu()
    d = (y - ref).abs()
    print("B%d Cin%d Cout%d H%d K%d s%d p%d: max err %.3e  frac_bad %.3f" % (B, Cin, Cout, H, K, s, p, d.max(), (d > 1e-3).float().mean()))
    if d.max() > 1e-3:
        bad = (d > 1e-3)
        print("   bad per out-channel:", bad.float().mean(dim=(0, 2, 3))[:16].tolist())
        print("   bad per row y:", bad.float().mean(dim=(0, 1, 3))[:16].tolist())
        print("   got/ref [0,0,0,:6]", y[0, 0, 0, :6].tolist(), ref[0, 0, 0, :6].tolist())
        # is it a scaled / permuted version?
        print("   ratio mean", (y / ref).median().item())
run(1, 32, 128, 16, 1, 1, 0)    # M=256 K=32 N=128: no OOB at all
run(1, 64, 128, 16, 1, 1, 0)    # two k tiles
run(1, 32, 32, 16, 1, 1, 0)     # BN=32
run(1, 8, 128, 16, 1, 1, 0)     # K tail
run(1, 32, 128, 10, 1, 1, 0)    # M tail (100 rows)
run(1, 32, 128, 16, 3, 1, 1)    # padding
run(2, 32, 128, 8, 3, 1, 1)     # two images per tile
