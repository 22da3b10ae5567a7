import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalsg2im_amd import ops
torch.set_printoptions(linewidth=200, precision=1, sci_mode=False)
# A probe: w = identity (32x32), x[b,c,y,x] = pixel_index + c/100
Cin = Cout = 32; H = 16
pix = torch.arange(H * H, dtype=torch.float32).view(1, 1, H, H)
x = pix + torch.arange(Cin, dtype=torch.float32).view(1, Cin, 1, 1) / 100.0
w = torch.eye(32).view(32, 32, 1, 1).contiguous()
y = ops.conv2d(x.cuda(), w.cuda(), None, 1, 0).cpu()
print("identity weights, BN=32: max err", (y - x).abs().max().item())
print("y[0,:8,0,:6] (channel x pixel):"); print(y[0, :8, 0, :6])
print("x[0,:8,0,:6]:"); print(x[0, :8, 0, :6])
# B probe: x = one-hot channel c0 at every pixel -> y[:, n] = w[n, c0]
for c0 in (0, 5, 31):
    x1 = torch.zeros(1, Cin, H, H); x1[:, c0] = 1.0
    w1 = (torch.arange(Cout, dtype=torch.float32).view(Cout, 1) * 100 + torch.arange(Cin, dtype=torch.float32).view(1, Cin)).view(Cout, Cin, 1, 1)
    y1 = ops.conv2d(x1.cuda(), w1.cuda(), None, 1, 0).cpu()
    print("one-hot c0=%d: y[0,:8,0,0] =" % c0, y1[0, :8, 0, 0].tolist(), " expect", w1[:8, c0, 0, 0].tolist())
    print("      y[0,:4,3,5] =", y1[0, :4, 3, 5].tolist())
