"""Timing of formulations of the 3D IC generator's layers on MI355X (stock MIOpen / rocBLAS only)."""
import time, torch, torch.nn as nn, torch.nn.functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

x1 = torch.rand(1, 2, 64, 64, 64, device=dev, requires_grad=True)
ct1 = nn.ConvTranspose3d(2, 8, 5, padding=2, stride=2, output_padding=1).to(dev)
ct2 = nn.ConvTranspose3d(8, 8, 5, padding=2, stride=1, output_padding=0).to(dev)
c3 = nn.Conv3d(8, 2, 1).to(dev)
y1 = torch.sigmoid(ct1(x1)).detach().requires_grad_(True)

def fb(fn, inp):
    def run():
        out = fn(inp)
        out.backward(torch.ones_like(out))
    return run

for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    print(f"cudnn.benchmark={bench}")
    print("  layer1 ConvTranspose3d(2->8,s2) fwd+bwd   %.2f ms" % timed(fb(ct1, x1)))
    print("  layer2 ConvTranspose3d(8->8,s1) fwd+bwd   %.2f ms" % timed(fb(ct2, y1)))
    w2 = ct2.weight.detach().flip(2, 3, 4).transpose(0, 1).contiguous().requires_grad_(True)   # [co, ci, 5,5,5]
    print("  layer2 as conv3d(flipped W)    fwd+bwd    %.2f ms" % timed(fb(lambda t: F.conv3d(t, w2, ct2.bias, padding=2), y1)))
    ref = ct2(y1); alt = F.conv3d(y1, w2, ct2.bias, padding=2)
    print("     max |diff| %.2e" % (ref - alt).abs().max().item())
    y1cl = y1.detach().contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    print("  layer2 conv3d channels_last_3d fwd+bwd    %.2f ms" % timed(fb(lambda t: F.conv3d(t, w2, ct2.bias, padding=2), y1cl)))
    # separable trick is not available (dense 5^3 kernel); im2col in z-slabs + matmul
    def im2col_mm(t):
        outs = []
        for z0 in range(0, 128, 16):
            zs = slice(max(z0 - 2, 0), min(z0 + 18, 128))
            sl = F.pad(t[:, :, zs], (2, 2, 2, 2, 2 if z0 == 0 else 0, 2 if z0 + 16 >= 128 else 0))
            cols = sl.unfold(2, 5, 1).unfold(3, 5, 1).unfold(4, 5, 1)          # [1,8,16,128,128,5,5,5]
            cols = cols.permute(0, 2, 3, 4, 1, 5, 6, 7).reshape(-1, 1000)
            outs.append((cols @ w2.reshape(8, 1000).t()).reshape(1, 16, 128, 128, 8))
        return torch.cat(outs, 1).permute(0, 4, 1, 2, 3) + ct2.bias.view(1, 8, 1, 1, 1)
    print("  layer2 im2col(z-slabs)+matmul  fwd+bwd    %.2f ms" % timed(fb(im2col_mm, y1)))
    print("     max |diff| %.2e" % (ref - im2col_mm(y1)).abs().max().item())
    y2 = ct2(y1).detach().requires_grad_(True)
    print("  layer3 Conv3d 1x1 fwd+bwd                 %.2f ms" % timed(fb(c3, y2)))
