import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from oracle import unet_torch
from read_amd import synthetic
from read_amd.gated_conv import PackedGatedConv, gated_conv
torch.manual_seed(0)
def run(cin, cout, k, H, W, wq, xq):
    st = synthetic.make_unet_state([("L", cin, cout, k)], 3)
    b = "L.block."
    if wq:
        for n in ("conv_f.weight", "conv_m.weight"):
            w = np.asarray(st[b + n]); st[b + n] = (np.round(w * 64) / 64).astype(np.float32)
    pk = PackedGatedConv(*[st[b + n] for n in ("conv_f.weight", "conv_f.bias", "conv_m.weight", "conv_m.bias", "norm.weight", "norm.bias", "norm.running_mean", "norm.running_var")], src_channels=[cin])
    x = torch.randn(cin, H, W)
    if xq: x = torch.round(x * 8) / 8
    ref = unet_torch.basic_conv(st, "L", x[None], k, stride=2, elu=False)[0]
    got = gated_conv(pk, [(x.permute(1, 2, 0).contiguous().cuda(), 0)], stride=2, elu=False).cpu().permute(2, 0, 1)
    return float((got - ref).abs().max())
for k in (3, 4):
    for (wq, xq) in ((0, 0), (1, 0), (0, 1), (1, 1)):
        print("k", k, "w exact" if wq else "w full ", "x exact" if xq else "x full ", "max err %.3e" % run(64, 64, k, 20, 36, wq, xq), flush=True)
for cin in (32, 64, 128):
    print("k4 cin", cin, "%.3e" % run(cin, 64, 4, 20, 36, 0, 0))
def run2(cin, cout, k, H, W, lo, hi):
    st = synthetic.make_unet_state([("L", cin, cout, k)], 3)
    b = "L.block."
    pk = PackedGatedConv(*[st[b + n] for n in ("conv_f.weight", "conv_f.bias", "conv_m.weight", "conv_m.bias", "norm.weight", "norm.bias", "norm.running_mean", "norm.running_var")], src_channels=[cin])
    x = torch.zeros(cin, H, W)
    x[lo:hi] = torch.round(torch.randn(hi - lo, H, W) * 8) / 8
    ref = unet_torch.basic_conv(st, "L", x[None], k, stride=2, elu=False)[0]
    got = gated_conv(pk, [(x.permute(1, 2, 0).contiguous().cuda(), 0)], stride=2, elu=False).cpu().permute(2, 0, 1)
    return float((got - ref).abs().max())
print("k4 cin 64, x in chunk 0 only: %.3e   chunk 1 only: %.3e" % (run2(64, 64, 4, 20, 36, 0, 32), run2(64, 64, 4, 20, 36, 32, 64)))
for c in range(32, 64, 8):
    print("  x in channels", c, c + 8, "%.3e" % run2(64, 64, 4, 20, 36, c, c + 8))
# single tap nonzero weights? use x nonzero at one pixel -> each output sees one tap
