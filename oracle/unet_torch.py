"""ORACLE — TEST INFRASTRUCTURE ONLY.

Functional torch-fp32 (CPU) restatement of READ's refinement CNN and descriptor lookup:
READ/models/unet.py:11-285, READ/models/texture.py:42-70, READ/models/compose.py:125-181.
Written against a flat state dict (name -> tensor) instead of the reference's module classes;
validated against the reference's own modules imported from /root/reference by
tests/golden/make_golden.py (which also writes the committed golden vectors).

Floating point: convolution summation order differs between oneDNN (here), cuDNN (the
reference's deployment) and the MFMA kernels, so CNN parity is stated as a tolerance
(max-abs / PSNR in the tests), never bit-exact.
"""
import torch
import torch.nn.functional as F

BASE = 32
NUM_RES = 4


def _t(v):
    return v if torch.is_tensor(v) else torch.from_numpy(v)


TRAINING = False      # True inside unet_forward(..., training=True): nn.BatchNorm2d in .train() (batch statistics, momentum 0.1)


def basic_conv(st, path, x, k, stride=1, elu=True):
    """BasicConv.forward, unet.py:44-53: BN( act(conv_f x) * sigmoid(conv_m x) ); zero padding int((k-1)/2).  BN = nn.BatchNorm2d
    (unet.py:40,51): running statistics in .eval(); in .train() the batch statistics (biased variance) normalise and the
    running buffers move by momentum 0.1 (unbiased variance) — torch's F.batch_norm(training=True) does both in place."""
    pad = int((k - 1) / 2)
    b = path + ".block."
    f = F.conv2d(x, _t(st[b + "conv_f.weight"]), _t(st[b + "conv_f.bias"]), stride=stride, padding=pad)
    m = F.conv2d(x, _t(st[b + "conv_m.weight"]), _t(st[b + "conv_m.bias"]), stride=stride, padding=pad)
    if elu:
        f = F.elu(f)
    y = f * torch.sigmoid(m)
    return F.batch_norm(y, _t(st[b + "norm.running_mean"]), _t(st[b + "norm.running_var"]),
                        _t(st[b + "norm.weight"]), _t(st[b + "norm.bias"]), training=TRAINING, momentum=0.1, eps=1e-5)


def res_blocks(st, prefix, x):
    """EBlock/DBlock = 4 x ResBlock: x + BC(BC(x)) (unet.py:11-20,56-76)."""
    for j in range(NUM_RES):
        p = f"{prefix}.layers.{j}.main."
        x = basic_conv(st, p + "1", basic_conv(st, p + "0", x, 3), 3, elu=False) + x
    return x


def scm(st, name, x):
    """SCM.forward, unet.py:103-106."""
    y = basic_conv(st, name + ".main.0", x, 3)
    y = basic_conv(st, name + ".main.1", y, 1)
    y = basic_conv(st, name + ".main.2", y, 3)
    y = basic_conv(st, name + ".main.3", y, 1)
    return basic_conv(st, name + ".conv", torch.cat([x, y], 1), 1, elu=False)


def fam(st, name, x1, x2):
    """FAM.forward, unet.py:114-117."""
    return x1 + basic_conv(st, name + ".merge", x1 * x2, 3, elu=False)


def aff(st, name, xs):
    """AFF.forward, unet.py:87-89."""
    y = basic_conv(st, name + ".conv.0", torch.cat(xs, 1), 1)
    return basic_conv(st, name + ".conv.1", y, 3, elu=False)


def unet_forward(st, x, x2, x4, x8, taps=None, training=False):
    """UNet.forward, unet.py:202-285.  `taps` (dict) optionally receives named intermediates.  training=True: the module in
    .train() — every BatchNorm uses batch statistics and updates st's running_mean / running_var tensors in place."""
    global TRAINING
    prev, TRAINING = TRAINING, bool(training)
    try:
        return _unet_forward(st, x, x2, x4, x8, taps)
    finally:
        TRAINING = prev


def _unet_forward(st, x, x2, x4, x8, taps=None):
    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v
    near = lambda t, s: F.interpolate(t, scale_factor=s)                       # nearest, unet.py:239-250
    up4 = lambda t: F.interpolate(t, scale_factor=4, mode="bilinear", align_corners=False)   # unet.py:200

    z2 = tap("z2", scm(st, "SCM2", x2))
    z4 = tap("z4", scm(st, "SCM1", x4))
    z8 = tap("z8", scm(st, "SCM0", x8))
    res1 = tap("res1", res_blocks(st, "Encoder.0", basic_conv(st, "feat_extract.0", x, 3)))
    z = fam(st, "FAM2", basic_conv(st, "feat_extract.1", res1, 3, stride=2), z2)
    res2 = tap("res2", res_blocks(st, "Encoder.1", z))
    z = fam(st, "FAM1", basic_conv(st, "feat_extract.2", res2, 3, stride=2), z4)
    res3 = tap("res3", res_blocks(st, "Encoder.2", z))
    z = fam(st, "FAM0", basic_conv(st, "feat_extract.6", res3, 3, stride=2), z8)
    z = tap("zb", res_blocks(st, "Encoder.3", z))

    z12, z13 = near(res1, 0.5), near(res1, 0.25)
    z21, z23 = near(res2, 2), near(res2, 0.5)
    z32, z31 = near(res3, 2), near(res3, 4)
    z43 = near(z, 2)
    z42 = near(z43, 2)
    z41 = near(z42, 2)
    r1 = tap("aff0", aff(st, "AFFs.0", [res1, z21, z31, z41]))
    r2 = tap("aff1", aff(st, "AFFs.1", [z12, res2, z32, z42]))
    r3 = tap("aff2", aff(st, "AFFs.2", [z13, z23, res3, z43]))

    z = res_blocks(st, "Decoder.0", z)
    z = up4(basic_conv(st, "feat_extract.7", z, 4, stride=2))
    z = res_blocks(st, "Decoder.1", basic_conv(st, "Convs.0", torch.cat([z, r3], 1), 1))
    z = up4(basic_conv(st, "feat_extract.3", z, 4, stride=2))
    z = res_blocks(st, "Decoder.2", basic_conv(st, "Convs.1", torch.cat([z, r2], 1), 1))
    z = up4(basic_conv(st, "feat_extract.4", z, 4, stride=2))
    z = tap("d3", res_blocks(st, "Decoder.3", basic_conv(st, "Convs.2", torch.cat([z, r1], 1), 1)))
    return basic_conv(st, "feat_extract.5", z, 3, elu=False)


def point_texture_forward(texture_1cn, ids_bhw):
    """PointTexture.forward, texture.py:42-70: out[b,c,y,x] = texture[0,c,ids[b,y,x]]."""
    tex = _t(texture_1cn)[0]
    ids = _t(ids_bhw).long()
    return tex[:, ids].permute(1, 0, 2, 3)


def net_and_texture_forward(st, texture_1cn, index_maps):
    """NetAndTexture.forward for one item (compose.py:125-181): gather every scale, run the net."""
    feats = [point_texture_forward(texture_1cn, _t(i)[None] if _t(i).dim() == 2 else _t(i)) for i in index_maps]
    return unet_forward(st, *feats[:4])


def net_and_texture_forward_batch(st, texture_1cn, index_maps, training=False):
    """NetAndTexture.forward for a batch (compose.py:137-178): the net is called ONCE PER ITEM, in item order, and the results
    are concatenated — so in .train() every BatchNorm layer normalises each item with that item's own statistics (a batch of
    one) and its running buffers in `st` move once per item.  index_maps: per scale a (B,h,w) array / tensor of point ids."""
    maps = [_t(m) for m in index_maps]
    outs = []
    for b in range(maps[0].shape[0]):
        feats = [point_texture_forward(texture_1cn, m[b][None]) for m in maps[:4]]
        outs.append(unet_forward(st, *feats, training=training))
    return torch.cat(outs, 0)


def unet_forward_per_item(st, x, x2, x4, x8, training=False):
    """The same per-item loop over ready-made feature pyramids (B,8,h,w)."""
    return torch.cat([unet_forward(st, x[b:b + 1], x2[b:b + 1], x4[b:b + 1], x8[b:b + 1], training=training)
                      for b in range(x.shape[0])], 0)


def psnr(a, b):
    """src/train.py:39-48: -10*log10(mean((a-b)^2))."""
    mse = torch.mean((_t(a).double() - _t(b).double()) ** 2).item()
    return float("inf") if mse == 0 else -10.0 * float(torch.log10(torch.tensor(mse)))
