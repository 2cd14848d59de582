"""Kernel-level parity: every HIP entry point (through the C ABI) against the CPU oracle on seeded inputs."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import memory as OM
from oracle import model as M
from oracle import ops as OO
from oracle import projector as OP


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from embodied_object_detection_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def nhwc(x):  # NCHW cpu -> NHWC contiguous
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def close(a, b, rtol=2e-4, atol=2e-4):
    a = a.detach().cpu().float()
    b = b.detach().cpu().float()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e} (|ref| max {b.abs().max().item():.3e})"


# ------------------------------------------------------------------------------------------------
# implicit-GEMM conv
# ------------------------------------------------------------------------------------------------
CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (1, 20, 24, 64, 64, 1, 1, 0),
    (1, 20, 24, 64, 128, 3, 1, 1),
    (1, 21, 19, 128, 96, 3, 2, 1),
    (1, 16, 16, 256, 512, 1, 2, 0),
    (3, 14, 14, 256, 256, 3, 1, 1),
    (2, 7, 9, 32, 5, 3, 1, 1),
    (1, 10, 10, 2048, 256, 1, 1, 0),
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("tile,splitk", [(0, 0), (1, 1), (2, 1), (3, 1), (3, 3), (1, 2), (13, 1), (22, 2), (51, 1), (52, 1), (53, 1), (53, 3), (54, 1), (54, 2)])
def test_conv_matches_f_conv2d(dev, case, tile, splitk):
    from embodied_object_detection_amd import ops
    N, H, W, Cin, Cout, k, stride, pad = case
    x = rnd(N, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=(1.0 / (Cin * k * k)) ** 0.5)
    b = rnd(Cout, seed=3)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    conv = ops.Conv(w, b, stride=stride, pad=pad, device=dev)
    y = conv(nhwc(x).to(dev), N, H, W, force_tile=tile, force_splitk=splitk)
    close(nchw(y), ref)


@pytest.mark.parametrize("case", [(8, 14, 14, 256, 256, 3, 1), (1, 20, 20, 2048, 256, 1, 0), (64, 1, 1, 12544, 128, 1, 0)])
@pytest.mark.parametrize("spread", [False, True])
def test_conv_bf16x3_accuracy(dev, case, spread):
    """The three-way bf16 split is an fp32-class computation: its error against an fp64 convolution stays within 2.5x of the
    fp32-MFMA kernel's (and of torch's CPU fp32 conv), also when the input channels span six decades."""
    from embodied_object_detection_amd import ops
    N, H, W, Cin, Cout, k, pad = case
    x = rnd(N, Cin, H, W, seed=11)
    if spread:
        x = x * torch.logspace(-3, 3, Cin).view(1, Cin, 1, 1)
    w = rnd(Cout, Cin, k, k, seed=12, scale=(1.0 / (Cin * k * k)) ** 0.5)
    b = rnd(Cout, seed=13)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=pad)
    scale = ref.abs().mean().item()
    conv = ops.Conv(w, b, stride=1, pad=pad, device=dev)
    xd = nhwc(x).to(dev)
    err = {}
    for name, tile in (("fp32", 23), ("b3_128", 51), ("b3_64", 53), ("b3_256", 54)):
        y = nchw(conv(xd, N, H, W, force_tile=tile, force_splitk=1)).double().cpu()
        e = (y - ref).abs()
        err[name] = (e.max().item() / scale, e.mean().item() / scale)
    e = (F.conv2d(x, w, b, padding=pad).double() - ref).abs()
    err["cpu"] = (e.max().item() / scale, e.mean().item() / scale)
    for name in ("b3_128", "b3_64", "b3_256"):
        assert err[name][1] <= 2.5 * max(err["fp32"][1], err["cpu"][1]), err
        assert err[name][0] <= 4.0 * max(err["fp32"][0], err["cpu"][0]), err
        assert err[name][1] < 3e-6, err


@pytest.mark.parametrize("case", [(3, 14, 14, 256, 256, 3, 1, 1), (1, 21, 19, 128, 96, 3, 2, 1), (1, 16, 16, 256, 512, 1, 2, 0)])
def test_conv_bf16x3_presplit_weights_equal_on_the_fly_split(dev, case):
    """Weights split once by eod_conv_split_weights_bf16x3 feed the 256x128 kernel the same pieces it would compute itself:
    bitwise identical outputs (also with N and Cout not multiples of the tile)."""
    from embodied_object_detection_amd import ops
    N, H, W, Cin, Cout, k, stride, pad = case
    x = rnd(N, Cin, H, W, seed=41)
    w = rnd(Cout, Cin, k, k, seed=42, scale=(1.0 / (Cin * k * k)) ** 0.5)
    b = rnd(Cout, seed=43)
    conv = ops.Conv(w, b, stride=stride, pad=pad, device=dev)
    xd = nhwc(x).to(dev)
    y0 = conv(xd, N, H, W, force_tile=54, force_splitk=1, presplit=False).clone()
    y1 = conv(xd, N, H, W, force_tile=54, force_splitk=1, presplit=True).clone()
    y2 = conv(xd, N, H, W, force_tile=54, force_splitk=2, presplit=True).clone()
    assert conv.w_split is not None and torch.equal(y0, y1)
    close(nchw(y1), F.conv2d(x, w, b, stride=stride, padding=pad))
    close(y2, y1, rtol=1e-5, atol=1e-5)


def test_conv_math_mode_switch(dev):
    """eod_set_conv_math routes force_tile == 0 launches; both modes agree to fp32 noise; unknown modes are refused."""
    from embodied_object_detection_amd import ops
    x = rnd(2, 64, 16, 16, seed=21)
    w = rnd(96, 64, 3, 3, seed=22, scale=0.05)
    conv = ops.Conv(w, None, stride=1, pad=1, device=dev)
    xd = nhwc(x).to(dev)
    assert ops.get_conv_math() == "fp32"
    y0 = conv(xd, 2, 16, 16).clone()
    assert ops.set_conv_math("bf16x3") == "fp32"
    try:
        y1 = conv(xd, 2, 16, 16).clone()
        y_in = conv(xd, 2, 16, 16, in_relu=True).clone()     # in_relu stays on the fp32 kernel
    finally:
        assert ops.set_conv_math("fp32") == "bf16x3"
    assert not torch.equal(y0, y1)                             # a different kernel ran
    close(y1, y0, rtol=1e-5, atol=1e-5)
    close(nchw(y_in), F.conv2d(F.relu(x), w, None, padding=1))
    with pytest.raises(ValueError):
        ops.set_conv_math("fp16")


@pytest.mark.parametrize("scatter", [False, True])
def test_deconv_predictor_sigmoid_fused(dev, scatter):
    """out_mode 2: ConvTranspose2d(k2,s2) + ReLU + 1x1 predictor + sigmoid in one launch == the torch chain; also with a
    device-side ROI count and the unit scatter of the lazy variant."""
    from embodied_object_detection_amd import ops
    R, Cc = 7, 256
    x = rnd(R, Cc, 14, 14, seed=31)
    wd = rnd(Cc, Cc, 2, 2, seed=32, scale=0.05)
    bd = rnd(Cc, seed=33, scale=0.1)
    pw = rnd(1, Cc, 1, 1, seed=34, scale=0.1)
    pb = 0.3
    ref = torch.sigmoid(F.conv2d(F.relu(F.conv_transpose2d(x, wd, bd, stride=2)), pw, torch.tensor([pb])))[:, 0]   # [R,28,28]
    conv = ops.Conv(wd, bd, device=dev, deconv=True)
    xd = nhwc(x).to(dev)
    count = torch.tensor([5], dtype=torch.int32, device=dev)
    out = torch.full((R, 28, 28), -1.0, device=dev)
    rows = torch.tensor([6, 0, 3, 2, 5, 0, 0], dtype=torch.int32, device=dev) if scatter else None
    conv(xd, R, 14, 14, relu=True, m_count=count, m_unit=196, out=out, fuse=(pw.reshape(-1).contiguous().to(dev), pb, rows))
    got = out.cpu()
    if scatter:
        for k, u in enumerate([6, 0, 3, 2, 5]):
            close(got[u], ref[k], rtol=1e-5, atol=1e-5)
        assert bool((got[1] == -1).all()) and bool((got[4] == -1).all())        # rows nobody maps to stay untouched
    else:
        close(got[:5], ref[:5], rtol=1e-5, atol=1e-5)
        assert bool((got[5:] == -1).all())                                        # beyond the device-side count
    # and it equals the two-launch form
    mup = conv(xd, R, 14, 14, relu=True)
    two = ops.mask_predictor_sigmoid(mup, pw.reshape(-1).contiguous().to(dev), pb, R * 784, Cc, None, 784,
                                     out=torch.empty((R, 28, 28), device=dev))
    if not scatter:
        close(got[:5], two.cpu()[:5], rtol=1e-6, atol=1e-6)


def test_conv_epilogues(dev):
    from embodied_object_detection_amd import ops
    N, H, W, Cin, Cout = 1, 12, 16, 64, 96
    x = rnd(N, Cin, H, W, seed=4)
    w = rnd(Cout, Cin, 3, 3, seed=5, scale=0.05)
    b = rnd(Cout, seed=6)
    res = rnd(N, Cout, H, W, seed=7)
    conv = ops.Conv(w, b, stride=1, pad=1, device=dev)
    xd = nhwc(x).to(dev)
    # residual add + relu (bottleneck tail)
    y = conv(xd, N, H, W, res=nhwc(res).to(dev), res_mode=1, relu=True)
    close(nchw(y), F.relu(F.conv2d(x, w, b, padding=1) + res))
    # nearest x2 upsampled residual (FPN top-down, timm.py:131-133)
    res_small = rnd(N, Cout, H // 2, W // 2, seed=8)
    y = conv(xd, N, H, W, res=nhwc(res_small).to(dev), res_mode=2)
    close(nchw(y), F.conv2d(x, w, b, padding=1) + F.interpolate(res_small, scale_factor=2.0, mode="nearest"))
    # (conv + bias) * weight + res (memory fusion, timm.py:174-182), also through split-K
    for sk in (1, 3):
        y = conv(xd, N, H, W, res=nhwc(res).to(dev), res_mode=1, out_scale=5.0, force_splitk=sk, force_tile=3)
        close(nchw(y), F.conv2d(x, w, b, padding=1) * 5.0 + res, rtol=3e-4, atol=3e-4)
    # relu on the input (p7 = conv(relu(p6)), timm.py:362)
    y = conv(xd, N, H, W, in_relu=True)
    close(nchw(y), F.conv2d(F.relu(x), w, b, padding=1))


@pytest.mark.parametrize("case,wavek", [
    ((1, 20, 20, 1024, 256, 1, 1, 0), True),     # FPN lateral at 640x640: 400 rows, 32 chunks -> 4 waves x 8 chunks
    ((1, 20, 20, 256, 256, 3, 2, 1), True),      # stride-2 3x3 (P6): 72 chunks -> 8 waves
    ((1, 20, 20, 512, 512, 3, 1, 1), True),      # layer4 conv2: 144 chunks
    ((1, 10, 10, 512, 2048, 1, 1, 0), False),    # layer4 conv3: 16 chunks = 4 per wave: stays on the slab form
    ((1, 40, 40, 1024, 256, 1, 1, 0), False),    # layer3 conv1: 1600 rows: stays on the slab form
    ((1, 5, 5, 256, 256, 3, 2, 1), True),        # P7: 9 output rows
    ((3, 10, 14, 512, 96, 3, 1, 1), True),       # a batch, Cout not a multiple of 32, 144 chunks
    ((37, 1, 1, 1024, 4, 1, 1, 0), True),        # bbox_pred.2: four output channels
])
def test_conv_wavek_few_rows_deep_k(dev, case, wavek):
    """Few-row layers with a deep K split it over the WAVES of a workgroup (conv_wavek_kernel): no slabs, no reduce launch; the
    planner keeps the slab form where it measured faster.  Against F.conv2d, with every epilogue; a batch gives bitwise the rows of
    the single-image calls; forced onto the other form the results agree to fp32 noise."""
    from embodied_object_detection_amd import ops
    N, H, W, Cin, Cout, k, stride, pad = case
    x = rnd(N, Cin, H, W, seed=51)
    w = rnd(Cout, Cin, k, k, seed=52, scale=(1.0 / (Cin * k * k)) ** 0.5)
    b = rnd(Cout, seed=53)
    conv = ops.Conv(w, b, stride=stride, pad=pad, device=dev)
    xd = nhwc(x).to(dev)
    y = conv(xd, N, H, W)
    assert (conv.desc.workspace_bytes == 0) == wavek, "wave-split-K plans need no slab workspace; the slab form does"
    y_forced = conv(xd, N, H, W, force_tile=7 if Cin * k * k >= 2048 else 6).clone()       # the 32x32-tile kernel, forced
    close(y_forced, conv(xd, N, H, W), rtol=1e-5, atol=1e-5)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    close(nchw(y), ref)
    OH, OW = ref.shape[2:]
    res = rnd(N, Cout, OH, OW, seed=54)
    y = conv(xd, N, H, W, res=nhwc(res).to(dev), res_mode=1, relu=True, in_relu=True, out_scale=2.0)
    close(nchw(y), F.relu(F.conv2d(F.relu(x), w, b, stride=stride, padding=pad) * 2.0 + res), rtol=3e-4, atol=3e-4)
    # the slab path on the same problem agrees to fp32 noise (another summation order)
    y_slab = conv(xd, N, H, W, force_tile=3, force_splitk=3)
    close(y_slab, conv(xd, N, H, W), rtol=1e-5, atol=1e-5)
    if N > 1 and wavek:
        yb = conv(xd, N, H, W, plan_rows=OH * OW).clone()
        for n in range(N):
            y1 = conv(xd[n:n + 1].contiguous(), 1, H, W)
            assert torch.equal(yb[n:n + 1], y1), "batched rows must be bitwise those of the single-image call"
    if H == 1 and W == 1:
        # a device-side row count: rows beyond it are left untouched
        cnt = torch.tensor([20], dtype=torch.int32, device=dev)
        out = torch.full((N, 1, 1, Cout), 7.0, device=dev)
        conv(xd, N, 1, 1, m_count=cnt, m_unit=1, out=out)
        close(nchw(out)[:20], ref[:20])
        assert float((out[20:] - 7.0).abs().max()) == 0.0


def test_two_linear_layers_as_one_gemm_with_split_outputs(dev):
    """cls_score.linear (no ReLU) + bbox_pred.0 (ReLU) on the same input: one GEMM with stacked weights, two outputs
    (EodConvDesc.split_n) -- bitwise the two separate layers, with a device-side row count."""
    from embodied_object_detection_amd import ops
    R, cap, K = 50, 64, 1024
    x = rnd(cap, K, seed=81)
    wa, ba = rnd(512, K, seed=82, scale=0.03), rnd(512, seed=83)
    wb, bb = rnd(1024, K, seed=84, scale=0.03), rnd(1024, seed=85)
    xd = x.view(cap, 1, 1, K).to(dev)
    cnt = torch.tensor([R], dtype=torch.int32, device=dev)
    ca = ops.Conv(wa[:, :, None, None], ba, device=dev)
    cb = ops.Conv(wb[:, :, None, None], bb, device=dev)
    cm = ops.Conv(torch.cat([wa, wb])[:, :, None, None], torch.cat([ba, bb]), device=dev)
    ya = torch.zeros((cap, 1, 1, 512), device=dev); yb = torch.zeros((cap, 1, 1, 1024), device=dev)
    ca(xd, cap, 1, 1, m_count=cnt, m_unit=1, out=ya)
    cb(xd, cap, 1, 1, relu=True, m_count=cnt, m_unit=1, out=yb)
    za = torch.zeros_like(ya); zb = torch.zeros_like(yb)
    cm(xd, cap, 1, 1, relu=True, m_count=cnt, m_unit=1, out=za, split=(512, zb))
    assert torch.equal(za, ya) and torch.equal(zb, yb)
    close(za.view(cap, 512)[:R], (x @ wa.t() + ba)[:R], rtol=3e-4, atol=3e-4)
    close(zb.view(cap, 1024)[:R], F.relu(x @ wb.t() + bb)[:R], rtol=3e-4, atol=3e-4)
    assert float(za.view(cap, 512)[R:].abs().max()) == 0.0 and float(zb.view(cap, 1024)[R:].abs().max()) == 0.0


def test_conv_wavek_pyramid_mode(dev):
    """The shared-weight head over a small pyramid (one row list, per-level zero padding) also takes the wave-split-K path."""
    from embodied_object_detection_amd import ops
    shapes = [(12, 16), (6, 8), (3, 4)]
    Cin, Cout = 256, 5
    w = rnd(Cout, Cin, 3, 3, seed=61, scale=0.02)
    b = rnd(Cout, seed=62)
    xs = [rnd(1, Cin, h, ww, seed=63 + i) for i, (h, ww) in enumerate(shapes)]
    off = [0]
    for (h, ww) in shapes:
        off.append(off[-1] + h * ww)
    rows = torch.cat([nhwc(x).reshape(-1, Cin) for x in xs]).contiguous().to(dev)
    conv = ops.Conv(w, b, pad=1, device=dev)
    out = torch.zeros((off[-1], Cout), device=dev)
    conv(rows, 1, 0, 0, out=out, levels=(off, shapes))
    assert conv.desc.workspace_bytes == 0
    for i, (h, ww) in enumerate(shapes):
        ref = F.conv2d(xs[i], w, b, padding=1)
        close(out[off[i]:off[i + 1]].view(1, h, ww, Cout).permute(0, 3, 1, 2), ref)


def test_conv_two_chunk_prefetch_is_bitwise_the_same(dev):
    """EodConvDesc.prefetch2 changes when operands are fetched, not what is computed: the mask-head shape with a device-side ROI
    count, odd and even chunk counts per split."""
    from embodied_object_detection_amd import ops
    x = rnd(40, 256, 14, 14, seed=91)
    w = rnd(256, 256, 3, 3, seed=92, scale=0.02)
    conv = ops.Conv(w, rnd(256, seed=93), pad=1, device=dev)
    xd = nhwc(x).to(dev)
    cnt = torch.tensor([33], dtype=torch.int32, device=dev)
    for splitk in (1, 3, 8):            # 72 chunks: 72 / 24 / 9 per split
        outs = []
        for pf in (0, 1, 2):           # one LDS buffer | two chunks of register prefetch | double-buffered LDS
            conv.prefetch2 = pf
            y = torch.zeros((40, 14, 14, 256), device=dev)
            conv(xd, 40, 14, 14, relu=True, m_count=cnt, m_unit=196, out=y, force_tile=13, force_splitk=splitk)
            outs.append(y)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), splitk
    conv.prefetch2 = 2
    close(nchw(conv(xd, 40, 14, 14, relu=True)), F.relu(F.conv2d(x, w, conv.bias.cpu(), padding=1)))


def test_conv_stem_tap4_with_bn_fold(dev):
    from embodied_object_detection_amd import ops
    H, W = 64, 96
    x = rnd(1, 3, H, W, seed=9)
    w = rnd(64, 3, 7, 7, seed=10, scale=0.1)
    g = torch.rand(64) + 0.5
    beta = rnd(64, seed=11, scale=0.1)
    mean = rnd(64, seed=12, scale=0.1)
    var = torch.rand(64) + 0.5
    ref = F.relu(F.batch_norm(F.conv2d(x, w, stride=2, padding=3), mean, var, g, beta, training=False, eps=1e-5))
    wf, bf = ops.fold_bn(w, g, beta, mean, var)
    conv = ops.Conv(wf, bf, stride=2, pad=3, device=dev, cin_pad=4)
    x4 = torch.zeros((1, H, W, 4))
    x4[..., :3] = nhwc(x)
    y = conv(x4.to(dev), 1, H, W, relu=True)
    close(nchw(y), ref)


def test_linear_as_conv_and_dynamic_rows(dev):
    from embodied_object_detection_amd import ops
    R, K, O = 37, 12544, 1024
    x = rnd(R, K, seed=13)
    w = rnd(O, K, seed=14, scale=(1.0 / K) ** 0.5)
    b = rnd(O, seed=15)
    fc = ops.Conv(w.view(O, K, 1, 1), b, device=dev)
    cnt = torch.tensor([29], dtype=torch.int32, device=dev)
    out = torch.full((R, 1, 1, O), -7.0, device=dev)
    fc(x.to(dev), R, 1, 1, relu=True, m_count=cnt, m_unit=1, out=out)
    ref = F.relu(F.linear(x, w, b))
    close(out.view(R, O)[:29], ref[:29])
    assert bool((out.view(R, O)[29:] == -7.0).all()), "rows beyond the device-side count must not be written"


def test_deconv2x2(dev):
    from embodied_object_detection_amd import ops
    R, Cc = 5, 256
    x = rnd(R, Cc, 14, 14, seed=16)
    w = rnd(Cc, Cc, 2, 2, seed=17, scale=0.05)
    b = rnd(Cc, seed=18)
    dc = ops.Conv(w, b, device=dev, deconv=True)
    y = dc(nhwc(x).to(dev), R, 14, 14, relu=True)
    assert tuple(y.shape) == (R, 28, 28, Cc)
    close(nchw(y), F.relu(F.conv_transpose2d(x, w, b, stride=2)))


def test_conv_rejects_bad_descriptors(dev):
    from embodied_object_detection_amd import _lib, ops
    w = rnd(8, 48, 1, 1)
    with pytest.raises(ValueError):
        ops.Conv(w, None, device=dev)  # Cin not a multiple of 32
    conv = ops.Conv(rnd(8, 32, 1, 1), None, device=dev)
    with pytest.raises(_lib.EodError):
        conv(torch.zeros((1, 4, 4, 32)), 1, 4, 4)  # host tensor: no CPU fallback


# ------------------------------------------------------------------------------------------------
# elementwise
# ------------------------------------------------------------------------------------------------
def test_preprocess_maxpool(dev):
    from embodied_object_detection_amd import ops
    cfg = M.OracleCfg()
    img = torch.randint(0, 256, (3, 50, 70), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    ref = M.preprocess_image(img, cfg)
    out, Hp, Wp = ops.preprocess_image(img.to(dev), cfg.pixel_mean, cfg.pixel_std)
    assert (Hp, Wp) == (64, 96)
    close(nchw(out)[:, :3], ref, rtol=1e-6, atol=1e-6)
    assert float(out[..., 3].abs().max()) == 0.0
    x = rnd(2, 64, 33, 41, seed=2)
    y, OH, OW = ops.maxpool3x3s2(nhwc(x).to(dev), 2, 33, 41, 64)
    close(nchw(y), F.max_pool2d(x, 3, 2, 1), rtol=0, atol=0)


def test_groupnorm_relu_multilevel(dev):
    from embodied_object_detection_amd import ops
    hw = [(8, 12), (4, 6), (2, 3), (1, 2), (1, 1)]
    Cc = 256
    xs = [rnd(1, Cc, h, w, seed=10 + i, scale=2.0) + 0.3 for i, (h, w) in enumerate(hw)]
    gamma = torch.rand(Cc) + 0.5
    beta = rnd(Cc, seed=20, scale=0.2)
    flat = torch.cat([nhwc(x).reshape(-1, Cc) for x in xs]).to(dev)
    off = [0]
    for h, w in hw:
        off.append(off[-1] + h * w)
    stats = ops.groupnorm_workspace(off, dev)
    y = ops.groupnorm_relu(flat, gamma.to(dev), beta.to(dev), off, Cc, stats).cpu()
    for i, x in enumerate(xs):
        ref = F.relu(F.group_norm(x, 32, gamma, beta, eps=1e-5))
        close(y[off[i]:off[i + 1]], nhwc(ref).reshape(-1, Cc), rtol=1e-4, atol=1e-4)


def test_groupnorm_statistics_ride_on_the_conv_slab_reduce(dev):
    """A pyramid-mode conv whose plan reduces split-K slabs (the CenterNet tower at 640x640: 536 tiles) also writes GroupNorm's
    partial sums: the conv output is bitwise that of the plain reduce; the statistics are the same double sums in another
    association (per-thread row subsets combined through LDS), so the GroupNorm output agrees to the last float bits."""
    from embodied_object_detection_amd import ops
    hw = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]
    Cc = 256
    off = [0]
    for h, w in hw:
        off.append(off[-1] + h * w)
    x = rnd(off[-1], Cc, seed=71).to(dev)
    w = rnd(Cc, Cc, 3, 3, seed=72, scale=0.02)
    b = rnd(Cc, seed=73)
    gamma, beta = (torch.rand(Cc) + 0.5).to(dev), rnd(Cc, seed=74, scale=0.2).to(dev)
    conv = ops.Conv(w, b, pad=1, device=dev)
    stats_a, stats_b = ops.groupnorm_workspace(off, dev), ops.groupnorm_workspace(off, dev)
    ya = torch.empty((off[-1], Cc), device=dev)
    conv(x, 1, 0, 0, out=ya, levels=(off, hw))
    assert not conv.gn_fused
    ga = ops.groupnorm_relu(ya, gamma, beta, off, Cc, stats_a)
    yb = torch.empty((off[-1], Cc), device=dev)
    conv(x, 1, 0, 0, out=yb, levels=(off, hw), gn_stats=stats_b)
    assert conv.gn_fused, "this layer's plan has a slab reduce"
    gb = ops.groupnorm_relu(yb, gamma, beta, off, Cc, stats_b, partial_ready=True)
    assert torch.equal(ya, yb)
    close(gb, ga, rtol=2e-6, atol=2e-6)
    assert float((ga == gb).float().mean()) > 0.999
    # a small pyramid has no slab reduce to ride on: the statistics launch stays
    hs = [(8, 12), (4, 6)]
    offs = [0, 96, 120]
    conv(x[:120].contiguous(), 1, 0, 0, out=torch.empty((120, Cc), device=dev), levels=(offs, hs), gn_stats=ops.groupnorm_workspace(offs, dev))
    assert not conv.gn_fused


def test_mask_predictor(dev):
    from embodied_object_detection_amd import ops
    rows, Cc = 3 * 784, 256
    x = rnd(rows, Cc, seed=30)
    w = rnd(Cc, seed=31, scale=0.1)
    cnt = torch.tensor([2], dtype=torch.int32, device=dev)
    out = torch.full((rows,), -1.0, device=dev)
    ops.mask_predictor_sigmoid(x.to(dev), w.to(dev), 0.25, rows, Cc, cnt, 784, out=out)
    ref = torch.sigmoid(x @ w + 0.25)
    close(out[:2 * 784], ref[:2 * 784], rtol=1e-5, atol=1e-5)
    assert bool((out[2 * 784:] == -1.0).all())


# ------------------------------------------------------------------------------------------------
# ROIAlign
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S", [7, 14])
def test_roi_align_matches_oracle(dev, S):
    from embodied_object_detection_amd import ops
    h3, w3, Cc = 16, 24, 256
    feats = [rnd(1, Cc, h3 >> i, w3 >> i, seed=40 + i) for i in range(3)]
    g = torch.Generator().manual_seed(5)
    R = 24
    ctr = torch.rand((R, 2), generator=g) * torch.tensor([w3 * 8.0, h3 * 8.0])
    size = torch.exp(torch.rand((R, 2), generator=g) * 4.0 + 1.0)   # ~3 .. 150 px (all three levels)
    boxes = torch.cat([ctr - size / 2, ctr + size / 2], dim=1)
    boxes[0] = torch.tensor([-20.0, -30.0, 60.0, 50.0])             # sticks out of the image
    boxes[1] = torch.tensor([10.0, 10.0, 10.01, 10.01])             # minimum-size proposal
    ref = OO.roi_pool(feats, boxes, S)
    cnt = torch.tensor([R - 2], dtype=torch.int32, device=dev)
    out = ops.roi_align(*[nhwc(f).to(dev) for f in feats], h3, w3, Cc, boxes.to(dev), cnt, R, S)
    close(nchw(out)[:R - 2], ref[:R - 2], rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------------------------------------
# selection
# ------------------------------------------------------------------------------------------------
def _head_case(level_hw, seed):
    g = torch.Generator().manual_seed(seed)
    agn, reg = [], []
    for (h, w) in level_hw:
        agn.append(torch.randn((1, 1, h, w), generator=g) * 1.5 - 1.0)
        reg.append(torch.randn((1, 4, h, w), generator=g) * 2.0 + 3.0)
    return agn, reg


@pytest.mark.parametrize("level_hw,topk,post", [
    ([(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)], 1000, 256),
    ([(40, 48), (20, 24), (10, 12), (5, 6), (3, 3)], 1000, 256),   # level 0 > pre-NMS top-k
    ([(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)], 100, 64),
])
def test_centernet_proposals_match_oracle(dev, level_hw, topk, post):
    from embodied_object_detection_amd import ops
    agn, reg = _head_case(level_hw, seed=3)
    scales = [0.9, 1.0, 1.1, 1.2, 0.8]
    cfg = M.OracleCfg(pre_nms_topk=topk, post_nms_topk=post)
    # oracle consumes relu(scale * bbox_pred)
    reg_o = [F.relu(r * s) for r, s in zip(reg, scales)]
    rb, rs = M.centernet_proposals(agn, reg_o, cfg)
    head = torch.cat([torch.cat([nhwc(a).reshape(-1, 1), nhwc(r).reshape(-1, 4)], dim=1) for a, r in zip(agn, reg)]).contiguous()
    dec = ops.ProposalDecoder(level_hw, M.FPN_STRIDES, scales, cfg.inference_th, topk, post, cfg.nms_th_proposal, cap=post + 64, device=dev)
    b, s, c = dec(head.to(dev))
    n = int(c.item())
    assert n == rb.shape[0]
    close(s[:n], rs, rtol=1e-6, atol=1e-6)
    close(b[:n], rb, rtol=1e-5, atol=1e-4)


def _pyramid_hw(H, W):
    return [((H + s - 1) // s, (W + s - 1) // s) for s in M.FPN_STRIDES]


@pytest.mark.parametrize("hw,nms,variant", [((640, 640), 0.9, "plain"), ((640, 640), 0.5, "plain"), ((640, 640), 0.9, "ties"),
                                            ((960, 960), 0.9, "plain"), ((512, 640), 0.9, "golden"), ((640, 640), 0.15, "plain"),
                                            ((64, 96), 0.9, "small"), ((128, 128), 0.3, "small")])
def test_centernet_train_proposals_match_oracle(dev, hw, nms, variant, golden_dir):
    """The TRAINING proposal lists (PRE / POST_NMS_TOPK_TRAIN 4000 / 2000, NMS_TH_TRAIN 0.9; centernet.py:214-219 ->
    predict_instances / nms_and_topK with `self.training`): the wide path of `eod_centernet_proposals` (rank merge, suppression bit
    matrix, scanning workgroup) against the oracle's decode of the same head outputs -- same proposals in the same order.  NMS 0.5
    makes the suppression matter (a third of the candidates goes), `ties` quantises the logits (long runs of equal scores: the
    rank merge's run walk and the '>= kth' rule), 960x960 has 14 400 positions on the finest level and 8 789 merged candidates;
    `golden` is the case the reference's own `CenterNet` decoded in training mode (tests/golden/centernet_decode_train.npz)."""
    from embodied_object_detection_amd import ops
    pre, post = 4000, 2000
    if variant == "golden":
        import _inputs as I
        agn, reg = I.centernet_decode_train_case()
        level_hw = [tuple(a.shape[2:]) for a in agn]
        scales = [1.0] * 5
        gd = np.load(os.path.join(golden_dir, "centernet_decode_train.npz"))
        pre, post = int(gd["pre_post"][0]), int(gd["pre_post"][1])
    else:
        level_hw = _pyramid_hw(*hw)
        agn, reg = _head_case(level_hw, seed=17)
        scales = [0.9, 1.0, 1.1, 1.2, 0.8]
        if variant == "ties":
            agn = [torch.round(a * 4.0) / 4.0 for a in agn]
            reg = [r * 0.1 for r in reg]
    cfg = M.OracleCfg(pre_nms_topk=pre, post_nms_topk=post, nms_th_proposal=nms)
    rb, rs = M.centernet_proposals(agn, [F.relu(r * s) for r, s in zip(reg, scales)], cfg)
    if variant == "golden":
        # bit for bit on the CPU that wrote the fixture (tests/test_oracle_golden.py); this host's sigmoid / sqrt may differ by an ulp,
        # which can exchange two neighbours of the list: the same scores, and every box of the fixture in the list
        np.testing.assert_allclose(rs.numpy(), gd["scores"], rtol=2e-7, atol=0)
        gb = torch.from_numpy(gd["boxes"])
        assert float((gb[:, None, :] - rb[None, :, :]).abs().amax(dim=2).min(dim=1).values.max()) <= 1e-4
    head = torch.cat([torch.cat([nhwc(a).reshape(-1, 1), nhwc(r).reshape(-1, 4)], dim=1) for a, r in zip(agn, reg)]).contiguous()
    cap = 4096 if variant == "ties" else post + 48
    dec = ops.ProposalDecoder(level_hw, M.FPN_STRIDES, scales, cfg.inference_th, pre, post, nms, cap=cap, device=dev)
    b, s, c = dec(head.to(dev))
    n = int(c.item())
    assert n == rb.shape[0], (n, rb.shape[0])
    if variant == "small":
        # fewer candidates than POST_NMS_TOPK_TRAIN (and than one 64-entry chunk / a few chunks): everything NMS leaves is kept
        assert 0 < n < post
    elif nms >= 0.5:
        assert n >= post and (variant != "ties" or n > post)
    else:
        assert 0 < n < post              # NMS 0.15 removes most of the 6 000 candidates: the walk runs to the end of the list
    if variant == "ties":
        # the order among EQUAL scores: position order in both, so the lists still agree entry by entry
        assert n > post + 48
    close(s[:n], rs, rtol=1e-6, atol=1e-6)
    if variant == "golden":
        # against the reference's own list, as sets (see above)
        assert float((gb[:, None, :] - b[:n].cpu()[None, :, :]).abs().amax(dim=2).min(dim=1).values.max()) <= 1e-4
    close(b[:n], rb, rtol=1e-5, atol=1e-4)
    # a second call on the same decoder (workspace reuse) gives the same list
    b2, s2, c2 = dec(head.to(dev))
    assert int(c2.item()) == n and torch.equal(b2[:n], b[:n])


def test_centernet_train_proposals_batch_of_two_equals_single_scenes(dev):
    """The wide path with `batch` = 2 (every buffer = two single-scene buffers back to back, the head rows level major over the
    scenes): each scene's list is bitwise the list of its own single-scene call."""
    from embodied_object_detection_amd import ops
    level_hw = _pyramid_hw(384, 512)
    scales = [0.9, 1.0, 1.1, 1.2, 0.8]
    pre, post, cap = 4000, 2000, 2048
    heads, singles = [], []
    for b in range(2):
        agn, reg = _head_case(level_hw, seed=40 + b)
        head = torch.cat([torch.cat([nhwc(a).reshape(-1, 1), nhwc(r).reshape(-1, 4)], dim=1) for a, r in zip(agn, reg)]).contiguous()
        heads.append(head)
        dec = ops.ProposalDecoder(level_hw, M.FPN_STRIDES, scales, 1e-4, pre, post, 0.9, cap=cap, device=dev)
        bx, sc, c = dec(head.to(dev))
        n = int(c.item())
        singles.append((bx[:n].clone(), sc[:n].clone(), n))
    off = [0]
    for (h, w) in level_hw:
        off.append(off[-1] + h * w)
    both = torch.cat([torch.cat([heads[b][off[l]:off[l + 1]] for b in range(2)]) for l in range(5)]).contiguous()
    dec2 = ops.ProposalDecoder(level_hw, M.FPN_STRIDES, scales, 1e-4, pre, post, 0.9, cap=cap, device=dev, batch=2)
    bx, sc, c = dec2(both.to(dev))
    for b in range(2):
        n = int(c[b].item())
        assert n == singles[b][2] and n >= post
        assert torch.equal(bx[b * cap:b * cap + n], singles[b][0]) and torch.equal(sc[b * cap:b * cap + n], singles[b][1])


def test_centernet_proposals_keep_ties(dev):
    from embodied_object_detection_amd import ops
    level_hw = [(12, 12), (6, 6), (3, 3), (2, 2), (1, 1)]
    agn = [torch.full((1, 1, h, w), 0.5) for h, w in level_hw]      # every score identical
    reg = [torch.full((1, 4, h, w), 0.2) for h, w in level_hw]      # tiny boxes: NMS removes nothing
    cfg = M.OracleCfg(post_nms_topk=32)
    rb, rs = M.centernet_proposals(agn, [F.relu(r) for r in reg], cfg)
    assert rb.shape[0] == 144 + 36 + 9 + 4 + 1                      # '>= kth' keeps every tie
    head = torch.cat([torch.cat([nhwc(a).reshape(-1, 1), nhwc(r).reshape(-1, 4)], dim=1) for a, r in zip(agn, reg)]).contiguous()
    dec = ops.ProposalDecoder(level_hw, M.FPN_STRIDES, [1.0] * 5, cfg.inference_th, 1000, 32, 0.9, cap=256, device=dev)
    b, s, c = dec(head.to(dev))
    assert int(c.item()) == rb.shape[0]
    close(b[:rb.shape[0]], rb, rtol=0, atol=1e-5)


@pytest.mark.parametrize("thresh,topk,R,spread", [(0.02, 300, 256, 1.0), (0.3, 100, 256, 1.0), (0.6, 50, 40, 1.0), (0.999, 10, 16, 1.0),
                                                  (0.02, 300, 256, 0.02), (0.02, 300, 320, 0.15), (0.0, 100, 300, 0.3)])
def test_fast_rcnn_inference_matches_oracle(dev, thresh, topk, R, spread):
    """`spread` < 1 packs the boxes on top of each other: per-class NMS then suppresses nearly everything, the best 1024 candidates
    do not fill the list and the kernel has to go on into its second sorted batch."""
    from embodied_object_detection_amd import ops
    g = torch.Generator().manual_seed(11)
    ctr = torch.tensor([100.0, 75.0]) + (torch.rand((R, 2), generator=g) - 0.5) * torch.tensor([200.0, 150.0]) * spread
    size = torch.rand((R, 2), generator=g) * 80 * min(1.0, spread * 4) + (4 if spread == 1.0 else 60)
    boxes = torch.cat([ctr - size / 2, ctr + size / 2], dim=1)
    scores = torch.rand((R, 21), generator=g)
    scores[3, 5] = float("nan")                                     # non-finite rows are dropped
    boxes[7, 2] = float("inf")
    rb, rs, rc, rr = OO.fast_rcnn_inference_single(boxes, scores, (150, 200), thresh, 0.5, topk)
    cap = 320
    bp = torch.zeros((cap, 4)); bp[:R] = boxes
    sp = torch.zeros((cap, 21)); sp[:R] = scores
    sel = ops.DetectionSelector(cap, 21, topk, dev)
    cnt = torch.tensor([R], dtype=torch.int32, device=dev)
    b, s, c, r, n = sel(bp.to(dev), sp.to(dev), cnt, 200.0, 150.0, thresh, 0.5)
    n = int(n.item())
    assert n == rb.shape[0]
    if n:
        close(s[:n], rs, rtol=0, atol=0)
        close(b[:n], rb, rtol=0, atol=0)
        assert torch.equal(c[:n].cpu().long(), rc.long())
        assert torch.equal(r[:n].cpu().long(), rr.long())
    # the same launch can also write torch.unique of the kept rows (custom_rcnn.py:875); a second call reuses the buffers
    selu = ops.DetectionSelector(cap, 21, topk, dev, unique=True)
    for _ in range(2):
        b2, s2, c2, r2, n2 = selu(bp.to(dev), sp.to(dev), cnt, 200.0, 150.0, thresh, 0.5)
        assert int(n2.item()) == n and torch.equal(r2[:n], r[:n]) and torch.equal(b2[:n], b[:n])
        u = torch.unique(rr.long())
        assert int(selu.uniq_count.item()) == u.numel() and torch.equal(selu.uniq_rows[:u.numel()].cpu().long(), u)


def test_box_head_glue(dev):
    from embodied_object_detection_amd import ops
    from embodied_object_detection_amd.checkpoint import load_zs_weight
    R, cap = 50, 64
    zs = load_zs_weight()
    feat = rnd(cap, 512, seed=21)
    cnt = torch.tensor([R], dtype=torch.int32, device=dev)
    prob = torch.zeros((cap, 21), device=dev)
    featn = torch.zeros((cap, 512), device=dev)
    ops.zs_classify(feat.to(dev), zs.to(dev), prob, False, featn, cnt, cap, 21)
    ops.zs_classify(feat.to(dev), zs.to(dev), prob, True, None, cnt, cap, 21)
    xn = 50.0 * F.normalize(feat, p=2, dim=1)
    ref = torch.sigmoid(xn @ zs)
    close(featn[:R], xn[:R], rtol=1e-5, atol=1e-5)
    close(prob[:R], 2 * ref[:R], rtol=1e-5, atol=1e-5)
    ps = torch.rand(cap)
    ops.cascade_scores(prob, ps.to(dev), cnt, cap, 21, 0.5)
    close(prob[:R], (ref[:R] * ps[:R, None]) ** 0.5, rtol=1e-5, atol=1e-6)
    ms = torch.zeros((cap, 21), device=dev)
    ps[4] = 1.0
    ops.memory_scores(featn, zs.to(dev), ps.to(dev), ms, cnt, cap, 21)
    ref_ms = (torch.sigmoid(xn @ zs) * ps[:, None]) ** 0.5
    ref_ms[4] = 0
    close(ms[:R], ref_ms[:R], rtol=1e-5, atol=1e-6)
    # the same two tails fused into the classifier launch (stage 0: memory re-score; last stage: cascade score fusion): bitwise
    # what the separate launches give
    zs2 = (zs * 0.5 + 0.01).contiguous()                               # a second class matrix (the meta-architecture's own)
    prob_f = torch.zeros((cap, 21), device=dev)
    featn_f = torch.zeros((cap, 512), device=dev)
    ms_f = torch.zeros((cap, 21), device=dev)
    ops.zs_classify(feat.to(dev), zs.to(dev), prob_f, False, featn_f, cnt, cap, 21, zs_mem=zs2.to(dev), prop_scores=ps.to(dev),
                    mem_scores_out=ms_f)
    ms_s = torch.zeros((cap, 21), device=dev)
    ops.memory_scores(featn_f, zs2.to(dev), ps.to(dev), ms_s, cnt, cap, 21)
    assert torch.equal(featn_f, featn) and torch.equal(ms_f[:R], ms_s[:R]) and float(ms_f[4].abs().max()) == 0.0
    ops.zs_classify(feat.to(dev), zs.to(dev), prob_f, True, None, cnt, cap, 21, prop_scores=ps.to(dev), final_inv_stages=0.5)
    prob_s = torch.zeros((cap, 21), device=dev)
    ops.zs_classify(feat.to(dev), zs.to(dev), prob_s, False, None, cnt, cap, 21)
    ops.zs_classify(feat.to(dev), zs.to(dev), prob_s, True, None, cnt, cap, 21)
    ops.cascade_scores(prob_s, ps.to(dev), cnt, cap, 21, 0.5)
    assert torch.equal(prob_f[:R], prob_s[:R])
    from embodied_object_detection_amd._lib import EodError
    with pytest.raises(EodError, match="classes"):                     # a vocabulary the LDS-staged class matrix cannot hold
        ops.zs_classify(feat.to(dev), torch.zeros((512, 30), device=dev), torch.zeros((cap, 30), device=dev), False, None, cnt, cap, 30)
    # apply_deltas + clip
    boxes = torch.rand((cap, 4)) * 100
    boxes[:, 2:] += boxes[:, :2] + 1
    deltas = rnd(cap, 4, seed=22, scale=3.0)
    deltas[0, 2] = 100.0                                            # hits the log(1000/16) clamp
    out = torch.zeros((cap, 4), device=dev)
    ops.apply_deltas(deltas.to(dev), 4, boxes.to(dev), out, cnt, cap, (10.0, 10.0, 5.0, 5.0), True, 160.0, 120.0)
    ref = OO.clip_boxes(OO.apply_deltas(deltas, boxes, (10.0, 10.0, 5.0, 5.0)), (120, 160))
    close(out[:R], ref[:R], rtol=1e-5, atol=1e-4)


def test_postprocess_and_paste_masks(dev):
    from embodied_object_detection_amd import ops
    K, H, W = 9, 96, 128
    g = torch.Generator().manual_seed(31)
    masks = torch.rand((K, 28, 28), generator=g)
    ctr = torch.rand((K, 2), generator=g) * torch.tensor([float(W), float(H)])
    size = torch.rand((K, 2), generator=g) * 60 + 3
    boxes = torch.cat([ctr - size / 2, ctr + size / 2], dim=1)
    boxes[2] = torch.tensor([-30.0, 10.0, -5.0, 40.0])             # fully outside -> empty after clip -> dropped
    boxes = OO.clip_boxes(boxes, (H, W))
    scores = torch.rand(K, generator=g)
    classes = torch.randint(0, 20, (K,), generator=g)
    ref = M.detector_postprocess(boxes, scores, classes, masks[:, None], (H, W), (H, W), M.OracleCfg())
    cap = 16
    bp = torch.zeros((cap, 4)); bp[:K] = boxes
    sp = torch.zeros(cap); sp[:K] = scores
    cp = torch.zeros(cap, dtype=torch.int32); cp[:K] = classes.int()
    ob = torch.zeros((cap, 4), device=dev); os_ = torch.zeros(cap, device=dev)
    oc = torch.zeros(cap, dtype=torch.int32, device=dev); osrc = torch.zeros(cap, dtype=torch.int32, device=dev)
    ocnt = torch.zeros(1, dtype=torch.int32, device=dev)
    cnt = torch.tensor([K], dtype=torch.int32, device=dev)
    ops.detector_postprocess(bp.to(dev), sp.to(dev), cp.to(dev), cnt, cap, 1.0, 1.0, float(W), float(H), ob, os_, oc, osrc, ocnt)
    n = int(ocnt.item())
    assert n == ref["pred_boxes"].shape[0] == K - 1
    close(ob[:n], ref["pred_boxes"], rtol=0, atol=0)
    mp = torch.zeros((cap, 28, 28)); mp[:K] = masks
    out = torch.zeros((cap, H, W), dtype=torch.uint8, device=dev)
    ops.paste_masks(mp.to(dev), ob, osrc, ocnt, cap, H, W, 0.5, out)
    got = out[:n].cpu().bool()
    mism = (got != ref["pred_masks"]).float().mean().item()
    assert mism < 1e-4, f"pasted mask mismatch fraction {mism}"


# ------------------------------------------------------------------------------------------------
# spatial memory
# ------------------------------------------------------------------------------------------------
def test_unproject_grid_index_bit_exact(dev):
    from embodied_object_detection_amd import ops
    rng = np.random.RandomState(3)
    H, W = 96, 160
    depth = np.clip(3 + rng.randn(H, W) * 1.5, 0.0, 10).astype(np.float32)
    depth[0, :5] = 0.0
    T = OP.transform3d([2.0, 1.5, 2.5, 0.7, math.pi])
    intr = OP.intrinsics_from_vfov(W, H, 67.5 * math.pi / 180)
    for order, (mw, mh) in ((0, (50, 40)), (1, (200, 200))):
        xyz_ref = OP.unproject_world(depth, T, *intr, proj_shift=(0.1, 0.0, -0.2))
        idx_ref = OP.grid_index(xyz_ref, (-5, 0, -5), 0.2, mw, mh, order)
        idx, xyz = ops.unproject_grid_index(torch.from_numpy(depth).to(dev), T, intr, (0.1, 0.0, -0.2), (-5, 0, -5), 0.2, mw, mh, order,
                                            want_xyz=True)
        assert np.array_equal(xyz.cpu().numpy(), xyz_ref), "world xyz must be bit-identical to oracle/projector.c"
        assert np.array_equal(idx.cpu().numpy(), idx_ref), "grid-cell indices must be bit-exact"


def _pooled_rows(pooled_ref):
    """oracle's three [1,512,h,w] fp16 tensors -> one [h8*w8 + h16*w16 + h32*w32, 512] row list"""
    return torch.cat([nhwc(r.float()).reshape(-1, 512) for r in pooled_ref])


def _fragments_to_rows(buf, H, W):
    """Kernel layout (include/eod_hip.h eod_memory_gather_pool): [level][32-row tile][k-step][hi][r][8] -> row-major rows per level."""
    out, t0 = [], 0
    flat = buf.reshape(-1)
    for s in (8, 16, 32):
        rows = (H // s) * (W // s)
        tiles = (rows + 31) // 32
        blk = flat[t0 * 32 * 512:(t0 + tiles) * 32 * 512].view(tiles, 32, 2, 32, 8)
        out.append(blk.permute(0, 3, 1, 2, 4).reshape(tiles * 32, 512)[:rows])
        t0 += tiles
    return torch.cat(out)


def _rows_to_fragments(level_rows):
    """inverse of `_fragments_to_rows` for a list of per-level [rows, 512] tensors (padding rows are filled with NaN: they must
    never reach a stored result)"""
    out = []
    for rws in level_rows:
        rows = rws.shape[0]
        tiles = (rows + 31) // 32
        pad = torch.full((tiles * 32, 512), float("nan"), dtype=rws.dtype)
        pad[:rows] = rws
        out.append(pad.view(tiles, 32, 32, 2, 8).permute(0, 2, 3, 1, 4).reshape(-1))
    return torch.cat(out).view(-1, 512)


@pytest.mark.parametrize("pattern", ["blocky", "distinct", "constant", "columns"])
def test_memory_read_matches_oracle(dev, pattern):
    """a4 + a8 gather / cascaded pooling.  `blocky`: runs of equal cells with 30 % noise (LDS row cache + mixed blocks);
    `distinct`: every pixel its own cell (1 024 distinct per tile: hash overflow -> direct reads); `constant`: one cell (every 4x4
    block takes the n*v shortcut); `columns`: one cell per pixel column (a wall seen at constant depth: 32 distinct per tile)."""
    from embodied_object_detection_amd import ops
    g = torch.Generator().manual_seed(7)
    N, H, W = (8000 if pattern == "distinct" else 500), 64, 96
    mem = torch.randn((N, 512), generator=g) * 30
    mem[::5] *= 1e-3                                  # wide exponent spread inside the pooling windows
    mem[::7] *= 1e-6                                  # fp16 subnormals
    obs = torch.randint(0, 6, (N,), generator=g).float()
    if pattern == "blocky":
        proj = torch.randint(0, N, (H, W), generator=g)
        blocky = ((torch.arange(H)[:, None] // 6) * 17 + (torch.arange(W)[None, :] // 9)) % N
        proj = torch.where(torch.rand((H, W), generator=g) < 0.7, blocky, proj)
    elif pattern == "distinct":
        proj = torch.randperm(N, generator=g)[:H * W].reshape(H, W)
    elif pattern == "constant":
        proj = torch.full((H, W), 123, dtype=torch.int64)
    else:
        proj = (torch.arange(W)[None, :] * 5 % N).expand(H, W).contiguous()
    ref_norm = OM.create_implicit_memory(mem, obs).to(torch.half)
    m16 = ops.memory_normalize_f16(mem.to(dev), obs.to(dev))
    assert torch.equal(m16.cpu(), ref_norm), "obs-normalised fp16 memory must be bit-exact"
    ref = _pooled_rows(M.memory_read_pooled(ref_norm, proj))
    err = torch.zeros((1,), dtype=torch.int32, device=dev)
    for torch_order in (True, False):
        raw = ops.memory_gather_pool(m16, proj.int().to(dev), H, W, err=err, torch_order=torch_order).cpu()
        assert raw.shape == (ops.pooled_rows(H, W), 512)
        out = _fragments_to_rows(raw, H, W).float()
        assert out.shape == ref.shape
        same = (out == ref).float().mean().item()
        print(f"[memory read / {pattern} / torch_order={torch_order}] identical fp16 values: {same:.6f}")
        if torch_order:
            assert same == 1.0, "pixel-by-pixel pooling order must be bit-identical to F.avg_pool2d"
        else:
            # per-cell block sums: same numbers, fewer roundings; this input spreads 2^20 inside the windows on purpose
            assert same > 0.9995, f"fp16 pooled values differ on {1 - same:.2e} of elements"
        close(out, ref, rtol=1e-3, atol=1e-3)
    assert int(err.item()) == 0
    # with <= 9 bits of exponent spread inside a window the sequential sum is exact, so both orders give the same bits
    tame = (torch.rand((N, 512), generator=g) * 3 + 1) * torch.where(torch.rand((N, 512), generator=g) < 0.5, -1.0, 1.0)
    t16 = tame.half().to(dev)
    a = ops.memory_gather_pool(t16, proj.int().to(dev), H, W, torch_order=True).cpu()
    b = ops.memory_gather_pool(t16, proj.int().to(dev), H, W, torch_order=False).cpu()
    assert torch.equal(_fragments_to_rows(a, H, W), _fragments_to_rows(b, H, W))


def test_memory_read_flags_out_of_range_indices(dev):
    """proj_indices written for another map size: clamped to the table, flagged in the device error word, no fault."""
    from embodied_object_detection_amd import ops
    g = torch.Generator().manual_seed(8)
    N, H, W = 300, 32, 64
    m16 = (torch.randn((N, 512), generator=g) * 10).half()
    proj = torch.randint(0, N, (H, W), generator=g)
    bad = proj.clone()
    bad[3, 5], bad[20, 40] = N + 7, -2
    fixed = proj.clone()
    fixed[3, 5], fixed[20, 40] = N - 1, 0
    err = torch.zeros((1,), dtype=torch.int32, device=dev)
    a = ops.memory_gather_pool(m16.to(dev), bad.int().to(dev), H, W, err=err).cpu()
    assert int(err.item()) == 1
    err.zero_()
    b = ops.memory_gather_pool(m16.to(dev), fixed.int().to(dev), H, W, err=err).cpu()
    assert int(err.item()) == 0
    assert torch.equal(_fragments_to_rows(a, H, W), _fragments_to_rows(b, H, W))


def test_memory_normalize_dirty_keeps_the_fp16_table_current(dev):
    """Incremental a4: after changing rows and observation counts and flagging them, the resident table equals a full re-normalise;
    the flags are consumed."""
    from embodied_object_detection_amd import ops
    g = torch.Generator().manual_seed(9)
    N = 1000
    mem = (torch.randn((N, 512), generator=g) * 20).to(dev)
    obs = torch.randint(0, 4, (N,), generator=g).float().to(dev)
    table = ops.memory_normalize_f16(mem, obs)
    dirty = torch.zeros((N,), dtype=torch.int32, device=dev)
    rows = torch.randperm(N, generator=g)[:137].to(dev)
    mem[rows] += torch.randn((137, 512), generator=g).to(dev)
    obs[rows] += 1
    dirty[rows] = 1
    untouched = table.clone()
    ops.memory_normalize_dirty_f16(mem, obs, dirty, table)
    assert torch.equal(table, ops.memory_normalize_f16(mem, obs))
    assert int(dirty.sum().item()) == 0
    keep = torch.ones(N, dtype=torch.bool, device=dev); keep[rows] = False
    assert torch.equal(table[keep], untouched[keep])
    assert not torch.equal(table[rows], untouched[rows])


@pytest.mark.parametrize("mode,weight", [("sum", 5.0), ("mem_only", 500.0)])
def test_memory_project_fuse_matches_oracle(dev, mode, weight):
    """a8 projection + fusion: three 1x1 convs (f32 in the reference, timm.py:174) on fp16-exact inputs, x weight, + P_l.  The
    two-piece f16 split reproduces every fp32 weight to 2^-24 relative (one fp32 rounding); products are exact, accumulation
    fp32: the result must be as close to an f64 convolution as torch's own fp32 convolution is."""
    from embodied_object_detection_amd import ops
    g = torch.Generator().manual_seed(31)
    H, W = 64, 96
    hw = [(H // s, W // s) for s in (8, 16, 32)]
    pooled = [(torch.randn((1, 512, h, w), generator=g) * 8).half() for (h, w) in hw]
    pooled[0][0, :, 0, 0] = 0                                  # an unobserved location
    res = [torch.randn((1, 256, h, w), generator=g) for (h, w) in hw]
    sd = {}
    for i in (1, 2, 3):
        wgt = torch.randn((256, 512, 1, 1), generator=g) * (1.0 / 512) ** 0.5 * 0.05
        wgt[i] *= 1e-4                                          # a row of tiny weights (own power-of-two scale)
        wgt[7, ::3] *= 1e3                                      # a row with a wide dynamic range
        sd[f"backbone.map_merge_projection{i}.weight"] = wgt
        sd[f"backbone.map_merge_projection{i}.bias"] = torch.randn((256,), generator=g) * 0.01
    ref = M.fuse_memory(res, pooled, sd, M.OracleCfg(map_feat_fusion=mode, map_feature_weight=weight))
    ref64 = [(F.conv2d(p.double(), sd[f"backbone.map_merge_projection{i + 1}.weight"].double(),
                       sd[f"backbone.map_merge_projection{i + 1}.bias"].double()) * weight + (r.double() if mode == "sum" else 0))
             for i, (p, r) in enumerate(zip(pooled, res))]
    proj = ops.MemoryProjector([sd[f"backbone.map_merge_projection{i}.weight"] for i in (1, 2, 3)],
                               [sd[f"backbone.map_merge_projection{i}.bias"] for i in (1, 2, 3)], dev)
    feats = torch.cat([nhwc(r).reshape(-1, 256) for r in res]).contiguous().to(dev)
    rows16 = _rows_to_fragments([nhwc(p).reshape(-1, 512) for p in pooled]).contiguous().to(dev)
    assert rows16.shape[0] == ops.pooled_rows(H, W)
    tail = torch.full((7, 256), 3.25, device=dev)              # rows behind P5 (P6/P7 in the model) must not be touched
    buf = torch.cat([feats, tail]).contiguous()
    proj(rows16, buf, H, W, weight, mode)
    got = buf[:feats.shape[0]].cpu()
    assert torch.equal(buf[feats.shape[0]:].cpu(), tail.cpu())
    refrows = torch.cat([nhwc(r).reshape(-1, 256) for r in ref])
    ref64rows = torch.cat([nhwc(r).reshape(-1, 256) for r in ref64])
    scale = refrows.abs().max().item()
    e_hip = (got.double() - ref64rows).abs().max().item()
    e_cpu = (refrows.double() - ref64rows).abs().max().item()
    print(f"[project_fuse {mode}] max err vs f64: HIP {e_hip:.3e}, torch CPU f32 {e_cpu:.3e}, scale {scale:.3e}")
    assert e_hip <= 2.0 * e_cpu + 1e-7 * scale, "the split-f16 projection must be as accurate as an fp32 convolution"
    close(got, refrows, rtol=1e-5, atol=2e-6 * scale)


@pytest.mark.parametrize("K,thresh", [(12, 0.5), (0, 0.5)])
def test_memory_write_matches_oracle(dev, K, thresh):
    from embodied_object_detection_amd import ops
    g = torch.Generator().manual_seed(17)
    H, W, N, R = 64, 96, 300, 40
    boxes = torch.zeros((R, 4))
    ctr = torch.rand((R, 2), generator=g) * torch.tensor([float(W), float(H)])
    size = torch.rand((R, 2), generator=g) * 40 + 6
    boxes = torch.cat([ctr - size / 2, ctr + size / 2], dim=1)
    masks = torch.rand((R, 28, 28), generator=g)
    featn = 50 * F.normalize(torch.randn((R, 512), generator=g), dim=1)
    proj = torch.randint(0, N, (H, W), generator=g)
    det_rows = torch.randint(0, R, (max(K, 1),), generator=g)
    mem0 = torch.randn((N, 512), generator=g)
    obs0 = torch.randint(0, 3, (N,), generator=g).float()
    # oracle
    mem_ref, obs_ref = mem0.clone(), obs0.clone()
    if K > 0:
        rows = torch.unique(det_rows[:K])
        pasted = OO.paste_masks(masks[rows], boxes[rows], (H, W), thresh)
        mean, observed_mem = OM.memory_write_sparse(featn[rows], pasted, proj, N)
        upd = torch.zeros((N, 512)); upd[observed_mem] = mean
        mem_ref = mem_ref + upd
        ou = torch.zeros(N); ou[torch.unique(proj)] = 1
        obs_ref = obs_ref + ou
    wr = ops.MemoryWriter(H, W, N, 100, R, dev, mask_thresh=thresh)
    mem_d, obs_d = mem0.to(dev), obs0.to(dev)
    dr = torch.zeros(100, dtype=torch.int32); dr[:max(K, 1)] = det_rows.int()
    cnt = torch.tensor([K], dtype=torch.int32, device=dev)
    for rep in range(2):   # second call must start from clean per-frame flags
        mem_d, obs_d = mem0.to(dev), obs0.to(dev)
        dirty = torch.zeros((N,), dtype=torch.int32, device=dev)
        err = torch.zeros((1,), dtype=torch.int32, device=dev)
        k_out = wr(featn.to(dev), boxes.to(dev), masks.to(dev), dr.to(dev), cnt, proj.int().to(dev), mem_d, obs_d, dirty=dirty, err=err)
        assert int(k_out.item()) == (len(torch.unique(det_rows[:K])) if K else 0)
        assert torch.equal(obs_d.cpu(), obs_ref), "observation counters are integers: exact"
        assert torch.equal(dirty.cpu().bool(), obs_ref != obs0), "dirty rows = cells whose observation count changed"
        assert int(err.item()) == 0
        touched_ref = (mem_ref != mem0).any(dim=1)
        touched = (mem_d.cpu() != mem0).any(dim=1)
        assert torch.equal(touched, touched_ref), "set of written cells must be bit-exact"
        close(mem_d, mem_ref, rtol=1e-5, atol=1e-4)
        assert bool(((mem_d.cpu() != mem0).any(dim=1) <= dirty.cpu().bool()).all()), "written cells are a subset of the dirty rows"
        # write-through snapshot: the same call with `snapshot` leaves exactly the table normalize_dirty would produce, bit for
        # bit, and the same memory / counters; rows of untouched cells keep whatever they held
        snap_a = ops.memory_normalize_f16(mem0.to(dev), obs0.to(dev))
        ops.memory_normalize_dirty_f16(mem_d, obs_d, dirty, snap_a)
        assert int(dirty.sum().item()) == 0
        mem_e, obs_e = mem0.to(dev), obs0.to(dev)
        snap_b = ops.memory_normalize_f16(mem_e, obs_e)
        marker = (obs_ref == obs0).nonzero().squeeze(1)[:5].to(dev)
        snap_b[marker] = 7.0                                    # must survive: those cells did not change
        snap_a[marker] = 7.0
        wr(featn.to(dev), boxes.to(dev), masks.to(dev), dr.to(dev), cnt, proj.int().to(dev), mem_e, obs_e, err=err, snapshot=snap_b)
        assert torch.equal(mem_e, mem_d) and torch.equal(obs_e, obs_d)
        assert torch.equal(snap_b, snap_a), "write-through snapshot == incremental normalise of the dirty rows"
        assert torch.equal(snap_b.cpu(), torch.where(((obs_ref == obs0)[:, None]) & (torch.arange(N)[:, None] >= 0) &
                                                     torch.isin(torch.arange(N), marker.cpu())[:, None],
                                                     torch.full((N, 512), 7.0).half(),
                                                     OM.create_implicit_memory(mem_d.cpu(), obs_ref).half()))
    if K > 0:
        # an index image written for another map size: clamped and flagged, no fault; valid pixels behave as before
        bad = proj.clone()
        bad[0, 0], bad[5, 7] = N + 100, -3
        fixed = proj.clone()
        fixed[0, 0], fixed[5, 7] = N - 1, 0
        res = []
        for pj in (bad, fixed):
            mem_d, obs_d = mem0.to(dev), obs0.to(dev)
            err = torch.zeros((1,), dtype=torch.int32, device=dev)
            wr(featn.to(dev), boxes.to(dev), masks.to(dev), dr.to(dev), cnt, pj.int().to(dev), mem_d, obs_d, err=err)
            res.append((mem_d.cpu(), obs_d.cpu(), int(err.item())))
        assert res[0][2] == 1 and res[1][2] == 0
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_semmap_labels_match_oracle(dev):
    from embodied_object_detection_amd import ops
    from embodied_object_detection_amd.checkpoint import load_zs_weight
    g = torch.Generator().manual_seed(23)
    N = 700
    zs = load_zs_weight()
    mem = torch.randn((N, 512), generator=g) * torch.rand((N, 1), generator=g) * 30
    mem[::9] = 0                                   # never-written cells
    obs = torch.randint(0, 5, (N,), generator=g).float()
    ref = OM.semmap_labels(mem, obs, zs, 0.4)
    got = ops.semmap_labels(mem.to(dev), obs.to(dev), zs.to(dev), 0.4).cpu()
    agree = (got == ref).float().mean().item()
    assert agree > 0.995, f"semantic-map labels agree on {agree:.4f} of the cells"
    assert (got == -1).sum().item() > 0 and (got >= 0).sum().item() > 0


@pytest.mark.parametrize("cout,splitk,tile", [(256, 0, 0), (5, 0, 0), (64, 3, 0), (256, 1, 53), (64, 2, 52), (256, 1, 54)])
def test_conv_pyramid_mode_matches_per_level_conv(dev, cout, splitk, tile):
    """One launch over the 5 FPN levels with shared weights == five per-level 'same' convolutions."""
    from embodied_object_detection_amd import ops
    hw = [(12, 20), (6, 10), (3, 5), (2, 3), (1, 2)]
    Cin = 64
    xs = [rnd(1, Cin, h, w, seed=50 + i) for i, (h, w) in enumerate(hw)]
    w = rnd(cout, Cin, 3, 3, seed=60, scale=0.05)
    b = rnd(cout, seed=61)
    off = [0]
    for h, ww in hw:
        off.append(off[-1] + h * ww)
    flat = torch.cat([nhwc(x).reshape(-1, Cin) for x in xs]).contiguous().to(dev)
    conv = ops.Conv(w, b, pad=1, device=dev)
    y = conv(flat, 1, 0, 0, relu=True, levels=(off, hw), force_splitk=splitk, force_tile=tile).cpu()
    for i, x in enumerate(xs):
        ref = F.relu(F.conv2d(x, w, b, padding=1))
        close(y[off[i]:off[i + 1]], nhwc(ref).reshape(-1, cout))
