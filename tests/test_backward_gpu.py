"""Training forward, first slice: backward of the memory READ (timm.py:142-192) on HIP against torch autograd
on the oracle's own forward (`oracle/model.py::memory_read_pooled` / `fuse_memory`, run on the CPU in fp32 with the reference's
fp16 casts): dW / db of the three `map_merge_projection` 1x1 convolutions and the gradients of the cascaded pools."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import model as M


@pytest.mark.parametrize("H,W,n_cells", [(64, 96, 300), (128, 160, 576)])
def test_memory_read_backward_matches_autograd(H, W, n_cells):
    from embodied_object_detection_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    weight = 5.0
    mem16 = (torch.randn((n_cells, 512), generator=g) * 3).half()
    proj = torch.randint(0, n_cells, (H, W), generator=g)
    Ws = [(torch.randn((256, 512, 1, 1), generator=g) * 0.05).requires_grad_() for _ in range(3)]
    bs = [(torch.randn((256,), generator=g) * 0.1).requires_grad_() for _ in range(3)]
    # ---- oracle forward with autograd (memory_read_pooled + fuse_memory, written out so that the pooled tensors keep their grads)
    ego = mem16[proj].permute(2, 0, 1).unsqueeze(0)
    e2 = F.avg_pool2d(ego.to(torch.float32), kernel_size=4, stride=4).detach().requires_grad_()       # E_2: fp32 leaf
    pooled, cur = [], e2
    for _ in range(3):
        cur = F.avg_pool2d(cur.to(torch.float32), kernel_size=2, stride=2).to(torch.half)
        cur.retain_grad()
        pooled.append(cur)
    ref_pooled = M.memory_read_pooled(mem16, proj)
    assert all(torch.equal(a.detach(), b) for a, b in zip(pooled, ref_pooled)), "the test's forward is the oracle's forward"
    res = [torch.randn((1, 256, H >> (3 + l), W >> (3 + l)), generator=g) for l in range(3)]
    G = [torch.randn((1, 256, H >> (3 + l), W >> (3 + l)), generator=g) for l in range(3)]
    loss = 0.0
    for l in range(3):
        out = F.conv2d(pooled[l].to(torch.float32), Ws[l], bs[l]) * weight + res[l]
        loss = loss + (out * G[l]).sum()
    loss.backward()
    # ---- HIP: forward's pooled buffer (fragment order) from the product's gather, then the backward kernels
    pooled_d = ops.memory_gather_pool(mem16.to(dev), proj.int().to(dev), H, W, torch_order=True)
    bwd = ops.MemoryProjectorBackward([w.detach() for w in Ws], dev)
    grads = [G[l][0].permute(1, 2, 0).reshape(-1, 256).contiguous().to(dev) for l in range(3)]
    out = bwd(grads, pooled_d, H, W, weight)
    for l in range(3):
        ref_dw = Ws[l].grad.reshape(256, 512)
        scale = ref_dw.abs().max().item()
        err = (out["dW"][l].cpu() - ref_dw).abs().max().item()
        assert err <= 2e-5 * scale, f"dW{l + 3}: {err:.3e} at scale {scale:.3e}"
        ref_db = bs[l].grad
        assert (out["db"][l].cpu() - ref_db).abs().max().item() <= 2e-5 * ref_db.abs().max().item(), f"db{l + 3}"
        # gradients of the half tensors: autograd rounds them to half on the way; identical up to one half ulp where the fp32
        # convolution's summation order moved a value across a rounding boundary
        ref_g = pooled[l].grad[0].permute(1, 2, 0).reshape(-1, 512).float()
        got_g = out["gE"][l].cpu().float()
        # (a value that is the half sum of two contributions of opposite sign carries the ulp of the larger contribution)
        tol = 2.0 ** -10 * float(ref_g.abs().max())
        assert bool(((got_g - ref_g).abs() <= tol).all()), f"gE{l + 3}: {float((got_g - ref_g).abs().max()):.3e} > {tol:.3e}"
        assert float((got_g == ref_g).float().mean()) > 0.99, f"gE{l + 3}: mostly bit-identical halves"
    ref_g2 = e2.grad[0].permute(1, 2, 0).reshape(-1, 512)
    got_g2 = out["gE2"].cpu()
    assert bool(((got_g2 - ref_g2).abs() <= 2.0 ** -10 * float(ref_g2.abs().max())).all())
    assert float((got_g2 == ref_g2).float().mean()) > 0.99


def test_memory_read_training_steps_match_torch(tmp_path):
    """Second slice: three TRAINING steps of the memory-specific parameters (`map_merge_projection{1,2,3}`: weights and biases) as the
    reference's configuration runs them -- forward of the memory read + fusion, a loss that is linear in the fused pyramid, backward,
    gradient clipping by value 1.0, AdamW at BASE_LR x CUSTOM_MULTIPLIER (custom_solver.py:19-79, ..._mp3d_recurrent.yaml:28-38) --
    entirely on the HIP kernels (gather/pool, project + fuse with re-prepared weights, the backward kernels, `eod_adamw_step`)
    against torch autograd + `torch.optim.AdamW` on the oracle's forward."""
    from embodied_object_detection_amd import ops, solver
    dev = torch.device("cuda:0")
    H, W, n_cells, weight = 64, 96, 300, 5.0
    g = torch.Generator().manual_seed(11)
    mem16 = (torch.randn((n_cells, 512), generator=g) * 3).half()
    proj = torch.randint(0, n_cells, (H, W), generator=g)
    W0 = [torch.randn((256, 512, 1, 1), generator=g) * 0.05 for _ in range(3)]
    b0 = [torch.randn((256,), generator=g) * 0.1 for _ in range(3)]
    rows = [(H >> (3 + l)) * (W >> (3 + l)) for l in range(3)]
    base_lr, mult, wd = 1e-3, 10.0, 1e-4           # a larger BASE_LR than the yaml's 1e-5 so that three steps move the weights visibly
    names = [f"backbone.map_merge_projection{l + 1}.{k}" for l in range(3) for k in ("weight", "bias")]
    # ---- torch reference
    Wt = [w.clone().requires_grad_() for w in W0]
    bt = [b.clone().requires_grad_() for b in b0]
    topt = torch.optim.AdamW([{"params": [t], "lr": base_lr * mult} for pair in zip(Wt, bt) for t in pair], base_lr, weight_decay=wd)
    # ---- HIP side: fp32 master parameters on the device
    Wd = [w.reshape(256, 512).contiguous().to(dev) for w in W0]
    bd = [b.clone().to(dev) for b in b0]
    groups = solver.build_param_groups(list(zip(names, [t for pair in zip(Wd, bd) for t in pair])), base_lr, wd, "ADAMW", 1.0, mult, ["map_merge"])
    assert [gr["lr"] for gr in groups] == [base_lr * mult] * 6
    opt = ops.AdamW(groups, weight_decay=wd, clip_value=1.0)
    pooled_d = ops.memory_gather_pool(mem16.to(dev), proj.int().to(dev), H, W, torch_order=True)
    pooled_ref = [p.to(torch.float32) for p in M.memory_read_pooled(mem16, proj)]
    for it in range(3):
        res = [torch.randn((1, 256, H >> (3 + l), W >> (3 + l)), generator=g) for l in range(3)]
        G = [torch.randn((1, 256, H >> (3 + l), W >> (3 + l)), generator=g) * 0.02 for l in range(3)]      # some |grad| above the clip, most below
        # torch: forward, loss, backward, clip, step
        topt.zero_grad()
        loss = sum(((F.conv2d(pooled_ref[l], Wt[l], bt[l]) * weight + res[l]) * G[l]).sum() for l in range(3))
        loss.backward()
        for t in Wt + bt:
            t.grad.clamp_(-1.0, 1.0)
        topt.step()
        # HIP: forward (weights re-prepared from the fp32 masters), backward, optimizer
        projector = ops.MemoryProjector([w.reshape(256, 512, 1, 1) for w in Wd], bd, dev)
        feats = torch.cat([res[l][0].permute(1, 2, 0).reshape(-1, 256) for l in range(3)]).contiguous().to(dev)
        projector(pooled_d, feats, H, W, weight, "sum")
        loss_hip = float((feats.cpu() * torch.cat([G[l][0].permute(1, 2, 0).reshape(-1, 256) for l in range(3)])).sum())
        assert abs(loss_hip - float(loss.detach())) <= 1e-4 * max(1.0, abs(float(loss.detach()))), (it, loss_hip, float(loss.detach()))
        grads_in = [G[l][0].permute(1, 2, 0).reshape(-1, 256).contiguous().to(dev) for l in range(3)]
        out = ops.MemoryProjectorBackward([w.reshape(256, 512, 1, 1) for w in Wd], dev)(grads_in, pooled_d, H, W, weight)
        opt.step([t for l in range(3) for t in (out["dW"][l], out["db"][l])])
        for l in range(3):
            for got, ref, what in ((Wd[l], Wt[l].detach().reshape(256, 512), "weight"), (bd[l], bt[l].detach(), "bias")):
                err = float((got.cpu() - ref).abs().max())
                # AdamW's first steps move every element by ~lr whatever the gradient's size: a gradient that differs in the last
                # bits moves m / sqrt(v) by ~1e-6 relative, far below this bound; an element whose clipped / unclipped state differed
                # would show up as ~lr
                assert err <= 2e-6, (it, l, what, err)
        assert float((Wd[0].cpu() - W0[0].reshape(256, 512)).abs().max()) > 0.5 * base_lr * mult * (it + 1) * 0.5


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,relu,stride", [(1, 20, 28, 256, 256, 3, True, 1), (2, 14, 14, 256, 256, 3, True, 1),
                                                          (1, 10, 12, 64, 96, 1, False, 1), (1, 9, 7, 32, 64, 5, True, 1),
                                                          (1, 20, 20, 256, 256, 3, False, 2), (1, 10, 10, 256, 256, 3, False, 2),
                                                          (2, 13, 9, 64, 32, 3, True, 2), (1, 16, 16, 64, 128, 1, False, 2)])
def test_conv_layer_backward_matches_autograd(N, H, W, Cin, Cout, k, relu, stride):
    """Third slice: dX, dW, db of a conv + bias (+ ReLU) layer -- stride 1 'same' (tower / FPN output / mask head: dX on the matrix
    cores) and stride 2 (P6 / P7 and the trunk's down-sampling layers: dX by the gather kernel) -- against torch autograd."""
    from embodied_object_detection_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21)
    w = (torch.randn((Cout, Cin, k, k), generator=g) * (0.5 / (Cin * k * k) ** 0.5)).requires_grad_()
    b = (torch.randn((Cout,), generator=g) * 0.1).requires_grad_()
    x = torch.randn((N, Cin, H, W), generator=g).requires_grad_()
    OH, OW = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    go = torch.randn((N, Cout, OH, OW), generator=g)
    y = F.conv2d(x, w, b, padding=k // 2, stride=stride)
    if relu:
        y = F.relu(y)
    (y * go).sum().backward()
    conv = ops.Conv(w.detach(), b.detach(), stride=stride, pad=k // 2, device=dev)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    yd = conv(xd, N, H, W, relu=relu)
    out = ops.ConvBackward(conv)(xd, yd, go.permute(0, 2, 3, 1).contiguous().to(dev), relu=relu)
    ref_dw = w.grad.permute(0, 2, 3, 1).reshape(Cout, -1)                      # [co, (ky, kx, ci)]
    for name, got, ref in (("dW", out["dw"].cpu(), ref_dw), ("db", out["db"].cpu(), b.grad),
                           ("dX", out["dx"].cpu(), x.grad.permute(0, 2, 3, 1))):
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max())
        assert err <= 3e-5 * scale, f"{name}: {err:.3e} at scale {scale:.3e}"
    # the weight gradient feeds the optimizer in the packed layout of the forward weights
    K = Cin * k * k
    opt = ops.AdamW([{"name": "w", "param": conv.w[:, :K].contiguous(), "lr": 1e-3}], weight_decay=0.0)
    before = opt.groups[0]["param"].clone()
    opt.step([out["dw"]])
    moved = (opt.groups[0]["param"] - before).abs()
    assert float(moved.max()) <= 1.01e-3 and float(moved.mean()) > 0.9e-3      # AdamW's first step: ~lr per element


def test_groupnorm_relu_backward_matches_autograd():
    """GroupNorm(32) + ReLU over a feature pyramid stored as one row list (per level image statistics, parameters shared by the
    levels: centernet_head.py:76-79): dx, dgamma, dbeta against torch autograd; and a whole tower layer (3x3 conv -> GroupNorm -> ReLU)
    chained with the conv backward of the third slice."""
    from embodied_object_detection_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31)
    shapes = [(12, 16), (6, 8), (3, 4)]
    off = [0]
    for h, w in shapes:
        off.append(off[-1] + h * w)
    Cc = 256
    gamma = (torch.rand(Cc, generator=g) + 0.5).requires_grad_()
    beta = (torch.randn(Cc, generator=g) * 0.2).requires_grad_()
    w = (torch.randn((Cc, Cc, 3, 3), generator=g) * 0.02).requires_grad_()
    b = (torch.randn((Cc,), generator=g) * 0.1).requires_grad_()
    xs = [torch.randn((1, Cc, h, ww), generator=g).requires_grad_() for h, ww in shapes]
    gos = [torch.randn((1, Cc, h, ww), generator=g) for h, ww in shapes]
    # ---- torch: the layer per level (shared parameters), autograd
    pre, loss = [], 0.0
    for x, go in zip(xs, gos):
        c = F.conv2d(x, w, b, padding=1)
        c.retain_grad()
        pre.append(c)
        loss = loss + (F.relu(F.group_norm(c, 32, gamma, beta, eps=1e-5)) * go).sum()
    loss.backward()
    rows = lambda ts: torch.cat([t[0].permute(1, 2, 0).reshape(-1, Cc) for t in ts]).contiguous()
    # ---- HIP forward (pyramid mode), then backward: GroupNorm + ReLU, then the conv
    conv = ops.Conv(w.detach(), b.detach(), pad=1, device=dev)
    xd = rows([x.detach() for x in xs]).to(dev)
    cd = conv(xd, 1, 0, 0, levels=(off, shapes))
    assert float((cd.cpu() - rows([c.detach() for c in pre])).abs().max()) < 1e-4
    stats = ops.groupnorm_workspace(off, dev)
    yd = ops.groupnorm_relu(cd, gamma.detach().to(dev), beta.detach().to(dev), off, Cc, stats)
    dyd = rows(gos).to(dev)
    dc, dgamma, dbeta = ops.groupnorm_relu_backward(cd, yd, dyd, gamma.detach().to(dev), off, Cc, stats)
    ref_dc = rows([c.grad for c in pre])
    for name, got, ref in (("d(pre-norm)", dc.cpu(), ref_dc), ("dgamma", dgamma.cpu(), gamma.grad), ("dbeta", dbeta.cpu(), beta.grad)):
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max())
        assert err <= 5e-5 * scale, f"{name}: {err:.3e} at scale {scale:.3e}"
    # the conv of the layer, level by level (the weight gradients of the levels add up: shared weights)
    bwd = ops.ConvBackward(conv)
    dw = torch.zeros((Cc, 9 * Cc), device=dev)
    db = torch.zeros((Cc,), device=dev)
    dxs = []
    for l, (h, ww) in enumerate(shapes):
        xl = xd[off[l]:off[l + 1]].view(1, h, ww, Cc)
        gl = dc[off[l]:off[l + 1]].view(1, h, ww, Cc).contiguous()
        o = bwd(xl, None, gl)
        dw += o["dw"]
        db += o["db"]
        dxs.append(o["dx"].reshape(-1, Cc))
    ref_dw = w.grad.permute(0, 2, 3, 1).reshape(Cc, -1)
    assert float((dw.cpu() - ref_dw).abs().max()) <= 5e-5 * float(ref_dw.abs().max())
    assert float((db.cpu() - b.grad).abs().max()) <= 5e-5 * float(b.grad.abs().max())
    ref_dx = rows([x.grad for x in xs])
    assert float((torch.cat(dxs).cpu() - ref_dx).abs().max()) <= 5e-5 * float(ref_dx.abs().max())


def test_head_loss_gradient_reaches_the_memory_projections():
    """End to end through the training slices: a loss on the CenterNet head's outputs (agn_hm + bbox_pred logits of P3..P5) is
    back-propagated on the HIP kernels through the 5-channel output conv, the four tower layers (3x3 conv -> GroupNorm -> ReLU, shared
    over the levels) and the memory fusion down to the `map_merge_projection` weights and biases -- the parameters the recurrent
    configuration trains at 10 x the base rate -- and compared with torch autograd on the oracle's forward (timm.py:142-192,
    centernet_head.py:141-161)."""
    from embodied_object_detection_amd import ops
    dev = torch.device("cuda:0")
    H, W, n_cells, weight, Cc = 64, 96, 300, 5.0, 256
    g = torch.Generator().manual_seed(41)
    mem16 = (torch.randn((n_cells, 512), generator=g) * 3).half()
    proj = torch.randint(0, n_cells, (H, W), generator=g)
    shapes = [(H >> (3 + l), W >> (3 + l)) for l in range(3)]
    off = [0]
    for h, w in shapes:
        off.append(off[-1] + h * w)
    Wm = [(torch.randn((256, 512, 1, 1), generator=g) * 0.02).requires_grad_() for _ in range(3)]
    bm = [(torch.randn((256,), generator=g) * 0.1).requires_grad_() for _ in range(3)]
    Wt = [(torch.randn((Cc, Cc, 3, 3), generator=g) * 0.03).requires_grad_() for _ in range(4)]
    bt = [(torch.randn((Cc,), generator=g) * 0.1).requires_grad_() for _ in range(4)]
    gam = [(torch.rand(Cc, generator=g) + 0.5).requires_grad_() for _ in range(4)]
    bet = [(torch.randn(Cc, generator=g) * 0.2).requires_grad_() for _ in range(4)]
    Wo = (torch.randn((5, Cc, 3, 3), generator=g) * 0.03).requires_grad_()
    bo = (torch.randn((5,), generator=g) * 0.1).requires_grad_()
    res = [torch.randn((1, Cc, h, w), generator=g) for h, w in shapes]
    Go = [torch.randn((1, 5, h, w), generator=g) for h, w in shapes]
    # ---- torch autograd on the oracle's forward
    pooled_ref = [p.to(torch.float32) for p in M.memory_read_pooled(mem16, proj)]
    loss = 0.0
    for l in range(3):
        t = F.conv2d(pooled_ref[l], Wm[l], bm[l]) * weight + res[l]
        for i in range(4):
            t = F.relu(F.group_norm(F.conv2d(t, Wt[i], bt[i], padding=1), 32, gam[i], bet[i], eps=1e-5))
        loss = loss + (F.conv2d(t, Wo, bo, padding=1) * Go[l]).sum()
    loss.backward()
    rows = lambda ts: torch.cat([t[0].permute(1, 2, 0).reshape(-1, t.shape[1]) for t in ts]).contiguous()
    # ---- HIP forward, every layer's input / pre-norm / output kept
    pooled_d = ops.memory_gather_pool(mem16.to(dev), proj.int().to(dev), H, W, torch_order=True)
    feats = rows(res).to(dev)
    ops.MemoryProjector([w.detach() for w in Wm], [b.detach() for b in bm], dev)(pooled_d, feats, H, W, weight, "sum")
    tower = [ops.Conv(Wt[i].detach(), bt[i].detach(), pad=1, device=dev) for i in range(4)]
    # the output conv padded to 32 channels (zero rows): the backward kernels work on 32-channel tiles
    Wo32 = torch.zeros((32, Cc, 3, 3)); Wo32[:5] = Wo.detach()
    bo32 = torch.zeros((32,)); bo32[:5] = bo.detach()
    outc = ops.Conv(Wo32, bo32, pad=1, device=dev)
    keep, x = [], feats
    for i in range(4):
        c = tower[i](x, 1, 0, 0, levels=(off, shapes))
        st = ops.groupnorm_workspace(off, dev)
        y = ops.groupnorm_relu(c, gam[i].detach().to(dev), bet[i].detach().to(dev), off, Cc, st)
        keep.append((x, c, st, y))
        x = y
    head = outc(x, 1, 0, 0, levels=(off, shapes))
    # ---- HIP backward
    def conv_bwd(conv, xin, gout):
        bwd = ops.ConvBackward(conv)
        dw = torch.zeros((conv.Cout, conv.KH * conv.KW * conv.Cin), device=dev)
        db = torch.zeros((conv.Cout,), device=dev)
        dxs = []
        for l, (h, w) in enumerate(shapes):
            o = bwd(xin[off[l]:off[l + 1]].view(1, h, w, conv.Cin), None, gout[off[l]:off[l + 1]].view(1, h, w, conv.Cout).contiguous())
            dw += o["dw"]
            db += o["db"]
            dxs.append(o["dx"].reshape(-1, conv.Cin))
        return torch.cat(dxs).contiguous(), dw, db
    G32 = torch.zeros((off[-1], 32), device=dev)
    G32[:, :5] = rows(Go).to(dev)
    gx, dWo, dbo = conv_bwd(outc, x, G32)
    ref = Wo.grad.permute(0, 2, 3, 1).reshape(5, -1)
    assert float((dWo[:5].cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    for i in reversed(range(4)):
        xin, c, st, y = keep[i]
        dc, dga, dbe = ops.groupnorm_relu_backward(c, y, gx, gam[i].detach().to(dev), off, Cc, st)
        gx, dw, db = conv_bwd(tower[i], xin, dc)
        for name, got, r in (("tower dW", dw.cpu(), Wt[i].grad.permute(0, 2, 3, 1).reshape(Cc, -1)), ("tower db", db.cpu(), bt[i].grad),
                             ("dgamma", dga.cpu(), gam[i].grad), ("dbeta", dbe.cpu(), bet[i].grad)):
            assert float((got - r).abs().max()) <= 2e-4 * float(r.abs().max()), (i, name, float((got - r).abs().max()), float(r.abs().max()))
    # gx = dL/d(fused P3..P5): into the backward of the memory read (first slice)
    out = ops.MemoryProjectorBackward([w.detach() for w in Wm], dev)([gx[off[l]:off[l + 1]].contiguous() for l in range(3)], pooled_d, H, W, weight)
    for l in range(3):
        rdw, rdb = Wm[l].grad.reshape(256, 512), bm[l].grad
        assert float((out["dW"][l].cpu() - rdw).abs().max()) <= 3e-4 * float(rdw.abs().max()), l
        assert float((out["db"][l].cpu() - rdb).abs().max()) <= 3e-4 * float(rdb.abs().max()), l


def test_fpn_add_and_maxpool_backward_match_autograd():
    """The two non-conv pieces of the backbone's backward: the FPN top-down add (nearest x2 of the coarser level, timm.py:128-133) and
    the trunk's 3x3 stride-2 max pool (timm.py:281; ties go to the first maximum of a window, as torch routes them)."""
    import ctypes as C
    from embodied_object_detection_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(51)
    # nearest x2 add
    coarse = torch.randn((2, 64, 5, 7), generator=g).requires_grad_()
    fine = torch.randn((2, 64, 10, 14), generator=g).requires_grad_()
    go = torch.randn((2, 64, 10, 14), generator=g)
    ((fine + F.interpolate(coarse, scale_factor=2.0, mode="nearest")) * go).sum().backward()
    gd = go.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.full((2, 5, 7, 64), 1.5, device=dev)
    _lib.check(lib.eod_upsample2_sum_backward(gd.data_ptr(), out.data_ptr(), 2, 5, 7, 64, 1, s), "up")
    ref = coarse.grad.permute(0, 2, 3, 1) + 1.5
    assert float((out.cpu() - ref).abs().max()) <= 1e-5
    _lib.check(lib.eod_upsample2_sum_backward(gd.data_ptr(), out.data_ptr(), 2, 5, 7, 64, 0, s), "up")
    assert float((out.cpu() - coarse.grad.permute(0, 2, 3, 1)).abs().max()) <= 1e-5
    # max pool, odd and even sizes, with ties (quantised values)
    for (H, W) in ((12, 16), (11, 9)):
        x = (torch.randn((2, 32, H, W), generator=g) * 2).round().requires_grad_()      # many equal values: ties inside the windows
        y = F.max_pool2d(x, 3, 2, 1)
        go = torch.randn(y.shape, generator=g)
        (y * go).sum().backward()
        OH, OW = y.shape[2], y.shape[3]
        xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev)
        yd = y.detach().permute(0, 2, 3, 1).contiguous().to(dev)
        gd = go.permute(0, 2, 3, 1).contiguous().to(dev)
        dx = torch.empty_like(xd)
        _lib.check(lib.eod_maxpool3x3s2_backward(xd.data_ptr(), yd.data_ptr(), gd.data_ptr(), dx.data_ptr(), 2, H, W, 32, OH, OW, s), "mp")
        assert float((dx.cpu() - x.grad.permute(0, 2, 3, 1)).abs().max()) <= 1e-5, (H, W)


def test_backbone_backward_matches_autograd(synthetic_sd):
    """The whole backbone's backward as a chain of the per-layer HIP kernels (`modeling/backward.py`): gradients of P3..P7 flow through
    P7 / P6, the FPN output convs, the top-down adds, the laterals and all 16 bottleneck blocks of the ResNet-50 trunk (residual
    splits, strided 3x3 and 1x1 convs, ReLU masks, max pool) down to the 7x7 stem -- every conv's dW / db (61 layers) and the gradient of the stem's pre-activation against
    torch autograd on the oracle's forward (timm.py:277-299, 118-136, 347-364).  The trunk's FrozenBatchNorm is folded into the
    product's weights: dW_raw = dW_folded * gamma / sqrt(var + eps) per output channel, d(bn.bias) = db_folded."""
    from embodied_object_detection_amd import build_model, ops, setup_cfg
    from embodied_object_detection_amd.modeling.backward import BackboneBackward
    dev = torch.device("cuda:0")
    cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5])
    model = build_model(cfg, synthetic_sd)
    H, W = 64, 96
    g = torch.Generator().manual_seed(61)
    img = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8)
    # ---- torch autograd on the oracle's forward (no memory term: the sum fusion adds a constant with respect to the backbone)
    ocfg = M.OracleCfg(memory_type="none")
    sd = {k: (v.clone().float().requires_grad_() if (k.startswith("backbone.") and v.is_floating_point()
                                                      and "running_" not in k and "map_merge" not in k) else v)
          for k, v in synthetic_sd.items()}
    x = M.preprocess_image(img, ocfg)
    base = "backbone.bottom_up.base"
    stem = F.relu(M.frozen_bn(F.conv2d(x, sd[f"{base}.conv1.weight"], stride=2, padding=3), sd, f"{base}.bn1"))
    stem.retain_grad()
    t = F.max_pool2d(stem, kernel_size=3, stride=2, padding=1)
    feats = {}
    for li, nblk in enumerate((3, 4, 6, 3), start=1):
        for b in range(nblk):
            t = M.bottleneck(t, sd, f"{base}.layer{li}.{b}", 2 if (b == 0 and li > 1) else 1)
        feats[f"layer{li + 1}"] = t
    ref_P = M.fpn_top_down(feats, sd)
    ref_P = ref_P + M.top_block(ref_P[2], sd)
    Go = [torch.randn(p.shape, generator=g) for p in ref_P]
    sum((p * go).sum() for p, go in zip(ref_P, Go)).backward()
    # ---- HIP forward keeping the activations, then the chain of backward kernels
    x4, Hp, Wp = ops.preprocess_image(img.to(dev), model.pixel_mean, model.pixel_std)
    bw = BackboneBackward(model.backbone)
    P, saved = bw.forward(x4, Hp, Wp)
    for l in range(5):
        ref = ref_P[l].detach().permute(0, 2, 3, 1)
        assert float((P[l].cpu() - ref).abs().max()) <= 2e-4 * float(ref.abs().max()), l
    grads, g_stem = bw.backward(saved, [go.permute(0, 2, 3, 1).contiguous().to(dev) for go in Go])
    torch.cuda.synchronize()
    rel = lambda a, b: float((a - b).abs().max()) / max(float(b.abs().max()), 1e-20)
    worst = {}

    def check(name, dw, db, w_raw, b_raw, scale=None):
        rw = w_raw.grad.permute(0, 2, 3, 1)
        if rw.shape[-1] == 3:                                   # the stem: 4-channel tap layout, the fourth channel is zero
            rw = F.pad(rw, (0, 1))
        rw = rw.reshape(w_raw.shape[0], -1)
        got = dw.cpu()
        if scale is not None:
            got = got * scale[:, None]
        worst[name] = max(rel(got, rw), rel(db.cpu(), b_raw.grad))
        assert worst[name] <= 1e-4, (name, worst[name])

    for l in (3, 4, 5):
        check(f"fpn_lateral{l}", *grads[f"fpn_lateral{l}"], sd[f"backbone.fpn_lateral{l}.weight"], sd[f"backbone.fpn_lateral{l}.bias"])
        check(f"fpn_output{l}", *grads[f"fpn_output{l}"], sd[f"backbone.fpn_output{l}.weight"], sd[f"backbone.fpn_output{l}.bias"])
    check("p6", *grads["p6"], sd["backbone.top_block.p6.weight"], sd["backbone.top_block.p6.bias"])
    check("p7", *grads["p7"], sd["backbone.top_block.p7.weight"], sd["backbone.top_block.p7.bias"])
    n_trunk = 0
    for li, nblk in enumerate((3, 4, 6, 3), start=1):
        for b in range(nblk):
            p = f"{base}.layer{li}.{b}"
            pairs = [(f"{p}.conv{i}", f"{p}.conv{i}.weight", f"{p}.bn{i}") for i in (1, 2, 3)]
            if f"{p}.downsample.0.weight" in sd:
                pairs.append((f"{p}.downsample", f"{p}.downsample.0.weight", f"{p}.downsample.1"))
            for name, wk, bnp in pairs:
                scale = (sd[f"{bnp}.weight"] / torch.sqrt(sd[f"{bnp}.running_var"] + 1e-5)).detach()
                check(name, *grads[name], sd[wk], sd[f"{bnp}.bias"], scale)
                n_trunk += 1
    bn1 = f"{base}.bn1"
    check("stem", *grads["stem"], sd[f"{base}.conv1.weight"], sd[f"{bn1}.bias"],
          (sd[f"{bn1}.weight"] / torch.sqrt(sd[f"{bn1}.running_var"] + 1e-5)).detach())
    assert n_trunk == 52 and len(grads) == 52 + 8 + 1
    ref_gs = (stem.grad * (stem.detach() > 0)).permute(0, 2, 3, 1)          # through the stem's ReLU: gradient of its pre-activation
    assert rel(g_stem.cpu(), ref_gs) <= 1e-4
    print("backbone backward: worst relative error %.2e (%s), stem-output gradient %.2e"
          % (max(worst.values()), max(worst, key=worst.get), rel(g_stem.cpu(), ref_gs)))


def test_roi_align_backward_matches_autograd():
    """Backward of the ROI pooler (detic_roi_heads.py:332,265): the gradient of 7x7 and 14x14 pooled features back into P3..P5 against
    torch autograd on the oracle's ROIAlignV2 (boxes on all three levels, one partly outside the image, one tiny); two poolers
    accumulate into the same level gradients, as the cascade stages and the mask pooler do."""
    from embodied_object_detection_amd import ops
    from oracle import ops as OO
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(71)
    H, W, Cc = 512, 640, 32
    shapes = [(H >> (3 + l), W >> (3 + l)) for l in range(3)]
    feats = [torch.randn((1, Cc, h, w), generator=g).requires_grad_() for h, w in shapes]
    boxes = torch.tensor([[10.0, 20.0, 60.0, 90.0], [5.5, 3.25, 330.0, 260.0], [100.0, 40.0, 638.0, 500.0], [-20.0, -10.0, 90.0, 70.0],
                          [200.0, 100.0, 203.0, 102.5], [30.0, 30.0, 150.0, 140.0], [0.0, 0.0, 640.0, 512.0], [450.0, 380.0, 800.0, 600.0]])
    assert len(set(OO.assign_boxes_to_levels(boxes).tolist())) == 3
    R = boxes.shape[0]
    Gs = {S: torch.randn((R, Cc, S, S), generator=g) for S in (7, 14)}
    loss = sum((OO.roi_pool(feats, boxes, S) * Gs[S]).sum() for S in (7, 14))
    loss.backward()
    d = [torch.zeros((h, w, Cc), device=dev) for h, w in shapes]
    count = torch.tensor([R], dtype=torch.int32, device=dev)
    for S in (7, 14):
        ops.roi_align_backward(d[0], d[1], d[2], shapes[0][0], shapes[0][1], Cc, boxes.to(dev), count, R, S,
                               Gs[S].permute(0, 2, 3, 1).contiguous().to(dev))
    for l in range(3):
        ref = feats[l].grad[0].permute(1, 2, 0)
        assert float((d[l].cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), l
    # the count caps the list: ROIs beyond it contribute nothing
    d2 = [torch.zeros_like(t) for t in d]
    ops.roi_align_backward(d2[0], d2[1], d2[2], shapes[0][0], shapes[0][1], Cc, boxes.to(dev), torch.zeros_like(count), R, 7,
                           Gs[7].permute(0, 2, 3, 1).contiguous().to(dev))
    assert all(float(t.abs().max()) == 0.0 for t in d2)
    # the gather form has no atomics: a crowd of overlapping ROIs (the training step's 512 sampled rows pile up on the objects) gives
    # bitwise the same gradient on every run, and the same numbers as the footprint / sample forms up to summation order
    gg = torch.Generator().manual_seed(72)
    R2, C2 = 512, 256
    ctr = torch.tensor([[200.0, 180.0], [420.0, 300.0], [90.0, 400.0]])[torch.randint(0, 3, (R2,), generator=gg)] + torch.randn((R2, 2), generator=gg) * 25
    half = torch.exp(torch.rand((R2, 2), generator=gg) * 2.8 + 2.2)                 # 9 .. 150 px half extents: levels 3 and 4
    crowd = torch.cat([ctr - half, ctr + half], dim=1).contiguous().to(dev)
    G7 = torch.randn((R2, 7, 7, C2), generator=gg).to(dev)
    cnt2 = torch.tensor([R2], dtype=torch.int32, device=dev)
    runs = []
    for _ in range(2):
        dd = [torch.zeros((h, w, C2), device=dev) for h, w in shapes]
        ops.roi_align_backward(dd[0], dd[1], dd[2], shapes[0][0], shapes[0][1], C2, crowd, cnt2, R2, 7, G7)
        runs.append(dd)
    assert all(torch.equal(a, b) for a, b in zip(*runs)) and float(runs[0][0].abs().max()) > 0 and float(runs[0][1].abs().max()) > 0


def _raw_from_reg(reg_pred, scales, off, g):
    """Head rows whose relu(scale_l * raw) is `reg_pred`: positives divided by the level's scale, zeros from negative raw values."""
    raw = torch.empty_like(reg_pred)
    for l, sc in enumerate(scales):
        r = reg_pred[off[l]:off[l + 1]]
        neg = -(torch.rand(r.shape, generator=g) + 0.1)
        raw[off[l]:off[l + 1]] = torch.where(r > 0, r / sc, neg)
    return raw


def test_centernet_loss_matches_the_reference_functions():
    """`eod_centernet_loss` (CenterNet.losses of the recurrent configuration, centernet.py:241-318) against the fixture made by the
    reference's OWN binary_heatmap_focal_loss_jit / IOULoss (tests/golden/gen_golden_losses.py): the three losses and the gradient with
    respect to the head's raw output rows (agnostic logit; bbox_pred before the level's Scale and the ReLU).  Then a second, larger
    case against the oracle's restatement with autograd through relu(scale * raw)."""
    import numpy as np
    from embodied_object_detection_amd import ops
    from oracle import losses as OL
    from test_losses_golden import load_fixture
    dev = torch.device("cuda:0")
    z, t, cfg = load_fixture()
    shapes = [tuple(x) for x in z["shapes"].tolist()]
    off = [0]
    for h, w in shapes:
        off.append(off[-1] + h * w)
    scales = [1.0, 0.8, 1.3, 2.0, 0.5]
    g = torch.Generator().manual_seed(5)

    def run(logits, reg_pred, heat, reg_targets, pos, off, scales):
        M = logits.numel()
        raw = _raw_from_reg(reg_pred, scales, off, g)
        # what relu(scale * raw) really is in fp32 (the division above is not exactly undone): the reference values are taken on it
        reg_eff = torch.cat([torch.relu(raw[off[l]:off[l + 1]] * scales[l]) for l in range(len(scales))])
        head = torch.zeros((M, 8))
        head[:, 0] = logits
        head[:, 1:5] = raw
        head[:, 5:] = torch.randn((M, 3), generator=g)
        n_reg = int((reg_targets.max(dim=1)[0] >= 0).sum())
        loss_fn = ops.CenterNetLoss(off, scales, dev, head_stride=8, **{k: cfg[k] for k in ("alpha", "beta", "gamma", "sigmoid_clamp",
                                                                                             "ignore_high_fp", "pos_weight", "neg_weight",
                                                                                             "reg_weight")})
        losses, dh = loss_fn(head.to(dev), heat.to(dev), reg_targets.contiguous().to(dev), pos.int().to(dev), max(float(pos.numel()), 1.0),
                             max(float(n_reg), 1.0))
        return losses.cpu(), dh.cpu(), raw, reg_eff

    # ---- the reference's numbers
    losses, dh, raw, reg_eff = run(t("logits"), t("reg_pred"), t("heat"), t("reg_targets"), t("pos_inds"), off, scales)
    for i, k in enumerate(("loss_loc", "loss_agn_pos", "loss_agn_neg")):
        assert abs(float(losses[i]) - float(z[k])) <= 2e-5 * abs(float(z[k])), (k, float(losses[i]), float(z[k]))
    gl = t("grad_logits")
    assert float((dh[:, 0] - gl).abs().max()) <= 2e-5 * float(gl.abs().max())
    sc_rows = torch.cat([torch.full((off[l + 1] - off[l], 1), scales[l]) for l in range(5)])
    ref_raw_grad = t("grad_reg") * sc_rows * (raw > 0)
    assert float((dh[:, 1:5] - ref_raw_grad).abs().max()) <= 1e-4 * float(ref_raw_grad.abs().max())
    assert float(dh[:, 5:].abs().max()) == 0.0
    # ---- a pyramid of the full-size frame (640x640: 8 525 positions) against the oracle with autograd through scale + ReLU
    shapes2 = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]
    off2 = [0]
    for h, w in shapes2:
        off2.append(off2[-1] + h * w)
    M = off2[-1]
    logits = torch.randn((M,), generator=g) * 3
    heat = torch.rand((M,), generator=g) ** 8
    pos = torch.randperm(M, generator=g)[:150]
    heat[pos] = 1.0
    reg_pred = torch.rand((M, 4), generator=g) * 60
    reg_pred[torch.rand((M, 4), generator=g) < 0.05] = 0.0
    reg_targets = torch.full((M, 4), -1e8)
    rows = torch.randperm(M, generator=g)[:900]
    reg_targets[rows] = torch.rand((900, 4), generator=g) * 70 + 0.5
    losses, dh, raw, _ = run(logits, reg_pred, heat, reg_targets, pos, off2, scales)
    zl = logits.clone().requires_grad_()
    rw = raw.clone().requires_grad_()
    reg = torch.cat([torch.relu(rw[off2[l]:off2[l + 1]] * scales[l]) for l in range(5)])
    ref = OL.centernet_proposal_losses(zl, reg, heat, reg_targets, pos, **cfg)
    sum(ref.values()).backward()
    for i, k in enumerate(("loss_centernet_loc", "loss_centernet_agn_pos", "loss_centernet_agn_neg")):
        assert abs(float(losses[i]) - ref[k].item()) <= 2e-5 * abs(ref[k].item()), k
    assert float((dh[:, 0] - zl.grad).abs().max()) <= 2e-5 * float(zl.grad.abs().max())
    assert float((dh[:, 1:5] - rw.grad).abs().max()) <= 1e-4 * float(rw.grad.abs().max())


def test_fast_rcnn_loss_matches_the_reference_methods():
    """`eod_fast_rcnn_loss` (one cascade stage's sigmoid CE + class-agnostic L1 / smooth-L1 box loss, detic_fast_rcnn.py:157-303) against
    the fixture made by the reference's OWN `sigmoid_cross_entropy_loss` / `box_reg_loss` methods (gen_golden_losses.py::main_box), with
    and without a class-weight mask; then beta > 0 and LVIS width (1 203 classes, zero-shot logits padded to a 1 216-column row)
    against the oracle with autograd."""
    from embodied_object_detection_amd import ops
    from oracle import losses as OL
    from test_losses_golden import load_box_fixture
    dev = torch.device("cuda:0")
    z, t = load_box_fixture()
    C = int(z["num_classes"])
    weights = tuple(float(v) for v in z["box_weights"])
    for tag, cw in (("plain", None), ("fed", t("class_weight"))):
        losses, ds, dd = ops.fast_rcnn_loss(t("logits").to(dev), t("deltas").to(dev), t("proposal_boxes").to(dev), t("gt_boxes").to(dev),
                                            t("gt_classes").int().to(dev), C, weights, None if cw is None else cw.to(dev))
        assert abs(float(losses[0]) - float(z[f"{tag}_loss_cls"])) <= 1e-5 * float(z[f"{tag}_loss_cls"]), tag
        assert abs(float(losses[1]) - float(z[f"{tag}_loss_box_reg"])) <= 1e-5 * float(z[f"{tag}_loss_box_reg"]), tag
        assert float((ds.cpu() - t(f"{tag}_grad_logits")).abs().max()) <= 1e-7, tag
        assert float((dd.cpu() - t(f"{tag}_grad_deltas")).abs().max()) <= 1e-7, tag
    # LVIS width, padded rows, smooth-L1 with beta > 0
    g = torch.Generator().manual_seed(81)
    B, C, ld = 512, 1203, 1216
    logits = torch.randn((B, ld), generator=g) * 4
    gt = torch.randint(0, C + 1, (B,), generator=g)
    gt[::4] = C
    xy = torch.rand((B, 2), generator=g) * 500
    prop = torch.cat([xy, xy + torch.rand((B, 2), generator=g) * 200 + 4], dim=1)
    gtb = prop + (torch.rand((B, 4), generator=g) - 0.5) * 20
    gtb[:, 2:] = torch.maximum(gtb[:, 2:], gtb[:, :2] + 2)
    deltas = torch.randn((B, 4), generator=g) * 0.4
    cw = (torch.rand((C,), generator=g) < 0.05).float()
    zl = logits[:, :C + 1].clone().requires_grad_()
    dl = deltas.clone().requires_grad_()
    lc = OL.sigmoid_cross_entropy_loss(zl, gt, cw)
    lb = OL.box_reg_loss(prop, gtb, dl, gt, C, (30.0, 30.0, 15.0, 15.0), 0.3)
    (lc + lb).backward()
    losses, ds, dd = ops.fast_rcnn_loss(logits.to(dev), deltas.to(dev), prop.to(dev), gtb.to(dev), gt.int().to(dev), C,
                                        (30.0, 30.0, 15.0, 15.0), cw.to(dev), 0.3)
    assert abs(float(losses[0]) - lc.item()) <= 1e-5 * lc.item() and abs(float(losses[1]) - lb.item()) <= 1e-5 * lb.item()
    assert float((ds.cpu()[:, :C + 1] - zl.grad).abs().max()) <= 1e-7 and float(ds.cpu()[:, C:].abs().max()) == 0.0
    assert float((dd.cpu() - dl.grad).abs().max()) <= 1e-6


def test_centernet_targets_match_the_reference():
    """`eod_centernet_targets` against the reference's own `_get_ground_truth` / `_get_label_inds` (centernet.py:342-479; fixture by
    gen_golden_losses.py::main_targets): positive locations and regression targets bit for bit, the heatmap within 2 ulp (expf), the
    image without objects; then a 640x640 pyramid with 300 boxes (two LDS chunks) against the oracle; and the targets feed
    `eod_centernet_loss` directly."""
    import numpy as np
    from embodied_object_detection_amd import ops
    from oracle import losses as OL
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "centernet_targets.npz"))
    shapes = [tuple(x) for x in z["shapes"].tolist()]

    def compare(boxes, ref_pos, ref_reg, ref_heat, level_hw):
        heat, reg, pos, counts = ops.centernet_targets(boxes.to(dev), level_hw)
        n_pos, n_reg = counts.cpu().tolist()
        assert pos.cpu()[:n_pos].tolist() == list(ref_pos)
        assert torch.equal(reg.cpu(), ref_reg)
        assert n_reg == int((ref_reg.max(dim=1)[0] >= 0).sum())
        h = heat.cpu()
        assert torch.equal(h == 0, ref_heat == 0)
        assert float(((h - ref_heat).abs() / ref_heat.clamp(min=1e-4)).max()) <= 3e-7
        return heat, reg, pos, counts

    compare(torch.from_numpy(z["gt_boxes"]), z["full_pos_inds"].tolist(), torch.from_numpy(z["full_reg_targets"]),
            torch.from_numpy(z["full_heatmap"][:, 0]), shapes)
    compare(torch.zeros((0, 4)), [], torch.from_numpy(z["empty_reg_targets"]), torch.from_numpy(z["empty_heatmap"][:, 0]), shapes)
    # full-size pyramid, 300 boxes
    g = torch.Generator().manual_seed(91)
    shapes2 = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]
    xy = torch.rand((300, 2), generator=g) * 560
    wh = torch.cat([torch.rand((200, 2), generator=g) * 60 + 4, torch.rand((100, 2), generator=g) * 500 + 40])
    boxes = torch.cat([xy, (xy + wh).clamp(max=639.0)], dim=1)          # inside the image: a centre beyond it has no grid cell
    rp, rr, rh = OL.centernet_targets(boxes, shapes2)
    heat, reg, pos, counts = compare(boxes, rp.tolist(), rr, rh, shapes2)
    # the targets drive the loss kernel as they are
    n_pos, n_reg = counts.cpu().tolist()
    off = [0]
    for h, w in shapes2:
        off.append(off[-1] + h * w)
    head = torch.randn((off[-1], 8), generator=g)
    scales = [1.0, 0.9, 1.1, 1.2, 0.8]
    losses, dh = ops.CenterNetLoss(off, scales, dev)(head.to(dev), heat, reg, pos[:n_pos], max(float(n_pos), 1.0), max(float(n_reg), 1.0))
    hz = head.clone().requires_grad_()
    regp = torch.cat([torch.relu(hz[off[l]:off[l + 1], 1:5] * scales[l]) for l in range(5)])
    ref = OL.centernet_proposal_losses(hz[:, 0], regp, rh, rr, rp)
    sum(ref.values()).backward()
    for i, k in enumerate(("loss_centernet_loc", "loss_centernet_agn_pos", "loss_centernet_agn_neg")):
        assert abs(float(losses[i]) - ref[k].item()) <= 3e-5 * abs(ref[k].item()), k
    assert float((dh.cpu() - hz.grad).abs().max()) <= 1e-4 * float(hz.grad.abs().max())


def test_proposal_training_step_matches_autograd(synthetic_sd):
    """The proposal half of the reference's training forward (custom_rcnn.py:584-679 up to `proposal_losses`) on the HIP kernels
    (`modeling/training.py`): image -> ResNet-50 / FPN with the memory read fused in -> CenterNet head -> target assignment ->
    CenterNet.losses -> the gradient of EVERY parameter upstream (level scales, agn_hm / bbox_pred, 4 x tower conv + GroupNorm, P6 / P7,
    FPN, map_merge projections, 53 trunk convs), against torch autograd on the oracle's forward + the oracle's targets / losses.

    A ReLU whose pre-activation is within fp32 rounding of zero can be open in one implementation and closed in the other; that is a
    discrete difference of the gradient (first seen at seed 101: one element of tower layer 1 on P4, 2e-3 of the affected tensors).
    The test counts such flips over all ReLUs of trunk and head and asserts the tight tolerance on an input without any."""
    from embodied_object_detection_amd import build_model, setup_cfg
    from embodied_object_detection_amd.modeling.training import ProposalTraining
    from oracle import losses as OL
    dev = torch.device("cuda:0")
    cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5])
    model = build_model(cfg, synthetic_sd)
    step = ProposalTraining(model, synthetic_sd)
    H, W, n_cells = 128, 160, 500
    gt = torch.tensor([[10.0, 12.0, 60.0, 70.0], [40.0, 30.0, 150.0, 120.0], [90.0, 8.0, 118.0, 40.0], [5.0, 80.0, 44.0, 124.0],
                       [100.0, 60.0, 156.0, 126.0], [64.0, 64.0, 72.0, 72.0], [2.0, 2.0, 158.0, 126.0]])
    ocfg = M.OracleCfg(map_feature_weight=5.0)
    h = "proposal_generator.centernet_head"
    base = "backbone.bottom_up.base"
    rel = lambda a, b: float((a - b).abs().max()) / max(float(b.abs().max()), 1e-20)

    def run(seed):
        g = torch.Generator().manual_seed(seed)
        img = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8)
        mem16 = (torch.randn((n_cells, 512), generator=g) * 2).half()
        proj = torch.randint(0, n_cells, (H, W), generator=g)
        # ---- torch autograd on the oracle (the head restated with its pre-activations kept: centernet_head.py:141-161)
        trainable = lambda k, v: v.is_floating_point() and "running_" not in k and (k.startswith("backbone.") or "centernet_head" in k)
        sd = {k: (v.clone().float().requires_grad_() if trainable(k, v) else v) for k, v in synthetic_sd.items()}
        # the trunk restated with every post-ReLU activation kept (timm.py:277-299), then FPN + memory fusion + P6 / P7 of the oracle
        x = M.preprocess_image(img, ocfg)
        t = F.relu(M.frozen_bn(F.conv2d(x, sd[f"{base}.conv1.weight"], stride=2, padding=3), sd, f"{base}.bn1"))
        trunk_acts = [t.detach()]
        t = F.max_pool2d(t, kernel_size=3, stride=2, padding=1)
        cfeats = {}
        for li, nblk in enumerate((3, 4, 6, 3), start=1):
            for b in range(nblk):
                p = f"{base}.layer{li}.{b}"
                st = 2 if (b == 0 and li > 1) else 1
                o1 = F.relu(M.frozen_bn(F.conv2d(t, sd[f"{p}.conv1.weight"]), sd, f"{p}.bn1"))
                o2 = F.relu(M.frozen_bn(F.conv2d(o1, sd[f"{p}.conv2.weight"], stride=st, padding=1), sd, f"{p}.bn2"))
                o3 = M.frozen_bn(F.conv2d(o2, sd[f"{p}.conv3.weight"]), sd, f"{p}.bn3")
                sc = t
                if f"{p}.downsample.0.weight" in sd:
                    sc = M.frozen_bn(F.conv2d(t, sd[f"{p}.downsample.0.weight"], stride=st), sd, f"{p}.downsample.1")
                t = F.relu(o3 + sc)
                trunk_acts += [o1.detach(), o2.detach(), t.detach()]
            cfeats[f"layer{li + 1}"] = t
        feats = M.fuse_memory(M.fpn_top_down(cfeats, sd), M.memory_read_pooled(mem16, proj), sd, ocfg)
        feats = feats + M.top_block(feats[2], sd)
        shapes = [(f.shape[2], f.shape[3]) for f in feats]
        pre = [[] for _ in range(4)]
        agn, reg = [], []
        for l, f in enumerate(feats):
            t = f
            for i in range(4):
                t = F.group_norm(F.conv2d(t, sd[f"{h}.bbox_tower.{3 * i}.weight"], sd[f"{h}.bbox_tower.{3 * i}.bias"], padding=1), 32,
                                 sd[f"{h}.bbox_tower.{3 * i + 1}.weight"], sd[f"{h}.bbox_tower.{3 * i + 1}.bias"], eps=1e-5)
                pre[i].append(t.detach()[0].permute(1, 2, 0).reshape(-1, 256))
                t = F.relu(t)
            agn.append(F.conv2d(t, sd[f"{h}.agn_hm.weight"], sd[f"{h}.agn_hm.bias"], padding=1))
            reg.append(F.relu(F.conv2d(t, sd[f"{h}.bbox_pred.weight"], sd[f"{h}.bbox_pred.bias"], padding=1) * sd[f"{h}.scales.{l}.scale"]))
        logits = torch.cat([a.permute(0, 2, 3, 1).reshape(-1) for a in agn])
        reg_pred = torch.cat([r.permute(0, 2, 3, 1).reshape(-1, 4) for r in reg])
        pos, reg_t, heat = OL.centernet_targets(gt, shapes)
        assert pos.numel() >= 7 and int((reg_t.max(dim=1)[0] >= 0).sum()) >= 20
        ref = OL.centernet_proposal_losses(logits, reg_pred, heat, reg_t, pos)
        sum(ref.values()).backward()
        # ---- the HIP step
        losses, grads = step.forward_backward(img.to(dev), gt.to(dev), memory=(mem16.to(dev), proj.int().to(dev)))
        torch.cuda.synchronize()
        flips = 0
        for i in range(4):
            p = torch.cat(pre[i])
            mism = (step.last["keep"][i][3].cpu() > 0) != (p > 0)
            flips += int(mism.sum())
            assert float(p[mism].abs().max()) < 1e-4 if mism.any() else True          # only knife-edge elements may differ
        saved = step.last["saved"]
        prod_acts = [saved["stem"][0]] + [a for blk in saved["blocks"] for a in (blk[1], blk[2], blk[3])]
        assert len(prod_acts) == len(trunk_acts)
        for pa, ra in zip(prod_acts, trunk_acts):
            flips += int(((pa.cpu() > 0) != (ra[0].permute(1, 2, 0).unsqueeze(0) > 0)).sum())
        for k in ref:
            assert abs(float(losses[k]) - ref[k].item()) <= 1e-4 * abs(ref[k].item()), (k, float(losses[k]), ref[k].item())
        err = {}
        packed = lambda w: w.grad.permute(0, 2, 3, 1).reshape(w.shape[0], -1)

        def check(name, got, want):
            err[name] = rel(got.cpu(), want)

        check("scales", grads["scales"], torch.stack([sd[f"{h}.scales.{l}.scale"].grad.reshape(()) for l in range(5)]))
        for name in ("agn_hm", "bbox_pred"):
            check(name + ".w", grads[name][0], packed(sd[f"{h}.{name}.weight"]))
            check(name + ".b", grads[name][1], sd[f"{h}.{name}.bias"].grad)
        for i in range(4):
            dw, db = grads[f"bbox_tower.{3 * i}"]
            dga, dbe = grads[f"bbox_tower.{3 * i}.norm"]
            check(f"tower{i}.w", dw, packed(sd[f"{h}.bbox_tower.{3 * i}.weight"]))
            check(f"tower{i}.b", db, sd[f"{h}.bbox_tower.{3 * i}.bias"].grad)
            check(f"tower{i}.gamma", dga, sd[f"{h}.bbox_tower.{3 * i + 1}.weight"].grad)
            check(f"tower{i}.beta", dbe, sd[f"{h}.bbox_tower.{3 * i + 1}.bias"].grad)
        for l in (3, 4, 5):
            for kind in ("lateral", "output"):
                check(f"fpn_{kind}{l}.w", grads[f"fpn_{kind}{l}"][0], packed(sd[f"backbone.fpn_{kind}{l}.weight"]))
                check(f"fpn_{kind}{l}.b", grads[f"fpn_{kind}{l}"][1], sd[f"backbone.fpn_{kind}{l}.bias"].grad)
        for name in ("p6", "p7"):
            check(name + ".w", grads[name][0], packed(sd[f"backbone.top_block.{name}.weight"]))
        for i in (1, 2, 3):
            dW, db = grads[f"map_merge_projection{i}"]
            check(f"map_merge{i}.w", dW, sd[f"backbone.map_merge_projection{i}.weight"].grad.reshape(256, 512))
            check(f"map_merge{i}.b", db, sd[f"backbone.map_merge_projection{i}.bias"].grad)
        n = 0
        for li, nblk in enumerate((3, 4, 6, 3), start=1):
            for b in range(nblk):
                p = f"{base}.layer{li}.{b}"
                pairs = [(f"{p}.conv{i}", f"{p}.conv{i}.weight", f"{p}.bn{i}") for i in (1, 2, 3)]
                if f"{p}.downsample.0.weight" in sd:
                    pairs.append((f"{p}.downsample", f"{p}.downsample.0.weight", f"{p}.downsample.1"))
                for name, wk, bnp in pairs:
                    scale = (sd[f"{bnp}.weight"] / torch.sqrt(sd[f"{bnp}.running_var"] + 1e-5)).detach()
                    check(name + ".w", grads[name][0].cpu() * scale[:, None], packed(sd[wk]))
                    check(name + ".b", grads[name][1], sd[f"{bnp}.bias"].grad)
                    n += 1
        assert n == 52
        return losses, err, flips

    seen = []
    for seed in (105, 106, 107, 108, 109, 110):           # measured: seeds 102 / 103 / 104 have 1 / 3 / 2 flips (errors 2e-3 .. 2e-2), 105 none
        losses, err, flips = run(seed)
        seen.append((seed, flips, "%.1e" % max(err.values())))
        if flips == 0:
            break
    print("seeds (seed, ReLU flips in trunk + head, worst error):", seen)
    assert flips == 0, seen
    bad = {k: "%.1e" % v for k, v in err.items() if v > 1e-4}
    assert not bad, bad
    print("proposal training step (seed %d): losses %s; worst relative gradient error %.2e (%s) over %d tensors"
          % (seed, {k: round(float(v), 5) for k, v in losses.items()}, max(err.values()), max(err, key=err.get), len(err)))


def test_proposal_trainer_steps_like_torch_and_reduces_the_loss(synthetic_sd):
    """`ProposalTrainer` (modeling/training.py): the reference's AdamW set-up over every parameter upstream of the proposal losses,
    applied to the layers the inference path runs.  (1) after ONE step the parameters equal those of torch.optim.AdamW (+ detectron2's
    clip by value) stepping the oracle's raw parameters on autograd's gradients, with the learning rates `build_custom_optimizer`
    assigns (custom_solver.py:19-79: backbone multiplier, map_merge x 10); (2) further steps on the same frame reduce the loss, and
    the inference path -- which shares the updated layers -- still runs."""
    from embodied_object_detection_amd import build_model, setup_cfg, solver
    from embodied_object_detection_amd.modeling.training import ProposalTrainer
    from oracle import losses as OL
    dev = torch.device("cuda:0")
    lr = 2e-5
    cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5,
                           "SOLVER.BASE_LR", lr, "FP16", False])
    sd0 = {k: v.clone() for k, v in synthetic_sd.items()}
    model = build_model(cfg, sd0)
    trainer = ProposalTrainer(model, sd0)
    H, W, n_cells = 128, 160, 500
    g = torch.Generator().manual_seed(105)                      # the flip-free input of the test above
    img = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8)
    mem16 = (torch.randn((n_cells, 512), generator=g) * 2).half()
    proj = torch.randint(0, n_cells, (H, W), generator=g)
    gt = torch.tensor([[10.0, 12.0, 60.0, 70.0], [40.0, 30.0, 150.0, 120.0], [90.0, 8.0, 118.0, 40.0], [5.0, 80.0, 44.0, 124.0],
                       [100.0, 60.0, 156.0, 126.0], [64.0, 64.0, 72.0, 72.0], [2.0, 2.0, 158.0, 126.0]])
    # ---- torch: autograd on the oracle, AdamW with the reference's groups
    ocfg = M.OracleCfg(map_feature_weight=5.0)
    trainable = lambda k, v: v.is_floating_point() and "running_" not in k and ".bn" not in k and ".downsample.1." not in k and \
        (k.startswith("backbone.") or "centernet_head" in k)
    sd = {k: (v.clone().float().requires_grad_() if trainable(k, v) else v) for k, v in synthetic_sd.items()}
    feats = M.backbone_forward(M.preprocess_image(img, ocfg), sd, ocfg, mem16, proj)
    agn, reg = M.centernet_head(feats, sd)
    shapes = [(f.shape[2], f.shape[3]) for f in feats]
    pos, reg_t, heat = OL.centernet_targets(gt, shapes)
    ref = OL.centernet_proposal_losses(torch.cat([a.permute(0, 2, 3, 1).reshape(-1) for a in agn]),
                                       torch.cat([r.permute(0, 2, 3, 1).reshape(-1, 4) for r in reg]), heat, reg_t, pos)
    sum(ref.values()).backward()
    named = [(k, v) for k, v in sd.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None]
    groups = solver.param_groups_from_cfg(cfg, named)
    assert {g_["lr"] for g_ in groups if "map_merge" in g_["name"]} == {lr * 10.0}
    opt = torch.optim.AdamW([{"params": [g_["param"]], "lr": g_["lr"]} for g_ in groups], lr=lr, weight_decay=float(cfg.SOLVER.WEIGHT_DECAY))
    for _, v in named:
        v.grad.clamp_(-float(cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE), float(cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE))
    opt.step()
    # ---- the HIP trainer, one step
    mem = (mem16.to(dev), proj.int().to(dev))
    l0 = trainer.step(img.to(dev), gt.to(dev), memory=mem)
    first = sum(float(v) for v in l0.values())
    assert abs(first - sum(v.item() for v in ref.values())) <= 1e-4 * first
    packed = lambda w: w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    stepped = {n: t for n, t, _ in trainer.entries}
    h = "proposal_generator.centernet_head"
    checked = 0
    for name, tensor in stepped.items():
        if name == f"{h}.scales":
            want = torch.stack([sd[f"{h}.scales.{l}.scale"].detach().reshape(()) for l in range(5)])
        elif name.endswith("conv1.weight") and "layer" not in name:
            want = F.pad(sd[name].detach().permute(0, 2, 3, 1), (0, 1)).reshape(64, -1)               # the stem's 4-channel tap layout
        elif sd[name].dim() == 4 and "map_merge" not in name:
            want = packed(sd[name])
        else:
            want = sd[name].detach().reshape(tensor.shape)
        group_lr = next(g_["lr"] for g_ in trainer.groups if g_["name"] == name)
        diff = (tensor.cpu().reshape(want.shape) - want).abs()
        # the first AdamW step moves every weight by lr * g / (|g| + eps): equal gradients give equal steps; an element whose gradient
        # is within rounding of zero may step the other way (2 lr)
        assert float(diff.max()) <= 2.01 * group_lr, (name, float(diff.max()), group_lr)
        assert float(diff.mean()) <= 2e-3 * group_lr, (name, float(diff.mean()), group_lr)
        checked += 1
    assert checked == len(trainer.entries) == 53 + 12 + 4 + 6 + 16 + 4 + 1
    # the trunk's re-folded layer really carries the stepped master
    c = model.backbone.bottom_up.blocks[5][2]
    bnp = c.name.rsplit(".conv", 1)[0] + ".bn" + c.name[-1]
    scale = (synthetic_sd[f"{bnp}.weight"] / torch.sqrt(synthetic_sd[f"{bnp}.running_var"] + 1e-5)).view(-1, 1)
    assert float((c.w.cpu() - stepped[c.name + ".weight"].cpu() * scale).abs().max()) <= 1e-6
    # ---- more steps on the same frame: the loss goes down; the inference path runs on the updated layers
    last = first
    for _ in range(7):
        last = sum(float(v) for v in trainer.step(img.to(dev), gt.to(dev), memory=mem).values())
    assert last < first, (first, last)
    from embodied_object_detection_amd.data.synthetic import SyntheticSequence
    out = model([[SyntheticSequence(0, H=H, W=W, n_frames=1).frame(0)]])
    assert len(out) == 1 and "instances" in out[0]
    print("proposal trainer: total loss %.4f -> %.4f after 8 steps at lr %.0e" % (first, last, lr))


def test_four_channel_weight_gradient_kernels_match_autograd():
    """Weight gradient of 4-channel (tap layout) layers against autograd: the stem's 7x7 stride-2 form on the matrix cores (odd image
    sizes, two images, position ranges through the workspace and the single-range form of the plain entry point) and the scalar
    fallback for kernel rows wider than 8 taps."""
    import ctypes as C
    from embodied_object_detection_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(111)
    for (kh, k, stride, pad, H, W) in ((7, 7, 2, 3, 37, 45), (7, 7, 2, 3, 128, 160), (3, 9, 1, 1, 20, 24)):
        w = (torch.randn((64, 3, kh, k), generator=g) * 0.1).requires_grad_()
        b = torch.zeros(64, requires_grad=True)
        x = torch.randn((2, 3, H, W), generator=g)
        y = F.conv2d(x, w, b, stride=stride, padding=pad)
        go = torch.randn(y.shape, generator=g)
        (y * go).sum().backward()
        ref = F.pad(w.grad.permute(0, 2, 3, 1), (0, 1)).reshape(64, -1)
        conv = ops.Conv(w.detach(), b.detach(), stride=stride, pad=pad, device=dev, cin_pad=4)
        x4 = F.pad(x.permute(0, 2, 3, 1), (0, 1)).contiguous().to(dev)
        gd = go.permute(0, 2, 3, 1).contiguous().to(dev)
        o = ops.ConvBackward(conv)(x4, None, gd, need_dx=False)
        assert float((o["dw"].cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (k, H, W)
        assert float((o["db"].cpu() - b.grad).abs().max()) <= 2e-5 * float(b.grad.abs().max()), (k, H, W)
        # the entry point without a workspace: one position range
        dw = torch.empty_like(o["dw"])
        db = torch.empty_like(o["db"])
        _lib.check(lib.eod_conv2d_backward_weights(x4.data_ptr(), gd.data_ptr(), 2, H, W, 4, 64, kh, k, pad, stride, dw.data_ptr(), db.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream), "wgrad")
        assert float((dw.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (k, H, W)


def test_weight_gradients_on_a_side_stream_give_the_same_gradients(synthetic_sd):
    """`ProposalTraining(..., side_stream=True)`: every dW / db launch on a second stream behind an event of the main one, joined at the
    end of `forward_backward` (`ops.ConvBackward`) -- a scheduling option (measured: no gain, off by default): the same kernels on the
    same operands give bitwise the gradients of the one-stream step."""
    from embodied_object_detection_amd import build_model, setup_cfg
    from embodied_object_detection_amd.modeling.training import ProposalTraining
    dev = torch.device("cuda:0")
    cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5])
    model = build_model(cfg, synthetic_sd)
    g = torch.Generator().manual_seed(12)
    H, W, n_cells = 128, 160, 300
    img = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8).to(dev)
    mem = ((torch.randn((n_cells, 512), generator=g) * 2).half().to(dev), torch.randint(0, n_cells, (H, W), generator=g).int().to(dev))
    gt = torch.tensor([[10.0, 12.0, 60.0, 70.0], [40.0, 30.0, 150.0, 120.0], [90.0, 8.0, 118.0, 40.0]]).to(dev)
    outs = []
    for side in (False, True):
        step = ProposalTraining(model, synthetic_sd, side_stream=side)
        step.pyramid_backward = False          # level by level on both sides: the side-stream form has no pyramid-mode launch
        losses, grads = step.forward_backward(img, gt, memory=mem)
        torch.cuda.synchronize()
        outs.append((losses, grads))
    (la, ga), (lb, gb) = outs
    assert set(ga) == set(gb) and all(torch.equal(la[k], lb[k]) for k in la)
    for k in ga:
        a, b = ga[k], gb[k]
        if torch.is_tensor(a):
            assert torch.equal(a, b), k
        else:
            assert all(torch.equal(x, y) for x, y in zip(a, b)), k


@pytest.mark.gpu
def test_weight_gradients_through_one_shared_workspace_repeat_bitwise():
    """The LDS-tiled weight-gradient kernel cuts the positions into ranges (blockIdx.z) whose partial tiles a second launch adds in
    range order.  Three layer shapes share ONE workspace, called in turn three times: every call gives bitwise the same dW / db
    (no atomics, a fixed order of summation) and they match autograd."""
    from embodied_object_detection_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(33)
    cases = [(1, 80, 80, 64, 64, 3), (2, 40, 56, 128, 96, 1), (1, 33, 47, 32, 160, 3)]
    layers = []
    for (N, H, W, Cin, Cout, k) in cases:
        w = (torch.randn((Cout, Cin, k, k), generator=g) * (0.5 / (Cin * k * k) ** 0.5)).requires_grad_()
        b = torch.zeros((Cout,)).requires_grad_()
        x = torch.randn((N, Cin, H, W), generator=g)
        go = torch.randn((N, Cout, H, W), generator=g)
        (F.conv2d(x, w, b, padding=k // 2) * go).sum().backward()
        conv = ops.Conv(w.detach(), b.detach(), stride=1, pad=k // 2, device=dev)
        assert ops._lib.load().eod_conv2d_backward_weights_workspace_bytes(N, H, W, Cin, Cout, k, k, k // 2, 1) > 0, \
            "the case is meant to run with several position ranges"
        layers.append((ops.ConvBackward(conv), x.permute(0, 2, 3, 1).contiguous().to(dev), go.permute(0, 2, 3, 1).contiguous().to(dev),
                       w.grad.permute(0, 2, 3, 1).reshape(Cout, -1), b.grad))
    first = {}
    for rep in range(3):
        for i, (bw, xd, gd, ref_dw, ref_db) in enumerate(layers):
            out = bw(xd, None, gd, need_dx=False)
            dw, db = out["dw"].cpu(), out["db"].cpu()
            if rep == 0:
                first[i] = (dw, db)
                for name, got, ref in (("dW", dw, ref_dw), ("db", db, ref_db)):
                    assert float((got - ref).abs().max()) <= 3e-5 * float(ref.abs().max()), (i, name)
            else:
                assert torch.equal(dw, first[i][0]) and torch.equal(db, first[i][1]), (rep, i)


@pytest.mark.gpu
def test_rotated_weights_of_many_layers_in_one_launch():
    """`ConvBackward.refresh_all` (`eod_conv_rotate_weights_multi`: the input-gradient convolutions' weights of all layers after an
    optimizer step) against the per-layer construction from the new weights; 30 layers = two launches."""
    from embodied_object_detection_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(8)
    shapes = [(64, 32, 3), (32, 64, 1), (96, 64, 3), (256, 256, 3), (64, 160, 5)] * 6
    bws = []
    for (Cout, Cin, k) in shapes:
        conv = ops.Conv(torch.randn((Cout, Cin, k, k), generator=g), torch.zeros((Cout,)), stride=1, pad=k // 2, device=dev)
        bw = ops.ConvBackward(conv)
        bw._dgrad_conv()
        bws.append(bw)
    never_built = ops.ConvBackward(ops.Conv(torch.randn((32, 32, 3, 3), generator=g), torch.zeros((32,)), stride=1, pad=1, device=dev))
    new_w = []
    for bw in bws:
        c = bw.conv
        w = torch.randn((c.Cout, c.Cin, c.KH, c.KW), generator=g)
        new_w.append(w)
        packed, _ = ops.pack_conv_weight(w)
        assert c.w.shape == packed.shape
        c.w.copy_(packed.to(dev))                               # the stepped weights, written in place
    ops.ConvBackward.refresh_all(bws + [never_built])
    assert never_built._flipped is None
    for bw, w in zip(bws, new_w):
        c = bw.conv
        fresh = ops.ConvBackward(ops.Conv(w, torch.zeros((c.Cout,)), stride=1, pad=c.pad, device=dev))._dgrad_conv()
        assert bw._flipped_of == (c.w.data_ptr(), c.w._version)
        assert torch.equal(bw._dgrad_conv().w, fresh.w), (c.Cout, c.Cin, c.KH)


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride", [(1, 20, 28, 256, 64, 1, 1), (2, 14, 14, 64, 64, 3, 1), (1, 20, 20, 128, 256, 3, 2),
                                                     (1, 16, 16, 64, 128, 1, 2), (1, 40, 40, 64, 256, 1, 1)])
def test_relu_backward_and_shortcut_add_ride_on_the_input_gradient_launch(N, H, W, Cin, Cout, k, stride):
    """`ConvBackward(..., dx_res=, dx_gate=)`: dx = relu'(gate) * (dX + res) out of the input-gradient convolution's epilogue
    (EodConvDesc.gate + res_mode 1) is bitwise the three-launch form conv -> torch add -> eod_relu_backward, for stride-1 layers, the
    zero-inserted stride-2 layers and a layer whose plan splits K; dW / db are untouched by the two arguments."""
    from embodied_object_detection_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(17)
    conv = ops.Conv(torch.randn((Cout, Cin, k, k), generator=g) * (0.5 / (Cin * k * k) ** 0.5), torch.randn((Cout,), generator=g),
                    stride=stride, pad=k // 2, device=dev)
    OH, OW = conv.out_hw(H, W)
    x = torch.randn((N, H, W, Cin), generator=g).to(dev)
    go = torch.randn((N, OH, OW, Cout), generator=g).to(dev)
    res = torch.randn((N, H, W, Cin), generator=g).to(dev)
    gate = torch.relu(torch.randn((N, H, W, Cin), generator=g)).to(dev)           # a ReLU output: zeros and positives
    bw = ops.ConvBackward(conv)
    plain = bw(x, None, go)
    want = torch.empty_like(plain["dx"])
    summed = plain["dx"] + res
    ops.check(ops._lib.load().eod_relu_backward(summed.data_ptr(), gate.data_ptr(), want.data_ptr(), summed.numel(), ops._stream()), "relu_backward")
    fused = bw(x, None, go, dx_res=res, dx_gate=gate)
    assert torch.equal(fused["dx"], want)
    assert bool((fused["dx"] == 0).any()) and bool((fused["dx"] != 0).any())
    assert torch.equal(fused["dw"], plain["dw"]) and torch.equal(fused["db"], plain["db"])
    only_gate = bw(x, None, go, dx_gate=gate)["dx"]
    assert torch.equal(only_gate, torch.where(gate > 0, plain["dx"], torch.zeros_like(plain["dx"])))
    assert torch.equal(bw(x, None, go, dx_res=res)["dx"], summed)


@pytest.mark.gpu
@pytest.mark.parametrize("Cin,Cout", [(256, 256), (256, 32), (64, 96)])
def test_level_shared_layer_backward_in_one_launch_per_gradient(Cin, Cout):
    """`ConvBackward(..., levels=)`: the weight gradient of a level-shared 3x3 layer summed over a five-level pyramid by ONE launch
    (`eod_conv2d_backward_weights_levels`) and its input gradient by one pyramid-mode launch, against torch autograd level by level
    (the tower / head convs of centernet_head.py:141-161 at the level sizes of a 640x640 frame, and odd sizes)."""
    from embodied_object_detection_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(44)
    shapes = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)] if Cin == 256 else [(13, 9), (7, 5), (3, 4)]
    off = [0]
    for (h, w) in shapes:
        off.append(off[-1] + h * w)
    wt = (torch.randn((Cout, Cin, 3, 3), generator=g) * (0.5 / (Cin * 9) ** 0.5)).requires_grad_()
    b = (torch.randn((Cout,), generator=g) * 0.1).requires_grad_()
    xs = [torch.randn((1, Cin, h, w), generator=g).requires_grad_() for (h, w) in shapes]
    gos = [torch.randn((1, Cout, h, w), generator=g) for (h, w) in shapes]
    sum((F.conv2d(x, wt, b, padding=1) * go).sum() for x, go in zip(xs, gos)).backward()
    conv = ops.Conv(wt.detach(), b.detach(), stride=1, pad=1, device=dev)
    xd = torch.cat([x.detach()[0].permute(1, 2, 0).reshape(-1, Cin) for x in xs]).contiguous().to(dev)
    gd = torch.cat([go[0].permute(1, 2, 0).reshape(-1, Cout) for go in gos]).contiguous().to(dev)
    out = ops.ConvBackward(conv)(xd, None, gd, levels=(off, shapes))
    ref_dx = torch.cat([x.grad[0].permute(1, 2, 0).reshape(-1, Cin) for x in xs])
    for name, got, ref in (("dW", out["dw"].cpu(), wt.grad.permute(0, 2, 3, 1).reshape(Cout, -1)), ("db", out["db"].cpu(), b.grad),
                           ("dX", out["dx"].cpu(), ref_dx)):
        scale = float(ref.abs().max())
        err = float((got - ref).abs().max())
        assert err <= 3e-5 * scale, f"{name}: {err:.3e} at scale {scale:.3e}"
