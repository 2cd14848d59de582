"""The batch convention of the C ABI (include/eod_hip.h, "Batches": BASELINE configs[4], B independent scenes in lock-step): every
entry point with a `batch` argument or descriptor field, called ONCE for B scenes laid back to back, must give -- bit for bit --
what B single-scene calls give on each scene's own buffers; counts of zero, ragged counts and batch = 1 included.  The single-scene
calls are themselves checked against the CPU oracle in tests/test_kernels_gpu.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from embodied_object_detection_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def gen(seed):
    return torch.Generator().manual_seed(seed)


def boxes_in(n, w, h, g, big=False):
    ctr = torch.rand((n, 2), generator=g) * torch.tensor([float(w), float(h)])
    size = torch.rand((n, 2), generator=g) * (min(w, h) * (0.9 if big else 0.4)) + 4
    return torch.cat([ctr - size / 2, ctr + size / 2], dim=1)


@pytest.mark.parametrize("B,counts", [(3, [40, 0, 17]), (4, [64, 64, 1, 33]), (1, [20])])
def test_linear_layers_over_scene_segments(dev, B, counts):
    """EodConvDesc.m_segments: B row lists back to back, a count per list; rows without work are not written."""
    from embodied_object_detection_amd import ops
    R, K, N = 64, 1024, 256
    g = gen(1)
    conv = ops.Conv(torch.randn((N, K, 1, 1), generator=g) * 0.05, torch.randn(N, generator=g), device=dev)
    x = torch.randn((B * R, 1, 1, K), generator=g).to(dev)
    cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
    for force_tile in (0, 6, 13):            # planner's choice (wave-split K), forced wave-split, forced 64x64 tiles
        out = torch.full((B * R, 1, 1, N), -7.0, device=dev)
        conv(x, B * R, 1, 1, relu=True, m_count=cnt, m_unit=1, m_segments=B, plan_rows=R, out=out, force_tile=force_tile)
        for b in range(B):
            ref = torch.full((R, 1, 1, N), -7.0, device=dev)
            conv(x[b * R:(b + 1) * R].contiguous(), R, 1, 1, relu=True, m_count=cnt[b:b + 1], m_unit=1, out=ref, force_tile=force_tile)
            c = counts[b]
            assert torch.equal(out[b * R:b * R + c], ref[:c]), (force_tile, b)
            assert bool((out[b * R + c:(b + 1) * R] == -7.0).all()), "rows beyond a list's count must stay untouched"
    # split-K slabs + reduce with segments
    out = torch.full((B * R, 1, 1, N), -7.0, device=dev)
    conv(x, B * R, 1, 1, m_count=cnt, m_unit=1, m_segments=B, plan_rows=R, out=out, force_tile=3, force_splitk=4)
    for b in range(B):
        ref = torch.full((R, 1, 1, N), -7.0, device=dev)
        conv(x[b * R:(b + 1) * R].contiguous(), R, 1, 1, m_count=cnt[b:b + 1], m_unit=1, out=ref, force_tile=3, force_splitk=4)
        assert torch.equal(out[b * R:(b + 1) * R], ref), b


def test_pyramid_conv_and_groupnorm_over_the_levels_of_a_batch(dev):
    """5 B level images as 5 B levels (level major over the scenes), planned like one scene: bitwise the per-scene calls."""
    from embodied_object_detection_amd import ops
    B = 3
    shapes = [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)]
    off = [0]
    for h, w in shapes:
        off.append(off[-1] + h * w)
    P = off[-1]
    g = gen(2)
    conv = ops.Conv(torch.randn((256, 256, 3, 3), generator=g) * 0.03, torch.randn(256, generator=g) * 0.1, pad=1, device=dev)
    gamma, beta = torch.rand(256, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
    scenes = [torch.randn((P, 256), generator=g).to(dev) for _ in range(B)]
    offB, shapesB = [0], []
    for l, (h, w) in enumerate(shapes):
        for _ in range(B):
            offB.append(offB[-1] + h * w)
            shapesB.append((h, w))
    xb = torch.cat([scenes[b][off[l]:off[l + 1]] for l in range(5) for b in range(B)]).contiguous()
    yb = conv(xb, 1, 0, 0, levels=(offB, shapesB), plan_rows=P)
    zb = ops.groupnorm_relu(yb, gamma, beta, offB, 256, ops.groupnorm_workspace(offB, dev))
    for b in range(B):
        y = conv(scenes[b], 1, 0, 0, levels=(off, shapes))
        z = ops.groupnorm_relu(y, gamma, beta, off, 256, ops.groupnorm_workspace(off, dev))
        for l in range(5):
            n = shapes[l][0] * shapes[l][1]
            lo = B * off[l] + b * n
            assert torch.equal(yb[lo:lo + n], y[off[l]:off[l + 1]]), (b, l)
            assert torch.equal(zb[lo:lo + n], z[off[l]:off[l + 1]]), (b, l)


@pytest.mark.parametrize("S", [7, 14])
def test_roi_align_over_the_images_of_a_batch(dev, S):
    from embodied_object_detection_amd import ops
    B, R, h3, w3 = 3, 24, 16, 20
    g = gen(3)
    feats = [[torch.randn((1, h3 >> l, w3 >> l, 256), generator=g).to(dev) for l in range(3)] for _ in range(B)]
    lv = [torch.cat([feats[b][l] for b in range(B)]).contiguous() for l in range(3)]
    boxes = torch.cat([boxes_in(R, w3 * 8, h3 * 8, g, big=True) for _ in range(B)]).to(dev)
    counts = [R, 0, 9]
    cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
    out = torch.full((B * R, S, S, 256), -3.0, device=dev)
    ops.roi_align(lv[0], lv[1], lv[2], h3, w3, 256, boxes, cnt, B * R, S, out=out, batch=B, boxes_per_image=R)
    singles = []
    for b in range(B):
        ref = ops.roi_align(feats[b][0], feats[b][1], feats[b][2], h3, w3, 256, boxes[b * R:(b + 1) * R].contiguous(), cnt[b:b + 1], R, S)
        singles.append(ref)
        assert torch.equal(out[b * R:b * R + counts[b]], ref[:counts[b]]), b
        assert bool((out[b * R + counts[b]:(b + 1) * R] == -3.0).all())
    # one compact list of global box indices over all images (eod_concat_lists)
    lists = torch.zeros((B, R), dtype=torch.int32)
    lc = [5, 0, 7]
    for b in range(B):
        lists[b, :lc[b]] = torch.randperm(counts[b] if counts[b] else R, generator=g)[:lc[b]].int() if lc[b] else 0
    glist, total = torch.zeros((B * R,), dtype=torch.int32, device=dev), torch.zeros((1,), dtype=torch.int32, device=dev)
    ops.concat_lists(lists.to(dev), torch.tensor(lc, dtype=torch.int32, device=dev), R, R, B, glist, total)
    assert int(total.item()) == sum(lc)
    want = [b * R + int(lists[b, k]) for b in range(B) for k in range(lc[b])]
    assert glist[:sum(lc)].cpu().tolist() == want
    out2 = ops.roi_align(lv[0], lv[1], lv[2], h3, w3, 256, boxes, total, B * R, S, box_rows=glist, batch=B, boxes_per_image=R)
    for i, gidx in enumerate(want):
        b, r = divmod(gidx, R)
        one = ops.roi_align(feats[b][0], feats[b][1], feats[b][2], h3, w3, 256, boxes[gidx:gidx + 1].contiguous(), None, 1, S)
        assert torch.equal(out2[i], one[0]), i


def test_centernet_proposals_of_a_batch(dev):
    from embodied_object_detection_amd import ops
    from oracle import model as M
    B = 3
    level_hw = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    off = [0]
    for h, w in level_hw:
        off.append(off[-1] + h * w)
    g = gen(4)
    heads = [torch.cat([torch.randn((off[-1], 1), generator=g) * 2 - 1, torch.randn((off[-1], 4), generator=g)], dim=1) for _ in range(B)]
    heads[1][:, 0] = -20.0                                          # a scene without a single candidate
    scales = [0.9, 1.0, 1.1, 1.2, 0.8]
    cap = 256 + 64
    mk = lambda batch: ops.ProposalDecoder(level_hw, M.FPN_STRIDES, scales, 0.05, 1000, 256, 0.9, cap=cap, device=dev, batch=batch)
    hb = torch.cat([heads[b][off[l]:off[l + 1]] for l in range(5) for b in range(B)]).contiguous().to(dev)
    bb, sb, cb = mk(B)(hb)
    for b in range(B):
        b1, s1, c1 = mk(1)(heads[b].to(dev))
        n = int(c1.item())
        assert int(cb[b].item()) == n and (n > 0) == (b != 1)
        assert torch.equal(bb[b * cap:b * cap + n], b1[:n]) and torch.equal(sb[b * cap:b * cap + n], s1[:n])


def test_box_head_glue_and_selection_of_a_batch(dev):
    """zs_classify (with the memory re-score and the cascade fusion tails), apply_deltas, fast_rcnn_inference (unique rows, groups),
    detector_postprocess and paste_masks with batch = B against B single calls."""
    from embodied_object_detection_amd import ops
    B, R, C1, topk, H, W = 3, 64, 21, 50, 96, 128
    g = gen(5)
    counts = [R, 23, 0]
    cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
    feat = torch.randn((B * R, 512), generator=g).to(dev)
    zs, zs_mem = torch.randn((512, C1), generator=g).to(dev), torch.randn((512, C1), generator=g).to(dev)
    ps = torch.rand((B * R,), generator=g).to(dev)
    prob = torch.rand((B * R, C1), generator=g).to(dev)
    probs = [prob[b * R:(b + 1) * R].clone() for b in range(B)]
    featn, mems = torch.zeros((B * R, 512), device=dev), torch.zeros((B * R, C1), device=dev)
    ops.zs_classify(feat, zs, prob, True, featn, cnt, R, C1, zs_mem=zs_mem, prop_scores=ps, mem_scores_out=mems, final_inv_stages=1 / 3, batch=B)
    for b in range(B):
        fn1, m1 = torch.zeros((R, 512), device=dev), torch.zeros((R, C1), device=dev)
        sl = slice(b * R, (b + 1) * R)
        ops.zs_classify(feat[sl].contiguous(), zs, probs[b], True, fn1, cnt[b:b + 1], R, C1, zs_mem=zs_mem, prop_scores=ps[sl].contiguous(),
                        mem_scores_out=m1, final_inv_stages=1 / 3)
        assert torch.equal(prob[sl], probs[b]) and torch.equal(featn[sl], fn1) and torch.equal(mems[sl], m1), b
    boxes = torch.cat([boxes_in(R, W, H, g) for _ in range(B)]).to(dev)
    deltas = (torch.randn((B * R, 4), generator=g) * 0.3).to(dev)
    ob = torch.zeros((B * R, 4), device=dev)
    ops.apply_deltas(deltas, 4, boxes, ob, cnt, R, (10.0, 10.0, 5.0, 5.0), True, float(W), float(H), batch=B)
    for b in range(B):
        sl = slice(b * R, (b + 1) * R)
        o1 = torch.zeros((R, 4), device=dev)
        ops.apply_deltas(deltas[sl].contiguous(), 4, boxes[sl].contiguous(), o1, cnt[b:b + 1], R, (10.0, 10.0, 5.0, 5.0), True, float(W), float(H))
        assert torch.equal(ob[sl], o1), b
    # selection: unique rows + detection groups, scene-local indices
    selB = ops.DetectionSelector(R, C1, topk, dev, unique=True, groups=True, batch=B)
    db, ds, dc, dr, dn = selB(ob, prob, cnt, float(W), float(H), 0.05, 0.5)
    singles = []
    for b in range(B):
        sl = slice(b * R, (b + 1) * R)
        s1 = ops.DetectionSelector(R, C1, topk, dev, unique=True, groups=True)
        b1, sc1, c1, r1, n1 = s1(ob[sl].contiguous(), prob[sl].contiguous(), cnt[b:b + 1], float(W), float(H), 0.05, 0.5)
        n = int(n1.item())
        singles.append((s1, n))
        assert int(dn[b].item()) == n and (n == 0) == (counts[b] == 0)
        t = slice(b * topk, b * topk + n)
        assert torch.equal(db[t], b1[:n]) and torch.equal(ds[t], sc1[:n]) and torch.equal(dc[t], c1[:n]) and torch.equal(dr[t], r1[:n])
        nu, ng = int(s1.uniq_count.item()), int(s1.rep_count.item())
        assert int(selB.uniq_count[b].item()) == nu and int(selB.rep_count[b].item()) == ng
        assert torch.equal(selB.uniq_rows[b * R:b * R + nu], s1.uniq_rows[:nu])
        assert torch.equal(selB.rep_list[t.start:t.start + ng], s1.rep_list[:ng]) and torch.equal(selB.rep_of[t], s1.rep_of[:n])
    # post-processing + paste, one workgroup / one grid slice per scene
    masks = torch.rand((B * topk, 28, 28), generator=g).to(dev)
    pb, psc, pc = torch.zeros((B * topk, 4), device=dev), torch.zeros((B * topk,), device=dev), torch.zeros((B * topk,), dtype=torch.int32, device=dev)
    src, pn = torch.zeros((B * topk,), dtype=torch.int32, device=dev), torch.zeros((B,), dtype=torch.int32, device=dev)
    ops.detector_postprocess(db, ds, dc, dn, topk, 1.0, 1.0, float(W), float(H), pb, psc, pc, src, pn, remap=selB.rep_of, batch=B)
    pasted = torch.zeros((B, topk, H, W), dtype=torch.uint8, device=dev)
    ops.paste_masks(masks, pb, src, pn, topk, H, W, 0.5, pasted, batch=B, prob_units=topk)
    for b in range(B):
        s1, n = singles[b]
        t = slice(b * topk, (b + 1) * topk)
        b1, sc1, c1 = torch.zeros((topk, 4), device=dev), torch.zeros((topk,), device=dev), torch.zeros((topk,), dtype=torch.int32, device=dev)
        src1, n1 = torch.zeros((topk,), dtype=torch.int32, device=dev), torch.zeros((1,), dtype=torch.int32, device=dev)
        ops.detector_postprocess(s1.boxes, s1.scores, s1.classes, s1.count, topk, 1.0, 1.0, float(W), float(H), b1, sc1, c1, src1, n1,
                                 remap=s1.rep_of)
        k = int(n1.item())
        assert int(pn[b].item()) == k
        assert torch.equal(pb[t][:k], b1[:k]) and torch.equal(psc[t][:k], sc1[:k]) and torch.equal(src[t][:k], src1[:k])
        one = torch.zeros((topk, H, W), dtype=torch.uint8, device=dev)
        ops.paste_masks(masks[t].contiguous(), b1, src1, n1, topk, H, W, 0.5, one)
        assert torch.equal(pasted[b, :k], one[:k]), b


def test_memory_read_of_a_batch(dev):
    """gather + pooling and projection + fusion with grid.y = scene; `feats` level major over the scenes."""
    from embodied_object_detection_amd import ops
    B, H, W, N = 3, 64, 96, 500
    g = gen(6)
    mem = (torch.randn((B, N, 512), generator=g) * 2).half().to(dev)
    proj = torch.randint(0, N, (B, H, W), generator=g).int().to(dev)
    rows = [(H >> (3 + l)) * (W >> (3 + l)) for l in range(3)]
    off = [0, rows[0], rows[0] + rows[1], sum(rows)]
    projector = ops.MemoryProjector([torch.randn((256, 512, 1, 1), generator=g) * 0.05 for _ in range(3)],
                                    [torch.randn(256, generator=g) * 0.1 for _ in range(3)], dev)
    scene_feats = [torch.randn((off[3], 256), generator=g).to(dev) for _ in range(B)]
    fb = torch.cat([scene_feats[b][off[l]:off[l + 1]] for l in range(3) for b in range(B)]).contiguous()
    err = torch.zeros((1,), dtype=torch.int32, device=dev)
    pr = ops.pooled_rows(H, W)
    # the pooled buffer pads every level to whole 32-row operand tiles; the padding is never written: start both from zeros
    pooled = ops.memory_gather_pool(mem, proj, H, W, err=err, torch_order=True, batch=B,
                                    out=torch.zeros((B * pr, 512), dtype=torch.float16, device=dev))
    projector(pooled, fb, H, W, 5.0, "sum", batch=B)
    for b in range(B):
        p1 = ops.memory_gather_pool(mem[b], proj[b], H, W, err=err, torch_order=True, out=torch.zeros((pr, 512), dtype=torch.float16, device=dev))
        assert torch.equal(pooled[b * pr:(b + 1) * pr], p1), b
        f1 = scene_feats[b].clone()
        projector(p1, f1, H, W, 5.0, "sum")
        for l in range(3):
            lo = B * off[l] + b * rows[l]
            assert torch.equal(fb[lo:lo + rows[l]], f1[off[l]:off[l + 1]]), (b, l)
    assert int(err.item()) == 0


def test_memory_write_of_a_batch(dev):
    """B independent states in three launches: each scene's memory, counters and fp16 snapshot are those of its own write; a scene
    without instances keeps its state."""
    from embodied_object_detection_amd import ops
    B, H, W, N, R = 3, 64, 96, 300, 40
    g = gen(7)
    boxes = torch.cat([boxes_in(R, W, H, g) for _ in range(B)]).to(dev)
    masks = torch.rand((B * R, 28, 28), generator=g).to(dev)
    featn = (50 * F.normalize(torch.randn((B * R, 512), generator=g), dim=1)).to(dev)
    proj = torch.randint(0, N, (B, H, W), generator=g).int().to(dev)
    rows = torch.randint(0, R, (B, 100), generator=g).int().to(dev)
    counts = [12, 0, 31]
    cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
    mem0 = torch.randn((B, N, 512), generator=g).to(dev)
    obs0 = torch.randint(0, 3, (B, N), generator=g).float().to(dev)
    wB = ops.MemoryWriter(H, W, N, 100, R, dev, batch=B)
    w1 = ops.MemoryWriter(H, W, N, 100, R, dev)
    err = torch.zeros((1,), dtype=torch.int32, device=dev)
    for rep in range(2):                    # the second call starts from the per-frame tables the first one left
        mem, obs = mem0.clone(), obs0.clone()
        snap = ops.memory_normalize_f16(mem.view(-1, 512), obs.view(-1)).view(B, N, 512)
        k_out = wB(featn, boxes, masks, rows.view(-1), cnt, proj, mem, obs, err=err, snapshot=snap)
        for b in range(B):
            m1, o1 = mem0[b].clone(), obs0[b].clone()
            s1 = ops.memory_normalize_f16(m1, o1)
            sl = slice(b * R, (b + 1) * R)
            k1 = w1(featn[sl].contiguous(), boxes[sl].contiguous(), masks[sl].contiguous(), rows[b].contiguous(), cnt[b:b + 1], proj[b].contiguous(),
                    m1, o1, err=err, snapshot=s1)
            assert int(k_out[b].item()) == int(k1.item())
            assert torch.equal(mem[b], m1) and torch.equal(obs[b], o1) and torch.equal(snap[b], s1), (rep, b)
        assert torch.equal(mem[1], mem0[1]) and torch.equal(obs[1], obs0[1]), "a write without instances leaves the state untouched"
        # dirty-mark form
        mem, obs = mem0.clone(), obs0.clone()
        dirty = torch.zeros((B, N), dtype=torch.int32, device=dev)
        wB(featn, boxes, masks, rows.view(-1), cnt, proj, mem, obs, dirty=dirty, err=err)
        for b in range(B):
            m1, o1, d1 = mem0[b].clone(), obs0[b].clone(), torch.zeros((N,), dtype=torch.int32, device=dev)
            sl = slice(b * R, (b + 1) * R)
            w1(featn[sl].contiguous(), boxes[sl].contiguous(), masks[sl].contiguous(), rows[b].contiguous(), cnt[b:b + 1], proj[b].contiguous(), m1,
               o1, dirty=d1, err=err)
            assert torch.equal(mem[b], m1) and torch.equal(obs[b], o1) and torch.equal(dirty[b], d1), (rep, b)
    assert int(err.item()) == 0


def test_batch_arguments_are_checked(dev):
    """More scenes than EOD_MAX_BATCH, segment counts without a count pointer, a row capacity that is not a multiple of the batch:
    refused with EOD_ERR_BAD_DIMS before anything is launched."""
    from embodied_object_detection_amd import _lib, ops
    g = gen(8)
    conv = ops.Conv(torch.randn((64, 64, 1, 1), generator=g), None, device=dev)
    x = torch.randn((18, 1, 1, 64), generator=g).to(dev)
    cnt = torch.zeros((9,), dtype=torch.int32, device=dev)
    with pytest.raises(_lib.EodError):
        conv(x, 18, 1, 1, m_count=cnt, m_unit=1, m_segments=9)              # > EOD_MAX_BATCH
    with pytest.raises(_lib.EodError):
        conv(x, 18, 1, 1, m_segments=2)                                      # segments need counts
    with pytest.raises(_lib.EodError):
        conv(x, 18, 1, 1, m_count=cnt, m_unit=1, m_segments=4)              # 18 rows are not 4 equal lists
    p = torch.zeros((2, 8, 8, 256), device=dev)
    with pytest.raises(_lib.EodError):
        ops.roi_align(p, p, p, 8, 8, 256, torch.zeros((10, 4), device=dev), cnt, 10, 7, batch=2, boxes_per_image=4)   # 10 != 2 x 4
    with pytest.raises(_lib.EodError):
        ops.concat_lists(cnt, cnt, 4, 4, 9, cnt, cnt)
