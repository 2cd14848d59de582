"""Full-size checks (BASELINE.json configs 3 and 5) through size-independent properties, plus one oracle frame at 960x960."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import memory as OM
from oracle import model as M
from oracle import ops as OO


def _cfg(**over):
    from embodied_object_detection_amd import setup_cfg
    opts = ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5]
    for k, v in over.items():
        opts += [k, v]
    return setup_cfg(None, opts)


def _run(model, frames):
    outs = []
    for f in frames:
        inst = model([[f]])[0]["instances"]
        outs.append((inst.pred_boxes.tensor.cpu(), inst.scores.cpu(), inst.pred_classes.cpu(), int(model.last_stats["mem_k"].item())))
    return outs


def test_640_properties(synthetic_sd):
    from embodied_object_detection_amd import build_model
    from embodied_object_detection_amd.data.synthetic import SyntheticSequence
    seq = SyntheticSequence(7, H=640, W=640, n_frames=3)              # grid 200x200 -> N = 40 000 (config 3)
    frames = [seq.frame(i) for i in range(3)]
    model = build_model(_cfg(), synthetic_sd)
    a = _run(model, frames)
    mem_a, obs_a = model.implicit_memory.cpu().clone(), model.observations.cpu().clone()
    # capacities and ordering
    for boxes, scores, classes, k in a:
        assert len(scores) <= 300 and 0 < k <= 100
        assert bool((scores[:-1] >= scores[1:]).all()), "detections are sorted by score"
        assert bool((boxes[:, 0] >= 0).all() and (boxes[:, 2] <= 640).all() and (boxes[:, 3] <= 640).all())
        assert int(classes.min()) >= 0 and int(classes.max()) < 20
    # observation counters: +1 per frame on exactly the cells any pixel of that frame projects to (custom_rcnn.py:699-701)
    expect = torch.zeros(seq.n_cells)
    for f in frames:
        expect[torch.from_numpy(np.unique(f["proj_indices"])).long()] += 1
    assert torch.equal(obs_a, expect)
    # only cells that some pixel hits can hold features
    written = mem_a.abs().sum(dim=1) > 0
    assert bool((expect[written] > 0).all()) and int(written.sum()) > 0
    # bitwise reproducibility (fixed-point atomics, deterministic split-K): same frames from reset -> same bits
    b = _run(model, frames)
    for (b1, s1, c1, k1), (b2, s2, c2, k2) in zip(a, b):
        assert torch.equal(b1, b2) and torch.equal(s1, s2) and torch.equal(c1, c2) and k1 == k2
    assert torch.equal(model.implicit_memory.cpu(), mem_a) and torch.equal(model.observations.cpu(), obs_a)
    # linearity of the fusion: weight 0 leaves the pyramid untouched -> identical to MEMORY_TYPE image_only
    m0 = build_model(_cfg(**{"MODEL.MAP_FEATURE_WEIGHT": 0}), synthetic_sd)
    mi = build_model(_cfg(**{"MODEL.MEMORY_TYPE": "image_only"}), synthetic_sd)
    for (b1, s1, c1, _), (b2, s2, c2, _) in zip(_run(m0, frames[:2]), _run(mi, frames[:2])):
        assert torch.equal(b1, b2) and torch.equal(s1, s2) and torch.equal(c1, c2)
    # the explicit semantic map (a20) of the final memory against the oracle
    labels = model.semantic_map().cpu()
    ref = OM.semmap_labels(mem_a, obs_a, model.zs_weight.cpu(), 0.4)
    assert (labels == ref).float().mean().item() > 0.999


def test_960_config5_frame_matches_oracle(synthetic_sd):
    """Config 5 geometry: 960x960 frame, 512x512 memory grid (262 144 cells, 0.08 m): level 0 has 14 400 positions (16 384-key
    sort path), the write workspace is sized for 115 200 selected pixels."""
    from embodied_object_detection_amd import build_model
    from embodied_object_detection_amd.data.synthetic import SyntheticSequence
    seq = SyntheticSequence(11, H=960, W=960, n_frames=2, map_w=512, map_h=512, cell=0.08)
    frames = [seq.frame(i) for i in range(2)]
    model = build_model(_cfg(), synthetic_sd)
    oracle = OM.RecurrentOracle(synthetic_sd, M.OracleCfg(map_feature_weight=5.0))
    f = frames[0]
    ref = oracle.step(f, 0, frames)["instances"]
    out = model([[f]])[0]["instances"]
    gb, gs, gc = out.pred_boxes.tensor.cpu(), out.scores.cpu(), out.pred_classes.cpu()
    n_ok = 0
    for b, s, c in zip(ref["pred_boxes"], ref["scores"], ref["pred_classes"]):
        cand = (gc == c).nonzero().squeeze(1)
        if cand.numel():
            iou = OO.iou_one_to_many(b, gb[cand])
            j = int(iou.argmax())
            n_ok += int(iou[j] > 0.99 and abs(float(gs[cand[j]] - s)) < 1e-3)
    assert n_ok / max(1, len(ref["scores"])) >= 0.97
    assert torch.equal(model.observations.cpu(), oracle.observations)
    cell_err = (model.implicit_memory.cpu() - oracle.implicit_memory).abs().max(dim=1).values
    assert (cell_err > 1e-2 * max(1.0, oracle.implicit_memory.abs().max().item())).float().mean().item() <= 0.02
    model([[frames[1]]])      # second frame reads the written memory at full size without faults


def _lockstep_model(kind, B, synthetic_sd):
    """"launches": N = B through every stage, one launch per stage (modeling/lockstep.py); "streams": B scene objects on B streams,
    only the trunk batched (modeling/batched.py)."""
    if kind == "launches":
        from embodied_object_detection_amd.modeling.lockstep import LockstepScenes
        return LockstepScenes(_cfg(), B, synthetic_sd)
    from embodied_object_detection_amd.modeling.batched import BatchedSequences
    return BatchedSequences(_cfg(), B, synthetic_sd)


def _batch_equals_singles(synthetic_sd, H, W, grid, cell, B, T, kind="streams", lengths=None):
    from embodied_object_detection_amd import build_model
    from embodied_object_detection_amd.data.synthetic import SyntheticSequence
    seqs = [SyntheticSequence(40 + b, H=H, W=W, n_frames=T, map_w=grid, map_h=grid, cell=cell) for b in range(B)]
    eps = [[s.frame(i) for i in range(T if lengths is None else lengths[b])] for b, s in enumerate(seqs)]
    batched = _lockstep_model(kind, B, synthetic_sd)
    outs = batched(eps)
    T = None
    assert len(outs) == B and all(len(o) == len(e) for o, e in zip(outs, eps))
    for b in range(B):
        single = build_model(_cfg(), synthetic_sd)
        ref = single([eps[b]])
        for t in range(len(eps[b])):
            a, r = outs[b][t]["instances"], ref[t]["instances"]
            assert torch.equal(a.pred_boxes.tensor, r.pred_boxes.tensor) and torch.equal(a.scores, r.scores), (b, t)
            assert torch.equal(a.pred_classes, r.pred_classes) and torch.equal(a.pred_masks, r.pred_masks), (b, t)
        assert torch.equal(batched.scenes[b].implicit_memory, single.implicit_memory), b
        assert torch.equal(batched.scenes[b].observations, single.observations), b
        del single
        torch.cuda.empty_cache()


def test_scenes_in_flight_do_not_change_results(synthetic_sd):
    """1, 2 or 3 scenes in flight, with or without the look-ahead trunk and the stagger: three scenes, bitwise the same outputs."""
    from embodied_object_detection_amd.data.synthetic import SyntheticSequence
    from embodied_object_detection_amd.modeling.batched import BatchedSequences
    seqs = [SyntheticSequence(90 + b, H=128, W=160, n_frames=4, map_w=24, map_h=24, cell=0.5) for b in range(3)]
    eps = [[s.frame(i) for i in range(4)] for s in seqs]
    ref = None
    for conc, look, stag in ((1, True, True), (2, True, True), (3, True, False), (2, False, True)):
        bs = BatchedSequences(_cfg(), 3, synthetic_sd, concurrent_scenes=conc)
        bs.trunk_lookahead, bs.stagger = look, stag
        outs = bs(eps)
        got = [[(o["instances"].pred_boxes.tensor.clone(), o["instances"].scores.clone(), o["instances"].pred_masks.clone()) for o in ob]
               for ob in outs] + [[m.implicit_memory.clone() for m in bs.scenes]]
        if ref is None:
            ref = got
            continue
        for a, b in zip(ref[:-1], got[:-1]):
            for (b1, s1, m1), (b2, s2, m2) in zip(a, b):
                assert torch.equal(b1, b2) and torch.equal(s1, s2) and torch.equal(m1, m2)
        assert all(torch.equal(x, y) for x, y in zip(ref[-1], got[-1]))
        del bs
        torch.cuda.empty_cache()


@pytest.mark.parametrize("kind", ["launches", "streams"])
def test_batch_of_3_equals_3_single_runs_small(synthetic_sd, kind):
    """Lock-step batch (planned like one image) == three independent runs, bit for bit."""
    _batch_equals_singles(synthetic_sd, 128, 160, 24, 0.5, 3, 3, kind)


def test_lockstep_longterm_two_calls_equal_single_runs(synthetic_sd):
    """MODEL.TEST_TYPE longterm (the fp16 snapshot is frozen after the first frame of a call: the write only marks rows) and a second
    call that continues the scenes: N = B through every stage still equals the single-scene runs bit for bit."""
    from embodied_object_detection_amd import build_model
    from embodied_object_detection_amd.data.synthetic import SyntheticSequence
    from embodied_object_detection_amd.modeling.lockstep import LockstepScenes
    cfg = lambda: _cfg(**{"MODEL.TEST_TYPE": "longterm"})
    B, T = 2, 4
    seqs = [SyntheticSequence(60 + b, H=128, W=160, n_frames=T, map_w=24, map_h=24, cell=0.5) for b in range(B)]
    eps = [[s.frame(i) for i in range(T)] for s in seqs]
    ls = LockstepScenes(cfg(), B, synthetic_sd)
    outs = [a + b for a, b in zip(ls([e[:2] for e in eps]), ls([e[2:] for e in eps]))]
    for b in range(B):
        single = build_model(cfg(), synthetic_sd)
        ref = single([eps[b][:2]]) + single([eps[b][2:]])
        for t in range(T):
            a, r = outs[b][t]["instances"], ref[t]["instances"]
            assert torch.equal(a.pred_boxes.tensor, r.pred_boxes.tensor) and torch.equal(a.scores, r.scores), (b, t)
            assert torch.equal(a.pred_masks, r.pred_masks), (b, t)
        assert torch.equal(ls.scenes[b].implicit_memory, single.implicit_memory) and torch.equal(ls.scenes[b].observations, single.observations)
        del single
        torch.cuda.empty_cache()


@pytest.mark.parametrize("over", [{}, {"MODEL.MEMORY_TYPE": "image_only"}, {"MODEL.MAP_FEAT_FUSION": "mem_only"}])
def test_lockstep_of_one_and_other_memory_modes_equal_single_runs(synthetic_sd, over):
    """B = 1 (the batch convention's degenerate case) and the other MEMORY_TYPE / fusion modes through LockstepScenes: bitwise the
    single-scene model's outputs and state (the write-back runs for every MEMORY_TYPE, custom_rcnn.py:515)."""
    from embodied_object_detection_amd import build_model
    from embodied_object_detection_amd.data.synthetic import SyntheticSequence
    from embodied_object_detection_amd.modeling.lockstep import LockstepScenes
    B = 1 if not over else 2
    seqs = [SyntheticSequence(30 + b, H=128, W=160, n_frames=3, map_w=24, map_h=24, cell=0.5) for b in range(B)]
    eps = [[s.frame(i) for i in range(3)] for s in seqs]
    ls = LockstepScenes(_cfg(**over), B, synthetic_sd)
    outs = ls(eps)
    for b in range(B):
        single = build_model(_cfg(**over), synthetic_sd)
        ref = single([eps[b]])
        for t in range(3):
            a, r = outs[b][t]["instances"], ref[t]["instances"]
            assert torch.equal(a.pred_boxes.tensor, r.pred_boxes.tensor) and torch.equal(a.scores, r.scores), (b, t)
            assert torch.equal(a.pred_classes, r.pred_classes) and torch.equal(a.pred_masks, r.pred_masks), (b, t)
        assert torch.equal(ls.scenes[b].implicit_memory, single.implicit_memory) and torch.equal(ls.scenes[b].observations, single.observations)
        del single
        torch.cuda.empty_cache()


def test_lockstep_schedules_are_bitwise_identical(synthetic_sd):
    """The step's streams (look-ahead trunk one step ahead, detection pass trailing under the next step) are a scheduling change only:
    with and without them, over two calls of 6 steps, every output and the final state of every scene are bitwise the same."""
    from embodied_object_detection_amd.data.synthetic import SyntheticSequence
    from embodied_object_detection_amd.modeling.lockstep import LockstepScenes
    B, T = 3, 12
    seqs = [SyntheticSequence(80 + b, H=128, W=160, n_frames=T, map_w=24, map_h=24, cell=0.5) for b in range(B)]
    eps = [[s.frame(i) for i in range(T)] for s in seqs]
    ref = None
    for look, trail in ((False, False), (True, True), (True, False), (False, True)):
        ls = LockstepScenes(_cfg(), B, synthetic_sd)
        ls.trunk_lookahead, ls.trail_detection_pass = look, trail
        outs = [a + b for a, b in zip(ls([e[:6] for e in eps]), ls([e[6:] for e in eps]))]
        got = [[(o["instances"].pred_boxes.tensor.clone(), o["instances"].scores.clone(), o["instances"].pred_classes.clone(),
                 o["instances"].pred_masks.clone()) for o in ob] for ob in outs]
        state = (ls.implicit_memory.clone(), ls.observations.clone())
        if ref is None:
            ref = (got, state)
        else:
            for a, b in zip(ref[0], got):
                assert len(a) == len(b) == T
                for x, y in zip(a, b):
                    assert all(torch.equal(u, v) for u, v in zip(x, y)), (look, trail)
            assert torch.equal(ref[1][0], state[0]) and torch.equal(ref[1][1], state[1]), (look, trail)
        del ls
        torch.cuda.empty_cache()


def test_lockstep_planned_for_the_batch_agrees_within_tolerance(synthetic_sd):
    """`plan_like_single = False`: the layers are planned for the rows the batched launches really have (other split-K / tile
    choices: the same fp32 arithmetic in another summation order).  Per frame -- the state before every frame set to the single-scene
    run's -- the detections agree with the single-scene run within the north-star tolerance (1e-3 px / 1e-3 score on matched
    detections, >= 98 % matched) and the observation counters exactly."""
    from embodied_object_detection_amd import build_model
    from embodied_object_detection_amd.data.synthetic import SyntheticSequence
    from embodied_object_detection_amd.modeling.lockstep import LockstepScenes
    B, T = 2, 3
    seqs = [SyntheticSequence(50 + b, H=128, W=160, n_frames=T, map_w=24, map_h=24, cell=0.5) for b in range(B)]
    eps = [[s.frame(i) for i in range(T)] for s in seqs]
    ls = LockstepScenes(_cfg(), B, synthetic_sd)
    ls.plan_like_single = False
    singles = [build_model(_cfg(), synthetic_sd) for _ in range(B)]
    for t in range(T):
        if t > 0:                                   # teacher forcing: one pass of the path per comparison
            for b in range(B):
                ls.implicit_memory[b].copy_(singles[b].implicit_memory)
                ls.observations[b].copy_(singles[b].observations)
            ls.invalidate_memory_snapshot()
        outs = ls([[dict(eps[b][t], memory_reset=eps[b][t]["memory_reset"] and t == 0)] for b in range(B)])
        for b in range(B):
            a, r = outs[b][0]["instances"], singles[b]([[eps[b][t]]])[0]["instances"]
            ab, rb = a.pred_boxes.tensor.cpu(), r.pred_boxes.tensor.cpu()
            matched = 0
            for i in range(rb.shape[0]):
                cand = (a.pred_classes.cpu() == r.pred_classes.cpu()[i]).nonzero().squeeze(1)
                if not cand.numel():
                    continue
                iou = OO.iou_one_to_many(rb[i], ab[cand])
                j = int(iou.argmax())
                if float(iou[j]) > 0.99:
                    matched += 1
                    assert float((ab[cand[j]] - rb[i]).abs().max()) < 1e-3 and abs(float(a.scores[cand[j]] - r.scores[i])) < 1e-3, (t, b, i)
            assert matched >= 0.98 * rb.shape[0] and abs(len(a) - len(r)) <= max(3, 0.02 * len(r)), (t, b, matched, len(a), len(r))
            assert torch.equal(ls.scenes[b].observations, singles[b].observations), (t, b)


def test_ragged_lockstep_batch_equals_single_runs(synthetic_sd):
    """Episodes of different lengths (one scene sits most of the call out as an idle slot): every scene still gets exactly its own
    run's results and state."""
    _batch_equals_singles(synthetic_sd, 128, 160, 24, 0.5, 3, 5, "launches", lengths=[5, 2, 4])


@pytest.mark.parametrize("kind", ["launches", "streams"])
def test_config5_batch_of_4_equals_4_single_runs(synthetic_sd, kind):
    """BASELINE.json configs[4]: 4 sequences batched per GPU at 960x960 with a 512x512 memory grid; detections, masks and memory
    state of every sequence are bitwise those of its own single-sequence run."""
    _batch_equals_singles(synthetic_sd, 960, 960, 512, 0.08, 4, 2, kind)


@pytest.mark.parametrize("kind", ["launches", "streams"])
def test_eval_loop_in_lockstep_gives_the_same_records(synthetic_sd, kind):
    """`inference_on_scenes` with a `BatchedSequences` of 2: three scenes of different lengths (25, 40 and 12 frames: ragged episodes,
    one scene without a partner) produce exactly the records of the one-scene-after-the-other loop (`train_mp3d.py:186`)."""
    from embodied_object_detection_amd import build_model
    from embodied_object_detection_amd.data.synthetic import SyntheticSequence
    from embodied_object_detection_amd.engine.eval_loop import episode_offsets, inference_on_scenes
    from embodied_object_detection_amd.modeling.batched import BatchedSequences
    lens = [25, 40, 12]
    mk = lambda: [SyntheticSequence(70 + s, H=128, W=160, n_frames=n, map_w=24, map_h=24, cell=0.5) for s, n in enumerate(lens)]
    offs = {70 + s: o for s, o in enumerate(episode_offsets(lens))}
    seq = inference_on_scenes(build_model(_cfg(), synthetic_sd), mk(), 0, max_rows=1 << 16, every=5, scene_episode_offset=offs)
    seen = []
    par = inference_on_scenes(_lockstep_model(kind, 2, synthetic_sd), mk(), 0, max_rows=1 << 16, every=5, scene_episode_offset=offs,
                              on_episode=lambda idx, inp, out: seen.append((idx, len(inp), len(out))))
    assert seq["frames"] == par["frames"] == sum(lens)
    assert sorted(seen) == [(0, 20, 20), (1, 5, 5), (2, 20, 20), (3, 20, 20), (4, 12, 12)]
    a, b = sorted(seq["records"].rows), sorted(par["records"].rows)
    assert len(a) == len(b) > 0
    assert a == b
