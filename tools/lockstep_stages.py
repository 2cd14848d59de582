#!/usr/bin/env python
"""Stage-by-stage comparison of `LockstepScenes` (N = B through every stage) with B single-scene models on the same frames: prints,
for every scene and frame, the first stage whose buffers differ (bitwise).  Debugging aid for modeling/lockstep.py.

    python tools/lockstep_stages.py [B] [T] [H] [W] [grid]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_object_detection_amd import build_model, setup_cfg
from embodied_object_detection_amd.checkpoint import synthetic_state_dict
from embodied_object_detection_amd.data.synthetic import SyntheticSequence
from embodied_object_detection_amd.modeling.lockstep import LockstepScenes

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2
H = int(sys.argv[3]) if len(sys.argv) > 3 else 128
W = int(sys.argv[4]) if len(sys.argv) > 4 else 160
G = int(sys.argv[5]) if len(sys.argv) > 5 else 24
cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5,
                       "MODEL.MEMORY_CLS_SCORE_THRESH", 0.3, "MODEL.DEVICE", "cuda:0"])
sd = synthetic_state_dict(0)
seqs = [SyntheticSequence(40 + b, H=H, W=W, n_frames=T, map_w=G, map_h=G, cell=0.5 if G < 200 else 0.2) for b in range(B)]
eps = [[s.frame(i) for i in range(T)] for s in seqs]
ls = LockstepScenes(cfg, B, sd)
ls.trunk_lookahead = False
ls.trail_detection_pass = False
singles = [build_model(cfg, sd) for _ in range(B)]
for s in singles:
    s.overlap_branches = False
    s.prefetch_trunk = False

ok_all = True
for t in range(T):
    outs = ls([[eps[b][t]] for b in range(B)])
    torch.cuda.synchronize()
    d = ls._bufs
    k = (ls._slot - 1) % 3
    sel = ls.selectors[k]
    which = ls._pyramid
    feats, views = d["pyr"][which]
    R, D = ls.R, ls.D
    for b in range(B):
        s = singles[b]
        ref = s([[eps[b][t]]])
        torch.cuda.synchronize()
        shapes, off, sfeats, sviews, _ = s.backbone._plan(H, W, s._pyramid)
        dec = s.proposal_generator._plans[next(iter(s.proposal_generator._plans))][3]
        rh = s.roi_heads
        n = int(dec.count.item())
        stages = []
        for l in range(5):
            stages.append((f"pyramid level {l + 3}", views[l][b], sviews[l][0]))
        stages += [
            ("proposal count", d["dec"].count[b], dec.count[0]),
            ("proposal boxes", d["dec"].boxes[b * R:b * R + n], dec.boxes[:n]),
            ("proposal scores", d["dec"].scores[b * R:b * R + n], dec.scores[:n]),
            ("stage-0 features", ls.feat0.view(-1, 512)[b * R:b * R + n], rh.feat0.view(-1, 512)[:n]),
            ("normalised features", ls.featn0[b * R:b * R + n], rh.featn0[:n]),
            ("memory scores", ls.mem_scores[b * R:b * R + n], s.mem_scores[:n]),
            ("cascade scores", ls.prob[b * R:b * R + n], rh.prob[:n]),
            ("cascade boxes", ls.boxes[-1][b * R:b * R + n], rh.boxes[-1][:n]),
            ("detection count", sel.count[b], rh.last_selector.count[0]),
        ]
        nd = int(rh.last_selector.count.item())
        stages += [
            ("detection boxes", sel.boxes[b * D:b * D + nd], rh.last_selector.boxes[:nd]),
            ("detection scores", sel.scores[b * D:b * D + nd], rh.last_selector.scores[:nd]),
            ("detection groups", sel.rep_of[b * D:b * D + nd], rh.last_selector.rep_of[:nd]),
            ("memory rows count", ls.mem_selector.count[b], s.mem_selector.count[0]),
            ("unique rows count", ls.mem_selector.uniq_count[b], s.mem_selector.uniq_count[0]),
        ]
        nu = int(s.mem_selector.uniq_count.item())
        ur = s.mem_selector.uniq_rows[:nu].long()
        stages += [
            ("unique rows", ls.mem_selector.uniq_rows[b * R:b * R + nu], s.mem_selector.uniq_rows[:nu]),
            ("proposal masks", ls.prop_masks[b * R:(b + 1) * R][ur], rh.prop_masks[ur]),
        ]
        reps = rh.last_selector.rep_list[:int(rh.last_selector.rep_count.item())].long()
        stages += [
            ("detection masks", ls.det_masks[b * D:(b + 1) * D][reps], rh.det_masks[reps]),
            ("observations", ls.observations[b], s.observations),
            ("memory", ls.implicit_memory[b], s.implicit_memory),
            ("fp16 snapshot", ls._mem_f16[b], s._mem_f16),
        ]
        a, r = outs[b][0]["instances"], ref[0]["instances"]
        stages += [("output boxes", a.pred_boxes.tensor, r.pred_boxes.tensor), ("output scores", a.scores, r.scores),
                   ("output classes", a.pred_classes, r.pred_classes), ("output masks", a.pred_masks, r.pred_masks)]
        first = None
        for name, x, y in stages:
            if x.shape != y.shape or not torch.equal(x, y):
                first = name
                diff = "shape" if x.shape != y.shape else f"{int((x != y).sum())} of {x.numel()} differ"
                break
        ok_all &= first is None
        print(f"frame {t} scene {b}: " + ("all stages identical" if first is None else f"FIRST DIFFERENCE at {first} ({diff})"), flush=True)
print("OK" if ok_all else "MISMATCH")
sys.exit(0 if ok_all else 1)
