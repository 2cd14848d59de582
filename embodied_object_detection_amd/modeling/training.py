"""Training forward + backward of the proposal half of `forward_model` (`custom_rcnn.py:584-679` up to the proposal
losses): image -> backbone with the memory read fused in -> CenterNet head -> target assignment -> `CenterNet.losses` -> gradients of
every parameter upstream (head tower + GroupNorm + output convs + level scales, FPN, map_merge projections, ResNet-50 trunk), all on
the HIP kernels.  The ROI heads' half (proposal matching, cascade / mask losses) is not part of it.

Not the inference hot path: the head here keeps each layer's activations and runs its 5-channel output conv in a 32-channel tile
(the weight-gradient kernel's tile width); the arithmetic per layer is the hot path's.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import ops
from .backward import BackboneBackward


class ProposalTraining:
    def __init__(self, model, sd: Dict[str, torch.Tensor], side_stream: bool = False):
        """`model`: the built `CustomRCNNRecurrent`; `sd`: its state dict (fp32 masters of the parameters the step differentiates).
        `side_stream`: the weight-gradient launches run on a second stream beside the dgrad chain (`ops.ConvBackward`); joined at
        the end of `forward_backward`.  Measured at 640x640 (tools/train_step_bench.py, same call): the whole step 15.6 ms either
        way, the proposal half 14.0 against 10.9 ms -- two events and four `record_stream` calls per layer cost the host more than
        the overlap gives a step whose kernels already fill the chip: off."""
        self.model = model
        self.side = bool(side_stream)
        self.dev = model.device
        self.pg = model.proposal_generator
        c = model.cfg.MODEL.CENTERNET
        self.loss_cfg = dict(alpha=float(c.HM_FOCAL_ALPHA), beta=float(c.HM_FOCAL_BETA), gamma=float(c.LOSS_GAMMA),
                             sigmoid_clamp=float(c.SIGMOID_CLAMP), ignore_high_fp=float(c.IGNORE_HIGH_FP), pos_weight=float(c.POS_WEIGHT),
                             neg_weight=float(c.NEG_WEIGHT), reg_weight=float(c.REG_WEIGHT))
        if str(c.LOC_LOSS_TYPE) != "giou" or not bool(c.NOT_NORM_REG) or bool(c.MORE_POS) or bool(c.NO_REDUCE):
            raise NotImplementedError("proposal losses: LOC_LOSS_TYPE giou, NOT_NORM_REG, no MORE_POS / NO_REDUCE (the recurrent yaml)")
        self.target_cfg = dict(strides=list(c.FPN_STRIDES), sizes_of_interest=[tuple(x) for x in c.SOI],
                               hm_min_overlap=float(c.HM_MIN_OVERLAP), min_radius=float(c.MIN_RADIUS))
        self.bb = BackboneBackward(model.backbone, [sd[f"backbone.map_merge_projection{i}.weight"] for i in (1, 2, 3)], side_stream=self.side)
        h = "proposal_generator.centernet_head"
        w = torch.zeros((32, 256, 3, 3))
        b = torch.zeros((32,))
        w[:5] = torch.cat([sd[f"{h}.agn_hm.weight"], sd[f"{h}.bbox_pred.weight"]], dim=0).float()
        b[:5] = torch.cat([sd[f"{h}.agn_hm.bias"], sd[f"{h}.bbox_pred.bias"]], dim=0).float()
        self.out32 = ops.Conv(w, b, pad=1, device=self.dev, name="agn_hm+bbox_pred")
        self._bw: Dict[int, ops.ConvBackward] = {}
        self._loss = {}
        self.last = None
        self.pyramid_backward = True     # False: the level-shared layers' backward level by level (25 calls + their sums; tests compare)
        # which part of the trunk half's backward anybody reads (`ProposalTrainer` clears them for frozen parameters,
        # MODEL.FREEZE_BACKBONE): the FPN laterals / output convs, and the ResNet blocks + stem below them
        self.backward_fpn = True
        self.backward_blocks = True

    # ---- helpers -------------------------------------------------------------------------------------------------------------------
    def _conv_bwd(self, conv: ops.Conv, xin: torch.Tensor, gout: torch.Tensor, shapes, off):
        """Backward of a level-shared conv over the pyramid row list: per level on that level's grid, dW / db summed over the levels."""
        if id(conv) not in self._bw:
            self._bw[id(conv)] = ops.ConvBackward(conv, side_stream=self.side)
        bwd = self._bw[id(conv)]
        if self.pyramid_backward and not self.side:
            # one weight-gradient launch and one input-gradient launch over all levels' rows (`ConvBackward(..., levels=)`)
            o = bwd(xin, None, gout, levels=(off, shapes))
            return o["dx"], o["dw"], o["db"]
        dw = db = None
        dxs = []
        for l, (h, w) in enumerate(shapes):
            o = bwd(xin[off[l]:off[l + 1]].view(1, h, w, conv.Cin), None, gout[off[l]:off[l + 1]].view(1, h, w, conv.Cout))
            with ops.ConvBackward.on_side(self.dev, self.side):      # the sums over the levels follow the launches they read
                dw = o["dw"] if dw is None else dw.add_(o["dw"])
                db = o["db"] if db is None else db.add_(o["db"])
            dxs.append(o["dx"].reshape(-1, conv.Cin))
        return torch.cat(dxs), dw, db

    # ---- one step ------------------------------------------------------------------------------------------------------------------
    def forward_backward(self, image_u8: torch.Tensor, gt_boxes: torch.Tensor, memory=None, world_size: int = 1, reduce_counts=None,
                         roi_half=None):
        """image_u8 [3,H,W] on the device, gt_boxes [N,4] fp32 on the device, `memory = (memory_f16, proj_indices)` or None ->
        (losses {name: float tensor on the device}, grads {layer or parameter name: tensors}).

        `roi_half(P, head, shapes, off) -> (losses, grads, [dP3, dP4, dP5])`: the ROI heads' half of `forward_model` on the same
        pyramid (`ForwardModelTraining`); its losses / gradients join the result, its pyramid gradients the backbone's backward.

        `reduce_counts(counts int32 [2]) -> counts` sums the positives / regression rows over the ranks (the reference's
        `reduce_sum`, centernet.py:263-265,293); both are then divided by `world_size`."""
        m = self.model
        x4, Hp, Wp = ops.preprocess_image(image_u8, m.pixel_mean, m.pixel_std)
        p345, saved = self.bb.forward_trunk(x4, Hp, Wp, 1)
        losses, grads, g_out = self._frame_on_pyramid(p345, saved, Hp, Wp, gt_boxes, memory, world_size, reduce_counts, roi_half)
        if self.backward_fpn:
            bgrads, _ = self.bb.backward_trunk(saved, g_out, blocks=self.backward_blocks)
            grads.update(bgrads)
        if self.side:
            ops.ConvBackward.join(self.dev)                           # the weight gradients ran on their own stream (ops.ConvBackward)
        return losses, grads

    def forward_backward_batch(self, images: List[torch.Tensor], gt_boxes: List[torch.Tensor], memories: List, world_size: int = 1,
                               reduce_counts=None, roi_halves: Optional[List] = None):
        """`forward_backward` for B frames of one image size: the memory-independent trunk half (ResNet-50, FPN laterals / top-down /
        output convs) runs ONCE for the B images, forward and backward -- one launch per layer over N = B, planned like a single
        image (every frame's pyramid is bitwise the single-frame one), weight gradients summed over the frames by the launch --
        and everything that is per scene (memory fusion, P6 / P7, heads, losses, their backward) frame by frame in between.
        -> ([losses of frame b], grads summed over the frames: what one `losses.backward()` over the summed loss gives)."""
        m, B = self.model, len(images)
        xs = []
        for img in images:
            x4, Hp, Wp = ops.preprocess_image(img, m.pixel_mean, m.pixel_std)
            xs.append(x4)
        p345, saved = self.bb.forward_trunk(torch.cat(xs, dim=0), Hp, Wp, B)
        all_losses, total, g_outs = [], None, []
        for b in range(B):
            losses, grads, g_out = self._frame_on_pyramid([p[b:b + 1] for p in p345], None, Hp, Wp, gt_boxes[b], memories[b], world_size,
                                                          reduce_counts, roi_halves[b] if roi_halves is not None else None)
            all_losses.append(losses)
            g_outs.append(g_out)
            total = grads if total is None else _sum_grads(total, grads)
        if self.backward_fpn:
            bgrads, _ = self.bb.backward_trunk(saved, [torch.cat([g[l] for g in g_outs], dim=0) for l in range(3)], blocks=self.backward_blocks)
            total.update(bgrads)
        if self.side:
            ops.ConvBackward.join(self.dev)
        return all_losses, total

    def _frame_on_pyramid(self, p345, saved, Hp, Wp, gt_boxes, memory, world_size, reduce_counts, roi_half):
        """Everything of a frame behind the trunk half: memory fusion + P6 / P7 (`forward_tail`), CenterNet head, targets, proposal
        losses, the ROI heads' half, and their backward down to the gradient of p3..p5 -> (losses, grads, [g3, g4, g5])."""
        pg, dev = self.pg, self.dev
        P, tail = self.bb.forward_tail(p345, Hp, Wp, memory)
        if saved is not None:                                        # single-frame step: the whole record under one roof (tests)
            saved["tail"], saved["pooled"] = tail, tail["pooled"]
            saved["fpn"].update(P=tail["P"], p6=tail["p6"])
        shapes = [(p.shape[1], p.shape[2]) for p in P]
        off = [0]
        for (h, w) in shapes:
            off.append(off[-1] + h * w)
        feats = torch.cat([p.reshape(-1, 256) for p in P])
        # CenterNet head (centernet_head.py:141-161), every layer's input / pre-norm / output kept
        keep, x = [], feats
        for (conv, gamma, beta) in pg.tower:
            c = conv(x, 1, 0, 0, levels=(off, shapes))
            st = ops.groupnorm_workspace(off, dev)
            y = ops.groupnorm_relu(c, gamma, beta, off, 256, st)
            keep.append((x, c, st, y))
            x = y
        head = self.out32(x, 1, 0, 0, levels=(off, shapes))                                   # [P, 32]: cols 0..4 are the head's
        self.last = dict(keep=keep, head=head, shapes=shapes, off=off, saved=saved)                        # for inspection (tests)
        # targets (centernet.py:342-479) and losses (:241-318)
        heat, reg_t, pos, counts = ops.centernet_targets(gt_boxes, shapes, **self.target_cfg)
        # the reference reads the counts back (`.item()`, centernet.py:264,293); here the losses' kernel reads them on the device
        total = reduce_counts(counts) if reduce_counts is not None else counts
        key = tuple(shapes)
        if key not in self._loss:
            self._loss[key] = ops.CenterNetLoss(off, pg.scales, dev, head_stride=32, **self.loss_cfg)
        losses_t, d_head = self._loss[key](head, heat, reg_t, pos, counts_local=counts, counts_total=total, world_size=world_size)
        losses_t = losses_t.clone()              # the loss object's own buffer: the next frame of a batch writes to it again
        losses = {"loss_centernet_loc": losses_t[0], "loss_centernet_agn_pos": losses_t[1], "loss_centernet_agn_neg": losses_t[2]}
        # ---- backward
        grads: Dict[str, tuple] = {}
        roi_dP = None
        if roi_half is not None:
            roi_losses, roi_grads, roi_dP = roi_half(P, head, shapes, off)
            losses.update(roi_losses)
            grads.update(roi_grads)
        # the levels' Scale parameters (centernet_head.py:153-155): d scale_l = sum over the level of d reg * raw = d raw * raw / scale_l
        prod = (d_head[:, 1:5] * head[:, 1:5]).sum(dim=1)
        grads["scales"] = torch.stack([prod[off[l]:off[l + 1]].sum() / pg.scales[l] for l in range(len(shapes))])
        gx, dw, db = self._conv_bwd(self.out32, x, d_head, shapes, off)
        grads["agn_hm"] = (dw[0:1], db[0:1])                          # views: no device work
        grads["bbox_pred"] = (dw[1:5], db[1:5])
        for i in reversed(range(len(pg.tower))):
            conv, gamma, beta = pg.tower[i]
            xin, c, st, y = keep[i]
            dc, dgamma, dbeta = ops.groupnorm_relu_backward(c, y, gx, gamma, off, 256, st)
            gx, dw, db = self._conv_bwd(conv, xin, dc, shapes, off)
            grads[conv.name] = (dw, db)
            grads[conv.name + ".norm"] = (dgamma, dbeta)
        dP = [gx[off[l]:off[l + 1]].view(1, shapes[l][0], shapes[l][1], 256) for l in range(len(shapes))]
        if roi_dP is not None:
            for l, d in enumerate(roi_dP):
                dP[l] = (dP[l] + d.view(dP[l].shape)).contiguous()
        tgrads, g_out = self.bb.backward_tail(tail, dP)
        grads.update(tgrads)
        return losses, grads, g_out


def _sum_grads(total: dict, grads: dict) -> dict:
    """total += grads, entry by entry (values: a tensor or a tuple of tensors; an entry only one side has is kept): the frames'
    gradients of one training iteration, added by one multi-tensor launch."""
    a_list, b_list = [], []
    for k, v in grads.items():
        if k not in total:
            total[k] = v
            continue
        t = total[k]
        if torch.is_tensor(v):
            a_list.append(t)
            b_list.append(v)
        else:
            fixed = []
            for ti, vi in zip(t, v):
                if ti is None or vi is None:
                    fixed.append(vi if ti is None else ti)
                    continue
                a_list.append(ti)            # (every gradient tensor is the fresh output of its launch, or a view of one)
                b_list.append(vi)
                fixed.append(ti)
            total[k] = tuple(fixed)
    if a_list:
        torch._foreach_add_(a_list, b_list)
    return total


class ProposalTrainer:
    """`ProposalTraining` closed into optimizer steps: the reference's AdamW set-up (`build_custom_optimizer`, custom_solver.py:19-79, as
    `solver.param_groups_from_cfg` restates it: BACKBONE_MULTIPLIER, `map_merge` at CUSTOM_MULTIPLIER, clip by value) over every
    parameter upstream of the proposal losses, updated IN the layers the inference path runs: FPN / tower / P6 / P7 convs step their
    packed weights in place; the trunk's convs step a raw master and are re-folded with their FrozenBatchNorm (whose weight / bias are
    buffers, not parameters: timm.py:277-299); the map_merge projections are re-prepared; the level scales go back to the decoder."""

    def __init__(self, model, sd: Dict[str, torch.Tensor], roi_heads: bool = False):
        """`roi_heads=True` (`Trainer`): `forward_model` with both halves -- the 30 tensors of the cascade's box heads / predictors
        join the optimizer and the ROI heads' losses and pyramid gradients the step."""
        from .. import solver
        self.model, self.dev = model, model.device
        self.step_fn = ProposalTraining(model, sd)
        self.fm = ForwardModelTraining(model, sd, prop=self.step_fn) if roi_heads else None
        cfg = model.cfg
        dev = self.dev
        bbm, pg = model.backbone, model.proposal_generator
        self.entries = []          # (reference parameter name, tensor stepped by AdamW, gradient getter)
        self.after = []            # callables run after every optimizer step
        self.folds = {}            # parameter name -> (layer weights, per-row FrozenBatchNorm scale): re-folded by the optimizer's launch
        self.raw_getters = {}      # parameter name -> gradient of the FOLDED weights (what the optimizer's launch takes for those)
        base = "backbone.bottom_up.base"

        def add(name, tensor, getter):
            self.entries.append((name, tensor, getter))

        def trunk_conv(conv, wname, bnp, cin_pad=None):
            # raw master in the packed layout; gradient of the raw weight = gradient of the folded one x gamma / sqrt(var + eps)
            master, _ = ops.pack_conv_weight(sd[wname].float(), cin_pad)
            K = conv.KH * conv.KW * conv.Cin
            master = master[:, :K].contiguous().to(dev)
            scale = (sd[f"{bnp}.weight"].float() / torch.sqrt(sd[f"{bnp}.running_var"].float() + 1e-5)).to(dev).view(-1, 1)
            add(wname, master, lambda g, n=conv.name, s=scale: (g[n][0] * s).contiguous())
            # the optimizer's launch writes the re-folded weights (master x scale) straight into the layer (`AdamW` group key "fold")
            # and takes the gradient of the folded weights as it comes out of the backward pass (key "grad_of_folded": x scale inside)
            self.folds[wname] = (conv.w, scale.view(-1).contiguous())
            self.raw_getters[wname] = lambda g, n=conv.name: g[n][0]

        trunk_conv(bbm.bottom_up.stem, f"{base}.conv1.weight", f"{base}.bn1", cin_pad=4)
        for (li, c1, c2, c3, ds) in bbm.bottom_up.blocks:
            for conv in (c1, c2, c3):
                p = conv.name                                    # '<base>.layerL.B.convI'
                trunk_conv(conv, p + ".weight", p.rsplit(".conv", 1)[0] + ".bn" + p[-1])
            if ds is not None:
                trunk_conv(ds, ds.name + ".0.weight", ds.name + ".1")

        def plain_conv(conv, prefix, rows=None):
            assert conv.Kpad == conv.KH * conv.KW * conv.Cin
            w = conv.w if rows is None else conv.w[rows[0]:rows[1]]
            b = conv.bias if rows is None else conv.bias[rows[0]:rows[1]]
            return w, b

        for l in (3, 4, 5):
            for kind, conv in (("lateral", bbm.lateral[l]), ("output", bbm.output[l])):
                w, b = plain_conv(conv, None)
                add(f"backbone.fpn_{kind}{l}.weight", w, lambda g, n=conv.name: g[n][0])
                add(f"backbone.fpn_{kind}{l}.bias", b, lambda g, n=conv.name: g[n][1])
        for name, conv in (("p6", bbm.p6), ("p7", bbm.p7)):
            w, b = plain_conv(conv, None)
            add(f"backbone.top_block.{name}.weight", w, lambda g, n=conv.name: g[n][0])
            add(f"backbone.top_block.{name}.bias", b, lambda g, n=conv.name: g[n][1])
        self.merge_w = [sd[f"backbone.map_merge_projection{i}.weight"].float().reshape(256, 512).contiguous().to(dev) for i in (1, 2, 3)]
        self.merge_b = [sd[f"backbone.map_merge_projection{i}.bias"].float().contiguous().to(dev) for i in (1, 2, 3)]
        for i in range(3):
            # a frame without a memory (MEMORY_TYPE '': the non-recurrent detector) leaves the projections without a gradient:
            # None, which the optimizer skips as torch skips a parameter whose .grad is None
            add(f"backbone.map_merge_projection{i + 1}.weight", self.merge_w[i],
                lambda g, i=i: g.get(f"map_merge_projection{i + 1}", (None, None))[0])
            add(f"backbone.map_merge_projection{i + 1}.bias", self.merge_b[i],
                lambda g, i=i: g.get(f"map_merge_projection{i + 1}", (None, None))[1])

        self.step_fn.bb.merge_weights = self.merge_w
        self.after.append(lambda: bbm.merge.refresh(self.merge_w, self.merge_b))

        def refresh_merge_backward():                             # the W^T convs of the pooled memory's gradient (read by tests only)
            if self.step_fn.bb._merge_bw is not None:
                self.step_fn.bb._merge_bw.refresh(self.merge_w)
        self.after.append(refresh_merge_backward)
        h = "proposal_generator.centernet_head"
        for i, (conv, gamma, beta) in enumerate(pg.tower):
            w, b = plain_conv(conv, None)
            add(f"{h}.bbox_tower.{3 * i}.weight", w, lambda g, n=conv.name: g[n][0])
            add(f"{h}.bbox_tower.{3 * i}.bias", b, lambda g, n=conv.name: g[n][1])
            add(f"{h}.bbox_tower.{3 * i + 1}.weight", gamma, lambda g, n=conv.name: g[n + ".norm"][0])
            add(f"{h}.bbox_tower.{3 * i + 1}.bias", beta, lambda g, n=conv.name: g[n + ".norm"][1])
        out32 = self.step_fn.out32
        for name, rows in (("agn_hm", (0, 1)), ("bbox_pred", (1, 5))):
            w, b = plain_conv(out32, None, rows)
            add(f"{h}.{name}.weight", w, lambda g, n=name: g[n][0].contiguous())
            add(f"{h}.{name}.bias", b, lambda g, n=name: g[n][1].contiguous())

        def sync_out_conv():                                      # the inference path's 5-channel layer is its own object
            pg.out_conv.w.copy_(out32.w[:5])
            pg.out_conv.bias.copy_(out32.bias[:5])
            pg.out_conv.w_split = None                               # bf16x3 pieces of the old weights, if that arithmetic was in use
        self.after.append(sync_out_conv)
        self.scales = torch.tensor(pg.scales, dtype=torch.float32, device=dev)
        add(f"{h}.scales", self.scales, lambda g: g["scales"])    # five scalar parameters `scales.{l}.scale`, stepped as one tensor

        def sync_scales():
            pg.scales = [float(v) for v in self.scales.cpu().tolist()]
            pg._plans.clear()                                    # the decoders carry the scales in their descriptors
            descs = [loss.desc for loss in self.step_fn._loss.values()]
            if self.fm is not None:
                descs += [dec.desc for dec in self.fm._dec.values()]
            for d in descs:
                for l, v in enumerate(pg.scales):
                    d.level_scale[l] = v
        self.after.append(sync_scales)

        if self.fm is not None:
            det, rh = self.fm.det, model.roi_heads
            for k, st in enumerate(rh.stages):
                for conv, ref in ((st["fc1"], f"roi_heads.box_head.{k}.fc1"), (st["fc2"], f"roi_heads.box_head.{k}.fc2"),
                                  (st["cls"], f"roi_heads.box_predictor.{k}.cls_score.linear"),
                                  (st["bb0"], f"roi_heads.box_predictor.{k}.bbox_pred.0")):
                    w, b = plain_conv(conv, None)                        # fc1's columns are in the pooled rows' (7,7,C) order
                    add(f"{ref}.weight", w, lambda g, n=conv.name: g[n][0])
                    add(f"{ref}.bias", b, lambda g, n=conv.name: g[n][1])
                b32, ref = det.bb2_32[k], f"roi_heads.box_predictor.{k}.bbox_pred.2"
                add(f"{ref}.weight", b32.w[:4], lambda g, n=st["bb2"].name: g[n][0])
                add(f"{ref}.bias", b32.bias[:4], lambda g, n=st["bb2"].name: g[n][1])

            def sync_roi_heads():                                     # the inference path's own objects for the same parameters
                for k, st in enumerate(rh.stages):
                    st["bb2"].w.copy_(det.bb2_32[k].w[:4])
                    st["bb2"].bias.copy_(det.bb2_32[k].bias[:4])
                    st["bb2"].w_split = None
                    st["cls_bb0"].w[:512].copy_(st["cls"].w)
                    st["cls_bb0"].w[512:].copy_(st["bb0"].w)
                    st["cls_bb0"].bias[:512].copy_(st["cls"].bias)
                    st["cls_bb0"].bias[512:].copy_(st["bb0"].bias)
                    st["cls_bb0"].w_split = None
            self.after.append(sync_roi_heads)

        def stale_caches():
            extra = list(self.fm.det._bw.values()) if self.fm is not None else []
            bws = list(self.step_fn._bw.values()) + list(self.step_fn.bb._bw.values()) + extra
            ops.ConvBackward.refresh_all(bws)                    # rotated weights of the dgrad convs: all layers in 4 launches
            for bw in bws:
                bw.conv.w_split = None                           # bf16x3 pieces, if that arithmetic was in use
        self.after.append(stale_caches)
        s = cfg.SOLVER
        if bool(cfg.FP16):
            # custom_rcnn.py:607-618 runs the backbone under autocast on a half image and train_mp3d.py:577-578,628-631 scales the loss
            # with a GradScaler; this step computes in fp32 throughout.  Silently training in another arithmetic is not an option.
            raise NotImplementedError("FP16: True (autocast backbone + GradScaler, custom_rcnn.py:607-618, train_mp3d.py:577-631) is not "
                                      "implemented: the training step computes in fp32.  Pass `FP16 False` to train in fp32.")
        if str(s.OPTIMIZER) != "ADAMW":
            raise NotImplementedError("the device optimizer is AdamW (SOLVER.OPTIMIZER ADAMW, Base-...recurrent.yaml)")
        frozen = []
        if bool(cfg.MODEL.FREEZE_BACKBONE):
            # train_mp3d.py:704-710: a parameter stays trainable iff one of UNFROZEN_LAYERS is a substring of its name
            keys = list(cfg.MODEL.UNFROZEN_LAYERS)
            frozen = [n for n, _, _ in self.entries if not any(k in n for k in keys)]
        self.groups = solver.param_groups_from_cfg(cfg, [(n, t) for n, t, _ in self.entries])
        self.groups = [g for g in self.groups if g["name"] not in frozen]
        for g in self.groups:
            if g["name"] in self.folds:
                g["fold"] = self.folds[g["name"]]
                g["grad_of_folded"] = True
        # frozen parameters have no reader for their gradients: the trunk half's backward stops where the last trainable layer is
        # (with the yaml's UNFROZEN_LAYERS ['roi', 'map_merge', 'proposal_generator'] it is skipped altogether)
        live = [g["name"] for g in self.groups]
        self.step_fn.backward_blocks = any("bottom_up" in n for n in live)
        self.step_fn.backward_fpn = self.step_fn.backward_blocks or any("backbone.fpn_" in n for n in live)
        self.getters = {n: f for n, _, f in self.entries}        # gradient of every stepped tensor (of the raw master for a trunk conv)
        self.step_getters = {**self.getters, **self.raw_getters}   # what `opt.step` is handed
        clip = s.CLIP_GRADIENTS
        if bool(clip.ENABLED) and str(clip.CLIP_TYPE) != "value":
            raise NotImplementedError("gradient clipping: CLIP_TYPE value (detectron2's default)")
        self.opt = ops.AdamW(self.groups, weight_decay=float(s.WEIGHT_DECAY), clip_value=float(clip.CLIP_VALUE) if bool(clip.ENABLED) else 0.0)
        self.iteration = 0

    def state_dict(self, base_sd: Dict[str, torch.Tensor]):
        """The stepped parameters in the reference's names and layouts, on top of `base_sd` (the state dict the model was built from):
        what `DetectionCheckpointer` would save (`checkpoint.save_checkpoint` writes it; `load_checkpoint` + `build_model` read it)."""
        from .. import checkpoint
        return checkpoint.export_state_dict(self.entries, base_sd, self.model.roi_heads.num_classes)

    def optimizer_state(self) -> Dict:
        """The optimizer's moments and step counts by reference parameter name (the 'optimizer' entry of a checkpoint)."""
        return self.opt.state_dict()

    def load_optimizer_state(self, sd: Dict) -> None:
        """`--resume` (train_mp3d.py:524): continue with the stored moments / step counts instead of a cold AdamW."""
        self.opt.load_state_dict(sd)
        self.iteration = max((int(e["step"]) for e in sd.values()), default=0)

    def step(self, image_u8: torch.Tensor, gt_boxes: torch.Tensor, memory=None, lr_factor: float = 1.0, gt_classes=None, proposals=None,
             keys=None, generator=None):
        """One training iteration on one frame -> the losses (device scalars, of the weights BEFORE the update): the three proposal
        losses, with `roi_heads=True` also the cascade's six and loss_mask (`gt_classes` int32 [N] needed)."""
        if self.fm is not None:
            losses, grads = self.fm.forward_backward(image_u8, gt_boxes, gt_classes, memory=memory, proposals=proposals, keys=keys,
                                                     generator=generator)
        else:
            losses, grads = self.step_fn.forward_backward(image_u8, gt_boxes, memory=memory)
        self.opt.step([self.step_getters[g["name"]](grads) for g in self.groups], lr_factor=lr_factor)
        for f in self.after:
            f()
        self.iteration += 1
        return losses


class DetectorTraining:
    """The ROI heads' half of `forward_model`, forward (custom_rcnn.py:642-650 -> `DeticCascadeROIHeads.forward` in training mode,
    detic_roi_heads.py:226-249 with ann_type 'box'): `label_and_sample_proposals` (ground truth appended, IoU matching, 512 rows at
    1/4 foreground), then per cascade stage ROIAlign(7) -> box head -> predictor -> `DeticFastRCNNOutputLayers.losses`, the next
    stage's proposals = the refined boxes, clipped, empty ones dropped, re-matched at the stage's IoU (`_match_and_label_boxes`,
    :115).  The MP3D loader carries no `gt_masks`, so the mask branch is `_get_empty_mask_loss` (:246-249): loss_mask = 0.

    Runs on the layers of the inference path (`roi_heads.stages`): matching / sampling / logits / losses are the kernels of
    `csrc/train_losses.hip`, the rest the hot path's ROIAlign and GEMMs at N = the sampled row count.  Also returns each stage's
    loss gradients w.r.t. the predictor outputs (d scores, d deltas) and keeps the activations a backward pass needs.  The
    proposals are an input (`ForwardModelTraining.train_proposals` decodes them with PRE / POST_NMS_TOPK_TRAIN 4000 / 2000)."""

    def __init__(self, model, side_stream: bool = False):
        cfg = model.cfg
        self.model, self.dev, self.rh = model, model.device, model.roi_heads
        self.side = bool(side_stream)                  # weight gradients on a second stream (`ops.ConvBackward`), joined in `backward`
        rhc, rb = cfg.MODEL.ROI_HEADS, cfg.MODEL.ROI_BOX_HEAD
        self.ious = tuple(float(v) for v in cfg.MODEL.ROI_BOX_CASCADE_HEAD.IOUS)
        if len(self.ious) != self.rh.num_stages or self.ious[0] != float(rhc.IOU_THRESHOLDS[0]):
            raise ValueError("ROI_BOX_CASCADE_HEAD.IOUS: one IoU per stage, the first equal to ROI_HEADS.IOU_THRESHOLDS[0] "
                             "(detectron2 CascadeROIHeads.from_config)")
        if bool(rb.USE_FED_LOSS) or bool(rb.IGNORE_ZERO_CATS):
            raise NotImplementedError("federated loss / zero-frequency categories need the LVIS frequency file (not on the MP3D path)")
        if str(rb.BBOX_REG_LOSS_TYPE) != "smooth_l1":
            raise NotImplementedError("ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE: smooth_l1 (the recurrent yaml)")
        self.batch, self.frac = int(rhc.BATCH_SIZE_PER_IMAGE), float(rhc.POSITIVE_FRACTION)
        self.append_gt = bool(rhc.PROPOSAL_APPEND_GT)
        self.beta, self.box_w = float(rb.SMOOTH_L1_BETA), float(rb.BBOX_REG_LOSS_WEIGHT)
        self.C = self.rh.num_classes
        self.last = None
        # optimistic assumptions about data-dependent sizes instead of host round trips in the middle of the step (see
        # `label_and_sample`, `losses`): device-side booleans, True = the assumption did NOT hold; `ForwardModelTraining` reads them
        # once, after the whole step has been enqueued, and repeats the frame on the exact path if one is set
        self.speculate = False
        self.checks: List[torch.Tensor] = []
        # bbox_pred.2 (1024 -> 4) in a 32-row tile, the weight-gradient kernel's tile width (as the proposal head's 5-channel output conv)
        self.bb2_32 = []
        for st in self.rh.stages:
            c = st["bb2"]
            w, b = torch.zeros((32, c.Cin)), torch.zeros((32,))
            w[:c.Cout], b[:c.Cout] = c.w[:, :c.Cin].cpu(), c.bias.cpu()
            self.bb2_32.append(ops.Conv(w[:, :, None, None], b, device=self.dev, name=c.name))
        self._bw: Dict[int, ops.ConvBackward] = {}

    def label_and_sample(self, prop_boxes: torch.Tensor, gt_boxes: torch.Tensor, gt_classes: torch.Tensor, keys: Optional[torch.Tensor] = None,
                         generator: Optional[torch.Generator] = None, prop_count: Optional[torch.Tensor] = None):
        """detic_roi_heads.py:232 -> (boxes [B,4], classes int32 [B], matched gt boxes [B,4], sampled rows int64 [B]).

        `prop_count` (int32 [1] on the device): `prop_boxes` is the decoder's capacity-sized list; the ground truth is appended
        behind its live rows and the rows beyond are ignored, all on the device (`eod_match_label_proposals`).  With `speculate`
        the number of sampled rows is not read back either: BATCH_SIZE_PER_IMAGE rows are taken (what the sampling yields whenever
        there are that many candidates) and the assumption is filed in `self.checks` for the caller to verify after the step."""
        if prop_count is not None:
            boxes, cls, gtb = ops.match_label_proposals(prop_boxes, prop_count, gt_boxes, gt_classes, self.ious[0], self.C, self.append_gt)
        else:
            boxes = torch.cat([prop_boxes, gt_boxes]).contiguous() if self.append_gt else prop_boxes.contiguous()
            _, _, cls, gtb = ops.match_label(boxes, gt_boxes, gt_classes, self.ious[0], self.C)
        if keys is None:
            keys = torch.rand((boxes.shape[0],), device=self.dev, generator=generator)
        else:
            keys = keys[:boxes.shape[0]].contiguous()              # one key per row; a caller that cannot know R may pass more
        idx, counts = ops.sample_proposals(cls, keys, self.C, self.batch, self.frac)
        if self.speculate:
            n = self.batch
            self.checks.append(counts[1:2] != n)
        else:
            n = int(counts.cpu()[1])                               # the reference's nonzero() / randperm sizes
        rows = idx[:n].long()
        self.last_rows = rows
        return boxes.index_select(0, rows), cls.index_select(0, rows), gtb.index_select(0, rows), rows

    def run_stage(self, P, boxes: torch.Tensor, k: int):
        """`_run_stage` (:328-349) on B rows -> dict of the stage's activations; logits [B, C+1], deltas [B,4]."""
        st, B = self.rh.stages[k], int(boxes.shape[0])
        h3, w3 = int(P[0].shape[1]), int(P[0].shape[2])
        pool = ops.roi_align(P[0], P[1], P[2], h3, w3, 256, boxes, None, B, 7)
        h1 = st["fc1"](pool, B, 1, 1, relu=True)
        h2 = st["fc2"](h1, B, 1, 1, relu=True)
        feat = st["cls"](h2, B, 1, 1)
        hb = st["bb0"](h2, B, 1, 1, relu=True)
        deltas = self.bb2_32[k](hb, B, 1, 1).view(B, 32)[:, :4].contiguous()
        featn = torch.empty((B, 512), dtype=torch.float32, device=self.dev)
        logits = ops.zs_logits(feat, st["zs"], self.rh.norm_temp, featn_out=featn)
        return dict(boxes=boxes, pool=pool, h1=h1, h2=h2, feat=feat, featn=featn, hb=hb, deltas=deltas, logits=logits)

    def losses(self, P: Sequence[torch.Tensor], prop_boxes: torch.Tensor, gt_boxes: torch.Tensor, gt_classes: torch.Tensor,
               image_hw: Tuple[int, int], keys: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
               prop_count: Optional[torch.Tensor] = None):
        """P: the pyramid's P3..P5 as [1,h,w,256] device tensors; prop_boxes [R,4]; gt_boxes [G,4] fp32, gt_classes int32 [G] ->
        {loss_cls_stage{k}, loss_box_reg_stage{k}, loss_mask} (device scalars) and the per-stage records (`self.last`)."""
        H, W = image_hw
        gt_boxes, gt_classes = gt_boxes.contiguous(), gt_classes.to(torch.int32).contiguous()
        boxes, cls, gtb, _ = self.label_and_sample(prop_boxes, gt_boxes, gt_classes, keys, generator, prop_count=prop_count)
        out: Dict[str, torch.Tensor] = {}
        stages = []
        for k in range(self.rh.num_stages):
            if k > 0:
                prev = stages[-1]
                B0 = int(prev["boxes"].shape[0])
                nxt = torch.empty((B0, 4), dtype=torch.float32, device=self.dev)
                ops.apply_deltas(prev["deltas"], 4, prev["boxes"], nxt, None, B0, self.rh.cascade_weights[k - 1], True, float(W), float(H))
                keep = (nxt[:, 2] - nxt[:, 0] > 0) & (nxt[:, 3] - nxt[:, 1] > 0)          # Boxes.nonempty (:317-319)
                if self.speculate:
                    boxes = nxt                                                          # refined boxes are practically never empty
                    self.checks.append((~keep).any().reshape(1))
                else:
                    boxes = nxt if bool(keep.all()) else nxt[keep].contiguous()
                if boxes.shape[0] == 0:
                    raise RuntimeError(f"cascade stage {k}: every refined box is empty")
                _, _, cls, gtb = ops.match_label(boxes, gt_boxes, gt_classes, self.ious[k], self.C)
            rec = self.run_stage(P, boxes, k)
            l, ds, dd = ops.fast_rcnn_loss(rec["logits"], rec["deltas"], boxes, gtb, cls, self.C, self.rh.cascade_weights[k], None, self.beta)
            rec.update(classes=cls, gt_boxes=gtb, d_logits=ds, d_deltas=dd)
            stages.append(rec)
            out[f"loss_cls_stage{k}"] = l[0]
            out[f"loss_box_reg_stage{k}"] = l[1] * self.box_w
        if bool(self.model.cfg.MODEL.MASK_ON):
            out["loss_mask"] = torch.zeros((), dtype=torch.float32, device=self.dev)
        self.last = stages
        return out

    def _conv_bw(self, conv: ops.Conv) -> ops.ConvBackward:
        if id(conv) not in self._bw:
            self._bw[id(conv)] = ops.ConvBackward(conv, side_stream=self.side)
        return self._bw[id(conv)]

    def backward(self, P: Sequence[torch.Tensor], join: bool = True):
        """`join=False` (inside `ForwardModelTraining`): the caller joins the weight gradients' stream at the end of the whole backward.

        Gradients of the sum of the stage losses of the last `losses()` call -> ({layer name: (dW packed [Cout, K], db)} for the
        five linear layers of every stage, [dP3, dP4, dP5] as [h,w,256]).  Per stage: d logits -> normalize / class matrix
        (`eod_zs_logits_backward`) -> cls_score.linear; d deltas -> bbox_pred.2 -> ReLU -> bbox_pred.0; both into fc2 -> fc1 -> the
        pooled features x 1 / num_stages (`_ScaleGradient`, detic_roi_heads.py:334) -> ROIAlign backward, added over the stages.  The
        stages are independent: the next stage's proposals are detached (`_create_proposals_from_boxes`, :310)."""
        rh = self.rh
        h3, w3 = int(P[0].shape[1]), int(P[0].shape[2])
        dP = [torch.zeros(tuple(p.shape[1:]), dtype=torch.float32, device=self.dev) for p in P[:3]]
        grads: Dict[str, tuple] = {}
        for k, rec in enumerate(self.last):
            st, B = rh.stages[k], int(rec["boxes"].shape[0])
            d_feat = ops.zs_logits_backward(rec["feat"], st["zs"], rec["d_logits"], rh.norm_temp)
            o = self._conv_bw(st["cls"])(rec["h2"], None, d_feat.view(B, 1, 1, 512))
            grads[st["cls"].name] = (o["dw"], o["db"])
            d_h2 = o["dx"]
            g32 = torch.zeros((B, 1, 1, 32), dtype=torch.float32, device=self.dev)
            g32[:, 0, 0, :4] = rec["d_deltas"] * self.box_w
            # every ReLU's backward (and the sum of h2's two gradients) rides on the epilogue of the input-gradient launch that
            # produces the gradient (`dx_gate` / `dx_res` of ops.ConvBackward)
            o = self._conv_bw(self.bb2_32[k])(rec["hb"], None, g32, dx_gate=rec["hb"])
            with ops.ConvBackward.on_side(self.dev, self.side):
                grads[st["bb2"].name] = (o["dw"][:4].contiguous(), o["db"][:4].contiguous())
            o = self._conv_bw(st["bb0"])(rec["h2"], None, o["dx"], dx_res=d_h2, dx_gate=rec["h2"])
            grads[st["bb0"].name] = (o["dw"], o["db"])
            o = self._conv_bw(st["fc2"])(rec["h1"], None, o["dx"], dx_gate=rec["h1"])
            grads[st["fc2"].name] = (o["dw"], o["db"])
            o = self._conv_bw(st["fc1"])(rec["pool"].view(B, 1, 1, -1), None, o["dx"])
            grads[st["fc1"].name] = (o["dw"], o["db"])
            d_pool = (o["dx"].view(B, 7, 7, 256) * (1.0 / rh.num_stages)).contiguous()
            ops.roi_align_backward(dP[0], dP[1], dP[2], h3, w3, 256, rec["boxes"], None, B, 7, d_pool)
        if join and self.side:
            ops.ConvBackward.join(self.dev)
        return grads, dP


class ForwardModelTraining:
    """`CustomRCNNRecurrent.forward_model` (custom_rcnn.py:584-679) of one frame, forward + backward: both halves on one pyramid --
    `ProposalTraining` (backbone with the memory read fused in, CenterNet head, targets, proposal losses) and `DetectorTraining` (the
    cascade's losses on proposals decoded from the same head outputs, as `CenterNet.forward` does in training, centernet.py:214-219);
    the ROI heads' pyramid gradients join the proposal head's before the backbone's backward.  Returns the reference's loss dict
    (:665-673: detector losses + proposal losses) and the gradient of every parameter.

    The training-mode proposal lists have the yaml's sizes (PRE / POST_NMS_TOPK_TRAIN 4000 / 2000, NMS_TH_TRAIN 0.9,
    Base-C2_L_R5021k_640b64_4x_recurrent.yaml:45-49): `eod_centernet_proposals` takes its wide path (rank merge of the per-level
    lists, suppression bit matrix over the chip, one scanning workgroup; csrc/select.hip)."""

    def __init__(self, model, sd: Dict[str, torch.Tensor], prop: Optional[ProposalTraining] = None):
        self.model, self.dev = model, model.device
        self.prop = prop if prop is not None else ProposalTraining(model, sd)
        self.det = DetectorTraining(model)
        c = model.cfg.MODEL.CENTERNET
        self.pre, self.post = int(c.PRE_NMS_TOPK_TRAIN), int(c.POST_NMS_TOPK_TRAIN)
        self.nms_train, self.score_thresh = float(c.NMS_TH_TRAIN), float(c.INFERENCE_TH)
        self._dec: Dict[tuple, ops.ProposalDecoder] = {}
        self._last_props = None
        self.speculate = True            # see forward_backward
        self._exact_sizes = set()
        self.repeated_frames = 0

    def decoder(self, shapes, head_stride: int) -> ops.ProposalDecoder:
        """One decoder per pyramid shape; the level scales are trained parameters and are patched into its descriptor after every
        optimizer step (`ProposalTrainer.sync_scales`)."""
        pg = self.prop.pg
        key = (tuple(shapes), head_stride)
        if key not in self._dec:
            cap = (self.post + 48 + 31) // 32 * 32                      # room for the '>= kth' ties (centernet.py:739)
            self._dec[key] = ops.ProposalDecoder(shapes, pg.strides, pg.scales, self.score_thresh, self.pre, self.post, self.nms_train, cap,
                                                 self.dev, head_stride=head_stride)
        return self._dec[key]

    def train_proposals(self, head: torch.Tensor, shapes):
        """`predict_instances` with the training thresholds on the head's raw rows [P, 32] -> (proposal boxes [cap,4], count int32 [1]),
        both on the device (detached; the decoder's own buffers: valid until its next call)."""
        boxes, _, count = self.decoder(shapes, int(head.shape[1]))(head)
        return boxes, count

    @property
    def last_proposals(self) -> Optional[torch.Tensor]:
        """The proposal list of the last `forward_backward` as [R,4] (reads the count back: for tests and reports)."""
        if self._last_props is None:
            return None
        boxes, count = self._last_props
        return boxes if count is None else boxes[:int(count.cpu()[0])]

    def forward_backward(self, image_u8: torch.Tensor, gt_boxes: torch.Tensor, gt_classes: torch.Tensor, memory=None, proposals=None,
                         keys=None, generator=None, world_size: int = 1, reduce_counts=None):
        """-> (losses: the nine training losses + loss_mask, grads: every layer's (dW, db) / parameter gradient).  `proposals` [R,4]
        replaces the decoded list (tests); `keys` / `generator`: the sampling's random keys."""
        H, W = int(image_u8.shape[1]), int(image_u8.shape[2])
        self._last_props = None
        det = self.det
        # No host round trip inside the step: the proposal count, the number of sampled rows and "no refined box is empty" stay on
        # the device; the last two are ASSUMED (512 rows; none empty) and verified by ONE read-back after everything has been
        # enqueued.  A frame on which an assumption fails is repeated on the exact path, and the optimism is switched off for its
        # image size (small images yield fewer than BATCH_SIZE_PER_IMAGE candidates every time).
        det.speculate = self.speculate and (H, W) not in self._exact_sizes
        det.checks = []

        def roi_half(P, head, shapes, off):
            if proposals is not None:
                props, count = proposals, None
            else:
                props, count = self.train_proposals(head, shapes)
            self._last_props = (props.clone(), None if count is None else count.clone())
            losses = det.losses(P[:3], props, gt_boxes, gt_classes, (H, W), keys=keys, generator=generator, prop_count=count)
            grads, dP = det.backward(P[:3], join=False)
            return losses, grads, dP
        out = self.prop.forward_backward(image_u8, gt_boxes, memory=memory, world_size=world_size, reduce_counts=reduce_counts,
                                         roi_half=roi_half)
        if det.speculate and det.checks and bool(torch.cat(det.checks).any().cpu()):
            self._exact_sizes.add((H, W))
            self.repeated_frames += 1
            det.speculate, det.checks = False, []
            out = self.prop.forward_backward(image_u8, gt_boxes, memory=memory, world_size=world_size, reduce_counts=reduce_counts,
                                             roi_half=roi_half)
        return out

    def forward_backward_batch(self, images: List[torch.Tensor], gt_boxes: List[torch.Tensor], gt_classes: List[torch.Tensor],
                               memories: List, generator: Optional[torch.Generator] = None, world_size: int = 1, reduce_counts=None):
        """`forward_backward` for B frames of one image size with ONE pass of the memory-independent trunk half for all of them
        (`ProposalTraining.forward_backward_batch`) -> ([loss dict of frame b], gradients summed over the frames).  The optimistic
        row counts are verified once for the whole batch; a batch in which a frame fails them is repeated on the exact path."""
        H, W = int(images[0].shape[1]), int(images[0].shape[2])
        det = self.det
        det.speculate = self.speculate and (H, W) not in self._exact_sizes
        det.checks = []

        def half(b):
            def roi_half(P, head, shapes, off):
                props, count = self.train_proposals(head, shapes)
                self._last_props = (props.clone(), None if count is None else count.clone())
                losses = det.losses(P[:3], props, gt_boxes[b], gt_classes[b], (H, W), generator=generator, prop_count=count)
                grads, dP = det.backward(P[:3], join=False)
                return losses, grads, dP
            return roi_half
        halves = [half(b) for b in range(len(images))]
        out = self.prop.forward_backward_batch(images, gt_boxes, memories, world_size=world_size, reduce_counts=reduce_counts,
                                               roi_halves=halves)
        if det.speculate and det.checks and bool(torch.cat(det.checks).any().cpu()):
            self._exact_sizes.add((H, W))
            self.repeated_frames += len(images)
            det.speculate, det.checks = False, []
            out = self.prop.forward_backward_batch(images, gt_boxes, memories, world_size=world_size, reduce_counts=reduce_counts,
                                                   roi_halves=halves)
        return out


class Trainer(ProposalTrainer):
    """One optimizer over every trainable parameter of the recurrent detector (build_custom_optimizer's groups, custom_solver.py:19-79)
    and `forward_model`'s full loss dict per step: `ProposalTrainer` with the ROI heads' half."""

    def __init__(self, model, sd: Dict[str, torch.Tensor]):
        super().__init__(model, sd, roi_heads=True)
        self._acc = None
        self._copy_stream = None
        # frames of a training batch that share one pass of the trunk half, forward and backward (`forward_backward_frames`);
        # 1 = every frame on its own (the two give the same losses, and gradients up to the order of summation over the frames)
        self.trunk_batch = 4
        model.trainer = self                      # `model.train(); model(data)` reaches `forward_backward_frames` (meta_arch.forward)

    @staticmethod
    def _gt(frame):
        inst = frame["instances"]
        if isinstance(inst, dict):
            boxes, classes = inst["gt_boxes"], inst["gt_classes"]
        else:
            boxes, classes = inst.gt_boxes, inst.gt_classes
        boxes = boxes.tensor if hasattr(boxes, "tensor") else boxes
        return torch.as_tensor(boxes, dtype=torch.float32).reshape(-1, 4), torch.as_tensor(classes).reshape(-1).to(torch.int32)

    def forward_backward_frames(self, batched_inputs, generator: Optional[torch.Generator] = None):
        """The training branch of `CustomRCNNRecurrent.forward` (custom_rcnn.py:435-461): for every frame of every sequence the memory
        the loader hands over (`frame['memory']` accumulated features [N,512], `frame['observations']` [N], `frame['proj_indices']`
        [H,W]; loader.py:199-223) is normalised by its observation counts (`create_implicit_memory`, :762-774 -> the a4 kernel) and
        `forward_model` runs on the frame; the loss terms are SUMMED over the frames (:455-460), and so are the gradients (one
        `losses.backward()` per call in `train_mp3d.py:619-625`) -> the summed loss dict; `optimizer_step` applies them."""
        dev = self.dev
        total: Dict[str, torch.Tensor] = {}
        acc = None
        frames = [frame for seq in batched_inputs for frame in seq]
        main = torch.cuda.current_stream(dev)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=dev)

        def stage(frame):
            """The frame's tensors on the device, copied on the copy stream (the memory table alone is 82 MB at 200 x 200 cells:
            ~2 ms per frame on the compute stream otherwise) while the frame before computes -> (tensors, event)."""
            with torch.cuda.stream(self._copy_stream):
                # (the loader's image is a permuted HWC view, train_mp3d.py:469: contiguous CHW on the device)
                t = {"image": torch.as_tensor(frame["image"]).to(dev, non_blocking=True).contiguous()}
                gt_boxes, gt_classes = self._gt(frame)
                t["gt_boxes"] = gt_boxes.to(dev, non_blocking=True).contiguous()
                t["gt_classes"] = gt_classes.to(dev, non_blocking=True)
                if self.model.memory_type == "implicit_memory":
                    if frame.get("observations") is None:
                        raise ValueError("training reads frame['observations'] (custom_rcnn.py:765): point MODEL.SEMMAP_PATH at the "
                                         "memory snapshots (`impicit_memory` / `observations` per episode file, loader.py:213-223)")
                    H, W = int(t["image"].shape[1]), int(t["image"].shape[2])
                    t["mem"] = torch.as_tensor(frame["memory"]).to(dev, torch.float32, non_blocking=True).contiguous()
                    t["obs"] = torch.as_tensor(frame["observations"]).to(dev, torch.float32, non_blocking=True).reshape(-1).contiguous()
                    t["proj"] = torch.as_tensor(frame["proj_indices"]).to(dev, non_blocking=True).reshape(H, W).to(torch.int32).contiguous()
                ev = torch.cuda.Event()
                ev.record(self._copy_stream)
            return t, ev

        # consecutive frames of one image size share a pass of the memory-independent trunk half (`trunk_batch` frames at a time)
        B = max(1, int(self.trunk_batch))
        batches: List[list] = []
        for frame in frames:
            shape = tuple(torch.as_tensor(frame["image"]).shape)
            if batches and len(batches[-1]) < B and batches[-1][0][0] == shape:
                batches[-1].append((shape, frame))
            else:
                batches.append([(shape, frame)])

        def stage_batch(batch):
            return [stage(frame) for _, frame in batch]

        staged = stage_batch(batches[0]) if batches else None
        for i in range(len(batches)):
            cur = staged
            staged = stage_batch(batches[i + 1]) if i + 1 < len(batches) else None    # the next frames' copies run beside this step
            ts = []
            for t, ev in cur:
                main.wait_event(ev)
                for v in t.values():
                    v.record_stream(main)
                ts.append(t)
            mems = [(ops.memory_normalize_f16(t["mem"], t["obs"]), t["proj"]) if "mem" in t else None for t in ts]
            if len(ts) == 1:
                t = ts[0]
                losses, grads = self.fm.forward_backward(t["image"], t["gt_boxes"], t["gt_classes"], memory=mems[0], generator=generator)
                loss_list = [losses]
            else:
                loss_list, grads = self.fm.forward_backward_batch([t["image"] for t in ts], [t["gt_boxes"] for t in ts],
                                                                  [t["gt_classes"] for t in ts], mems, generator=generator)
            gl = [self.step_getters[g["name"]](grads) for g in self.groups]
            if acc is None:
                acc = [None if g_ is None else g_.clone() for g_ in gl]
            else:
                # the frames' gradients are summed as the single `losses.backward()` of the reference sums them; one multi-tensor
                # add for the tensors both sides have
                both = [(a, g_) for a, g_ in zip(acc, gl) if a is not None and g_ is not None]
                if both:
                    torch._foreach_add_([a for a, _ in both], [g_ for _, g_ in both])
                acc = [a if g_ is None else (g_.clone() if a is None else a) for a, g_ in zip(acc, gl)]
            for losses in loss_list:
                for k, v in losses.items():
                    total[k] = v.clone() if k not in total else total[k] + v
        self._acc = acc
        return total

    def optimizer_step(self, lr_factor: float = 1.0):
        """`optimizer.step()` + `scheduler.step()`'s factor (train_mp3d.py:625,633) on the gradients of the last `forward_backward_frames`."""
        if self._acc is None:
            raise RuntimeError("optimizer_step without gradients: call forward_backward_frames (or model(data) in training mode) first")
        self.opt.step(self._acc, lr_factor=lr_factor)
        for f in self.after:
            f()
        self._acc = None
        self.iteration += 1
