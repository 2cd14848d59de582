"""Evidence for the memory WRITE path at full size (custom_rcnn.py:884-936), shared by test_write_parity_gpu.py and
test_tolerance_gpu.py.  TEST INFRASTRUCTURE: the oracle is the checker, the HIP model is the thing checked.

Two statements are separated here:

1. *Given the HIP frame's own memory instances* (its kept proposal rows, their CLIP-space features and their pasted masks, read
   back from the device after the frame), the state the HIP write leaves must be what `oracle.memory.memory_write_sparse`
   computes from exactly those inputs: written-cell set bit-exact, values to 1e-5 relative on every cell.
2. Where the HIP frame's pasted masks differ from the oracle frame's, every differing pixel must be a knife-edge decision of the
   0.5 threshold: `|p - 0.5|` in the oracle's own bilinear sample no larger than the measured difference of the two sides' pasted
   probabilities of that instance -- the only way two fp32 implementations of the same frame can disagree on a mask pixel.  A single such flip shifts the phase of the every-8th-observed-pixel rule (custom_rcnn.py:913-914)
   for all later pixels, so it legitimately changes the written values of the frame (not a kernel error).
"""
from __future__ import annotations

from typing import Dict

import torch

from oracle import memory as OM
from oracle import ops as OO

# A flipped pixel is explained by the MEASURED difference of the two pasted probabilities, not by a constant: for every paired
# instance the band is max over the image of |p_hip - p_oracle|, both sampled by the oracle's own bilinear paste -- p_hip from the
# HIP frame's 28x28 probabilities and box, p_oracle from the oracle frame's -- plus the rounding of the product's paste kernel
# against that sampler.  Two implementations that agree on the probabilities to `band` can only disagree on a pixel whose
# probability lies within `band` of the threshold.
PASTE_KERNEL_SLACK = 4 * 2.0 ** -24        # eod_paste_masks against oracle.ops.paste_masks_prob on the SAME inputs: a few ulp of 0.5
# What the derived band itself may be when both sides started the frame from one state.  Measured over 15 frames at 128x160,
# 480x640 and 640x640 on two sequences (profiles/r04_parity_report_*.json): 1.6e-5 .. 3.6e-5, of which the 28x28 probabilities
# themselves differ by <= 4.9e-6; the rest is the proposal boxes (unclipped, coordinates up to ~1e3 px: they agree to 1.2e-3 px = 1e-6
# relative) shifting the sampling grid across a steep mask edge.  Every flipped pixel lay within 1.4e-5 of the threshold.
BAND_CAP_IDENTICAL_STATE = 5e-5
MASK28_DIFF_CAP = 1e-5                     # the mask head's own agreement on the 28x28 probabilities


def hip_write_inputs(model, H: int, W: int) -> Dict[str, torch.Tensor]:
    """What the HIP memory write of the frame just run consumed, read back from the device: unique kept proposal rows
    (ascending, custom_rcnn.py:875), their boxes, 50 x normalised stage-0 features, 28x28 mask probabilities, and the masks pasted
    by the product's own paste kernel (`eod_paste_masks`, the arithmetic `mw_coverage` / `mw_accumulate` apply per pixel)."""
    from embodied_object_detection_amd import ops
    prop_boxes, prop_masks, rows, cnt, proj = model._last_write
    torch.cuda.synchronize()
    k = int(cnt.item())
    urows = torch.unique(rows[:k].long())
    K = int(urows.numel())
    boxes = prop_boxes[urows].contiguous()
    pasted = torch.zeros((max(K, 1), H, W), dtype=torch.uint8, device=prop_boxes.device)
    if K:
        count = torch.tensor([K], dtype=torch.int32, device=prop_boxes.device)
        ops.paste_masks(prop_masks, boxes, urows.to(torch.int32).contiguous(), count, K, H, W, 0.5, pasted)
        torch.cuda.synchronize()
    return dict(rows=urows.cpu(), boxes=boxes.cpu(), featn=model.roi_heads.featn0[urows].cpu(),
                masks28=prop_masks[urows].cpu(), pasted=pasted[:K].cpu().bool(), proj=proj.cpu().long(), K=K)


def check_write_against_oracle(model, mem_before: torch.Tensor, obs_before: torch.Tensor, H: int, W: int, rel: float = 1e-5) -> dict:
    """Statement 1.  `mem_before` / `obs_before`: the state the HIP model held before the frame (CPU copies)."""
    ev = hip_write_inputs(model, H, W)
    n_cells = mem_before.shape[0]
    got_mem, got_obs = model.implicit_memory.cpu(), model.observations.cpu()
    exp_mem = mem_before.clone()
    observed = torch.zeros((n_cells,), dtype=torch.bool)
    if ev["K"]:
        mean, observed = OM.memory_write_sparse(ev["featn"], ev["pasted"], ev["proj"], n_cells)
        exp_mem[observed] = exp_mem[observed] + mean
    exp_obs = obs_before.clone()
    if ev["K"]:                                           # update_implicit_memory returns early without instances (:689-690)
        exp_obs[torch.unique(ev["proj"])] += 1
    written = (got_mem != mem_before).any(dim=1)
    scale = exp_mem.abs().max(dim=1).values.clamp_min(1.0)
    cell_rel = (got_mem - exp_mem).abs().max(dim=1).values / scale
    return dict(K=ev["K"], written_cells=int(written.sum()), expected_cells=int(observed.sum()),
                cell_set_exact=bool(torch.equal(written, observed)), max_rel_err=float(cell_rel.max()),
                cells_over_tol=int((cell_rel > rel).sum()), observations_exact=bool(torch.equal(got_obs, exp_obs)), evidence=ev)


def mask_flip_attribution(ev: Dict[str, torch.Tensor], oracle_last: dict, H: int, W: int) -> dict:
    """Statement 2.  Pairs the HIP frame's memory instances with the oracle frame's (same box to 1e-2 px), derives every pair's
    band from the measured difference of their pasted probabilities (see above) and classifies every pasted-mask pixel on which
    the two disagree by the oracle's own pre-threshold sample against THAT pair's band."""
    out = dict(instances_hip=ev["K"], instances_oracle=int(oracle_last.get("K", 0)), paired=0, unpaired=0, flipped_pixels=0,
               flips_outside_band=0, flips_outside_pixel_band=0, max_flip_distance=0.0, max_band=0.0, max_m28_diff=0.0, max_box_diff_px=0.0, masks_identical=False)
    if not ev["K"] or not out["instances_oracle"]:
        out["masks_identical"] = ev["K"] == out["instances_oracle"]
        return out
    ob, om28, opasted = oracle_last["boxes"], oracle_last["masks28"], oracle_last["masks"]
    identical = ev["K"] == out["instances_oracle"]
    pairs = []
    for i in range(ev["K"]):
        d = (ob - ev["boxes"][i][None]).abs().max(dim=1).values
        j = int(d.argmin())
        if float(d[j]) > 1e-2:        # the same proposal: its box agrees to the proposal tolerance (unclipped boxes, up to ~1e3 px)
            out["unpaired"] += 1
            identical = False
            continue
        pairs.append((i, j, float(d[j])))
    out["paired"] = len(pairs)
    if pairs:
        ii = torch.tensor([p[0] for p in pairs])
        jj = torch.tensor([p[1] for p in pairs])
        # both sides' pasted probabilities through the oracle's sampler, all pairs at once (16 instances per grid_sample)
        p_or = OO.paste_masks_prob(om28[jj].reshape(-1, 28, 28).float(), ob[jj], (H, W))
        p_hip = OO.paste_masks_prob(ev["masks28"][ii].reshape(-1, 28, 28).float(), ev["boxes"][ii], (H, W))
        dp = (p_hip - p_or).abs()
        bands = dp.flatten(1).max(dim=1).values + PASTE_KERNEL_SLACK
        out["max_band"] = float(bands.max())
        out["max_m28_diff"] = float((ev["masks28"][ii].reshape(-1, 784) - om28[jj].reshape(-1, 784)).abs().max())
        out["max_box_diff_px"] = max(p[2] for p in pairs)
        diff = ev["pasted"][ii] != opasted[jj]
        per = diff.flatten(1).sum(dim=1)
        if int(per.sum()):
            identical = False
            dist = (p_or - 0.5).abs()
            out["flipped_pixels"] = int(per.sum())
            out["flips_outside_band"] = int((diff & (dist > bands[:, None, None])).sum())
            # the sharper, per-pixel statement: the two probabilities AT the flipped pixel differ by at least its distance to 0.5
            out["flips_outside_pixel_band"] = int((diff & (dist > dp + PASTE_KERNEL_SLACK)).sum())
            out["max_flip_distance"] = float(dist[diff].max())
    out["masks_identical"] = bool(identical and out["unpaired"] == 0 and out["paired"] == out["instances_oracle"])
    return out


def count_mask_differences(ev: Dict[str, torch.Tensor], oracle_last: dict) -> int:
    """Pasted-mask pixels (plus 1 per unpaired instance) on which a HIP frame's memory instances differ from the oracle frame's:
    0 = the two frames write from identical masks."""
    if ev["K"] != int(oracle_last.get("K", 0)):
        return abs(ev["K"] - int(oracle_last.get("K", 0))) or 1
    n = 0
    for i in range(ev["K"]):
        d = (oracle_last["boxes"] - ev["boxes"][i][None]).abs().max(dim=1).values
        j = int(d.argmin())
        n += 1 if float(d[j]) > 1e-2 else int((ev["pasted"][i] != oracle_last["masks"][j]).sum())
    return n
