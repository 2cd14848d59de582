"""Deterministic inputs shared by the golden generator (development container only) and the tests.

Everything here is seeded numpy/torch data; no reference code.  Keeping the *inputs* procedural keeps
the committed fixtures small: `.npz` files hold the reference's OUTPUTS plus the few inputs that
are not regenerated here.
"""
from __future__ import annotations

import math

import numpy as np
import torch

H480, W640 = 480, 640


def projector_case(seed: int = 11, H: int = 96, W: int = 128):
    rng = np.random.RandomState(seed)
    # smooth-ish depth in metres
    base = 3.0 + 2.0 * np.sin(np.linspace(0, 3, W))[None, :] * np.cos(np.linspace(0, 2, H))[:, None]
    depth = np.clip(base + 0.3 * rng.randn(H, W), 0.5, 10.0).astype(np.float32)
    xyzhe = np.array([[2.5, 1.5, 3.25, 0.35, math.pi]], dtype=np.float32)
    vfov = 67.5 * math.pi / 180.0
    world_shift = np.array([0.25, 0.0, -0.5], dtype=np.float32)     # ProjectorUtils.world_shift_origin
    map_shift = np.array([-5.0, 0.0, -5.0], dtype=np.float32)       # map_world_shift
    return dict(depth=depth, xyzhe=xyzhe, vfov=vfov, world_shift=world_shift, map_shift=map_shift,
                cell=0.02 * 10, map_w=60, map_h=50)


def memory_state_case(seed: int = 5, n_cells: int = 300):
    g = torch.Generator().manual_seed(seed)
    mem = torch.randn((n_cells, 512), generator=g) * 20.0
    obs = torch.randint(0, 5, (n_cells,), generator=g).to(torch.float32)
    return mem, obs


def instance_masks_case(seed: int = 9, K: int = 6, H: int = H480, W: int = W640, n_cells: int = 300):
    """K elliptical instance masks (overlapping), 512-d features, a proj index image."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    masks = np.zeros((K, H, W), dtype=bool)
    for k in range(K):
        cy, cx = rng.uniform(0.2 * H, 0.8 * H), rng.uniform(0.2 * W, 0.8 * W)
        ry, rx = rng.uniform(20, 90), rng.uniform(20, 120)
        masks[k] = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
    feats = (rng.randn(K, 512) * 3.0).astype(np.float32)
    # piecewise-constant projection image with some noise, values in [0, n_cells)
    proj = ((yy // 24) * 14 + (xx // 46)) % n_cells
    noise = rng.randint(0, n_cells, size=(H, W))
    proj = np.where(rng.rand(H, W) < 0.05, noise, proj).astype(np.int64)
    return torch.from_numpy(masks), torch.from_numpy(feats), torch.from_numpy(proj)


def fpn_case(seed: int = 3, H: int = 64, W: int = 96, n_cells: int = 300):
    """Bottom-up features C3..C5 for an HxW image, fp16 memory, proj indices, random FPN weights."""
    g = torch.Generator().manual_seed(seed)
    c3 = torch.randn((1, 512, H // 8, W // 8), generator=g)
    c4 = torch.randn((1, 1024, H // 16, W // 16), generator=g)
    c5 = torch.randn((1, 2048, H // 32, W // 32), generator=g)
    mem = (torch.randn((n_cells, 512), generator=g) * 10.0)
    mem = 50.0 * torch.nn.functional.normalize(mem, dim=1)
    mem[::7] = 0
    proj = torch.randint(0, n_cells, (H, W), generator=g)
    # make it blocky so neighbouring pixels often share a cell (like real projections)
    blocky = (torch.arange(H)[:, None] // 5) * 13 + (torch.arange(W)[None, :] // 7)
    proj = torch.where(torch.rand((H, W), generator=g) < 0.7, blocky % n_cells, proj).to(torch.int64)
    return c3, c4, c5, mem, proj


def fpn_weights(seed: int = 4):
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def rn(*shape, std):
        return torch.randn(shape, generator=g) * std

    for lvl, cin in ((3, 512), (4, 1024), (5, 2048)):
        sd[f"backbone.fpn_lateral{lvl}.weight"] = rn(256, cin, 1, 1, std=(1.0 / cin) ** 0.5)
        sd[f"backbone.fpn_lateral{lvl}.bias"] = rn(256, std=0.1)
        sd[f"backbone.fpn_output{lvl}.weight"] = rn(256, 256, 3, 3, std=(1.0 / 2304) ** 0.5)
        sd[f"backbone.fpn_output{lvl}.bias"] = rn(256, std=0.1)
    for n in ("p6", "p7"):
        sd[f"backbone.top_block.{n}.weight"] = rn(256, 256, 3, 3, std=(1.0 / 2304) ** 0.5)
        sd[f"backbone.top_block.{n}.bias"] = rn(256, std=0.1)
    for i in (1, 2, 3):
        sd[f"backbone.map_merge_projection{i}.weight"] = rn(256, 512, 1, 1, std=(1.0 / 512) ** 0.5 * 0.05)
        sd[f"backbone.map_merge_projection{i}.bias"] = rn(256, std=0.01)
    return sd


def centernet_head_case(seed: int = 6):
    g = torch.Generator().manual_seed(seed)
    feats = [torch.randn((1, 256, h, w), generator=g) for (h, w) in ((8, 12), (4, 6), (2, 3), (1, 2), (1, 1))]
    return feats


def centernet_head_weights(seed: int = 7):
    g = torch.Generator().manual_seed(seed)
    h = "proposal_generator.centernet_head"
    sd = {}
    for i in range(4):
        sd[f"{h}.bbox_tower.{3 * i}.weight"] = torch.randn((256, 256, 3, 3), generator=g) * 0.03
        sd[f"{h}.bbox_tower.{3 * i}.bias"] = torch.randn((256,), generator=g) * 0.1
        sd[f"{h}.bbox_tower.{3 * i + 1}.weight"] = torch.rand((256,), generator=g) + 0.5
        sd[f"{h}.bbox_tower.{3 * i + 1}.bias"] = torch.randn((256,), generator=g) * 0.1
    sd[f"{h}.bbox_pred.weight"] = torch.randn((4, 256, 3, 3), generator=g) * 0.03
    sd[f"{h}.bbox_pred.bias"] = torch.full((4,), 2.0)
    sd[f"{h}.agn_hm.weight"] = torch.randn((1, 256, 3, 3), generator=g) * 0.03
    sd[f"{h}.agn_hm.bias"] = torch.full((1,), -2.0)
    for l in range(5):
        sd[f"{h}.scales.{l}.scale"] = torch.tensor([0.8 + 0.1 * l])
    return sd


def zs_case(seed: int = 8, R: int = 17):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((R, 1024), generator=g)
    w = torch.randn((512, 1024), generator=g) * (1.0 / 1024) ** 0.5
    b = torch.randn((512,), generator=g) * 0.1
    return x, w, b


def robot_case(seed: int = 21, H: int = 120, W: int = 160):
    """Depth in millimetres (uint16-like), planar robot pose (x, y, theta)."""
    rng = np.random.RandomState(seed)
    base = 2500 + 1500 * np.sin(np.linspace(0, 4, W))[None, :] * np.cos(np.linspace(0, 3, H))[:, None]
    depth_mm = np.clip(base + 200 * rng.randn(H, W), 0, 9000).astype(np.uint16)   # cv2.IMREAD_ANYDEPTH gives uint16 mm
    depth_mm[:3] = 0                         # no-return pixels
    pose = np.array([1.75, -2.5, 0.6], dtype=np.float32)
    return dict(depth_mm=depth_mm, pose=pose)


# ---------------------------------------------------------------------------------------------------------
# round 2: decode / cascade / memory-update cases
# ---------------------------------------------------------------------------------------------------------
DECODE_HW = (256, 320)             # P3 = 32x40 = 1280 positions > PRE_NMS_TOPK_TEST (1000): the per-level top-k cut is exercised


def centernet_decode_case(variant: str = "plain", seed: int = 31):
    """Raw head outputs (agn_hm logits, ReLU'd regressions in stride units) for a 256x320 image.

    `plain`: continuous logits (no exact ties anywhere).  `ties`: a band of logits is quantised so that many positions share a
    score; boxes are small (little NMS suppression), so more than 256 survive and the `>= kth` rule has ties to keep."""
    g = torch.Generator().manual_seed(seed + (0 if variant == "plain" else 1000))
    H, W = DECODE_HW
    agn, reg = [], []
    for l, s in enumerate((8, 16, 32, 64, 128)):
        h, w = (H + s - 1) // s, (W + s - 1) // s
        a = torch.randn((1, 1, h, w), generator=g) * 1.5 - 1.0
        a[0, 0, ::5, ::7] = -20.0                      # sigmoid < 1e-4: not a candidate
        if variant == "ties":
            a = torch.round(a * 2.0) / 2.0              # half-integer logits: heavy ties, incl. at the kth value
            r = torch.rand((1, 4, h, w), generator=g) * 0.6 + 0.1
        else:
            r = torch.rand((1, 4, h, w), generator=g) * 3.0
            r[0, 2:, 1::3, :] = 0.0                     # zero extent to the right / bottom: the +0.01 minimum size applies
        agn.append(a)
        reg.append(r)
    return agn, reg


DECODE_TRAIN_HW = (512, 640)       # P3 = 64x80 = 5120 positions > PRE_NMS_TOPK_TRAIN (4000); 6820 positions > POST_NMS_TOPK_TRAIN (2000)


def centernet_decode_train_case(seed: int = 77):
    """Raw head outputs for a 512x640 image, continuous logits: the training thresholds' per-level cut (4000) and post-NMS cut (2000)
    are both reached; boxes of 2-6 strides overlap their grid neighbours enough for NMS 0.9 to suppress some."""
    g = torch.Generator().manual_seed(seed)
    H, W = DECODE_TRAIN_HW
    agn, reg = [], []
    for l, s in enumerate((8, 16, 32, 64, 128)):
        h, w = (H + s - 1) // s, (W + s - 1) // s
        a = torch.randn((1, 1, h, w), generator=g) * 1.5 - 0.5
        a[0, 0, ::9, ::11] = -20.0                     # sigmoid < 1e-4: not a candidate
        r = torch.rand((1, 4, h, w), generator=g) * 2.0 + 1.0
        r = r * (1.0 + 0.02 * torch.randn((1, 1, h, w), generator=g))
        # a smooth field: neighbouring positions predict nearly the same box corners -> IoU > 0.9 pairs exist
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        r[0, 0] = 3.0 + (xx % 4) * 1.0
        r[0, 2] = 6.0 - (xx % 4) * 1.0
        agn.append(a)
        reg.append(r)
    return agn, reg


CASCADE_HW = (128, 160)


def cascade_case(seed: int = 41, R: int = 48):
    """P3..P5 of a 128x160 image, R proposals (xyxy, some partly outside the image) with CenterNet scores."""
    g = torch.Generator().manual_seed(seed)
    H, W = CASCADE_HW
    feats = [torch.randn((1, 256, H // s, W // s), generator=g) for s in (8, 16, 32)]
    cx = torch.rand((R,), generator=g) * W
    cy = torch.rand((R,), generator=g) * H
    bw = torch.rand((R,), generator=g) * 90 + 4
    bh = torch.rand((R,), generator=g) * 70 + 4
    boxes = torch.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], dim=1)
    scores = torch.rand((R,), generator=g) * 0.9 + 0.05
    return feats, boxes, scores


def cascade_weights(seed: int = 42):
    """Reference-keyed weights of the three box heads / predictors (FC_DIM 1024, 7x7x256 input)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def rn(*shape, std):
        return torch.randn(shape, generator=g) * std

    for k in range(3):
        sd[f"roi_heads.box_head.{k}.fc1.weight"] = rn(1024, 12544, std=(2.0 / 12544) ** 0.5)
        sd[f"roi_heads.box_head.{k}.fc1.bias"] = rn(1024, std=0.05)
        sd[f"roi_heads.box_head.{k}.fc2.weight"] = rn(1024, 1024, std=(2.0 / 1024) ** 0.5)
        sd[f"roi_heads.box_head.{k}.fc2.bias"] = rn(1024, std=0.05)
        p = f"roi_heads.box_predictor.{k}"
        sd[f"{p}.cls_score.linear.weight"] = rn(512, 1024, std=(1.0 / 1024) ** 0.5)
        sd[f"{p}.cls_score.linear.bias"] = rn(512, std=0.05)
        sd[f"{p}.bbox_pred.0.weight"] = rn(1024, 1024, std=(2.0 / 1024) ** 0.5)
        sd[f"{p}.bbox_pred.0.bias"] = rn(1024, std=0.05)
        sd[f"{p}.bbox_pred.2.weight"] = rn(4, 1024, std=0.05)        # large enough that the cascade moves the boxes
        sd[f"{p}.bbox_pred.2.bias"] = rn(4, std=0.2)
    return sd


N_CELLS_200 = 200 * 200


def memory_update_case(seed: int = 51, n_frames: int = 4, R: int = 40):
    """Canned per-frame proposals (what `inference` hands to `update_implicit_memory`) and projection images for a 480x640
    camera over the 200x200 fallback map.  Frame f looks at a window of the map shifted by a few cells, so cells are re-observed."""
    rng = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(seed)
    H, W = H480, W640
    yy, xx = np.mgrid[0:H, 0:W]
    frames = []
    for f in range(n_frames):
        oy, ox = 40 + 3 * f, 30 + 5 * f
        iz = oy + yy // 7
        ix = ox + xx // 7
        proj = iz * 200 + ix
        noise = rng.randint(0, N_CELLS_200, size=(H, W))
        proj = np.where(rng.rand(H, W) < 0.01, noise, proj).astype(np.int32)[..., None]      # [H,W,1] int32 as the loader gives it
        cx = torch.rand((R,), generator=g) * W
        cy = torch.rand((R,), generator=g) * H
        bw = torch.rand((R,), generator=g) * 200 + 20
        bh = torch.rand((R,), generator=g) * 160 + 20
        boxes = torch.stack([(cx - bw / 2).clamp(0, W), (cy - bh / 2).clamp(0, H), (cx + bw / 2).clamp(0, W), (cy + bh / 2).clamp(0, H)], 1)
        scores = torch.rand((R,), generator=g) * 0.9 + 0.05
        scores[3] = 1.0                                   # a GT-style row (score == 1) that `< 1` must drop
        feat = torch.randn((R, 512), generator=g) * 2.0
        low = torch.randn((R, 1, 7, 7), generator=g) * 3.0
        m28 = torch.sigmoid(torch.nn.functional.interpolate(low, size=(28, 28), mode="bilinear", align_corners=False))
        frames.append(dict(proj=proj, boxes=boxes, scores=scores, feat=feat, masks28=m28))
    return frames


def digest_matrix(seed: int = 99, cols: int = 8) -> np.ndarray:
    """Fixed [512, cols] float64 matrix: fixtures keep `rows @ digest_matrix` instead of full 512-d rows."""
    return np.random.RandomState(seed).randn(512, cols)


# ------------------------------------------------------------------------------------------------------
# on-disk episodes (memory_data/*.h5 + sensor_data/*.h5 + JPEGImages/) for the loader / eval-driver fixtures
# ------------------------------------------------------------------------------------------------------
MP3D_MINI = dict(H=32, W=48, n_cells=40, n_jpeg=5,
                 # (scene prefix, number of episodes): episode ids are 0..n-1, so 'sA_1_10' must sort after 'sA_1_9'
                 scenes=(("sA_1", 50), ("sB_22", 50), ("sC_3", 10)))
MP3D_EPISODE_LENGTHS = (3, 6, 11, 23)           # by episode id % 4; 23 > max_sequence_length 20 (loader.py:74)


def mp3d_mini_jpeg_pixels(seed: int = 61):
    """The RGB arrays the generator encodes as JPEG (the encoded bytes travel in the fixture, the decoder is PIL on both sides)."""
    rng = np.random.RandomState(seed)
    c = MP3D_MINI
    base = rng.randint(0, 255, size=(c["n_jpeg"], c["H"] // 4, c["W"] // 4, 3))
    return np.repeat(np.repeat(base, 4, axis=1), 4, axis=2).astype(np.uint8)      # blocky: survives JPEG well


def mp3d_mini_episode(scene: str, ep: int):
    """Contents of one episode file pair, seeded by its name: (proj [T,H,W,1] i32, memory [n,256] f32, semmap_gt [n] i32,
    segmentation [T,H,W] u8, detection_data strings [T])."""
    c = MP3D_MINI
    seed = (sum(ord(ch) for ch in scene) * 1009 + ep * 7919 + 13) % (2 ** 31 - 1)
    rng = np.random.RandomState(seed)
    T = MP3D_EPISODE_LENGTHS[ep % 4]
    proj = rng.randint(0, c["n_cells"], size=(T, c["H"], c["W"], 1)).astype(np.int32)
    mem = rng.rand(c["n_cells"], 256).astype(np.float32)
    sem = rng.randint(0, 13, size=(c["n_cells"],)).astype(np.int32)
    seg = rng.randint(0, 21, size=(T, c["H"], c["W"])).astype(np.uint8)
    recs = []
    for i in range(T):
        n = int(rng.randint(0, 5))                       # 0 boxes happens: empty GT must survive the whole chain
        boxes = [[float(rng.randint(0, 20)) + 0.5 * float(rng.randint(0, 2)), float(rng.randint(0, 12)),
                  float(rng.randint(2, 20)) + 0.25, float(rng.randint(2, 14))] for _ in range(n)]
        classes = [int(rng.randint(0, 20)) for _ in range(n)]      # incl. ids outside the evaluated subset (1, 8, 10, 11, 18)
        fn = f"img_{int(rng.randint(0, c['n_jpeg']))}.jpg"
        # str(dict) as build_data.py:245 writes it
        recs.append(str({"file_name": fn, "image": "x", "gt_boxes": boxes, "gt_classes": classes}))
    return proj, mem, sem, seg, recs


def mp3d_mini_names():
    return [f"{scene}_{e}.h5" for scene, n in MP3D_MINI["scenes"] for e in range(n)]


def write_mp3d_mini(root: str, jpeg_bytes):
    """Writes the dataset under `root` through the product's HDF5 binding (data/h5io.py); `jpeg_bytes[i]` -> JPEGImages/img_i.jpg."""
    import os
    from embodied_object_detection_amd.data import h5io
    c = MP3D_MINI
    for d in ("memory_data", "sensor_data", "JPEGImages"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    for i, b in enumerate(jpeg_bytes):
        with open(os.path.join(root, "JPEGImages", f"img_{i}.jpg"), "wb") as fh:
            fh.write(bytes(bytearray(np.asarray(b, dtype=np.uint8).tolist())))
    for scene, n in c["scenes"]:
        for e in range(n):
            proj, mem, sem, seg, recs = mp3d_mini_episode(scene, e)
            name = f"{scene}_{e}.h5"
            with h5io.H5File(os.path.join(root, "memory_data", name), "w") as f:
                f.write("memory_features", mem)
                f.write("proj_indices", proj)
                f.write("semmap_gt", sem)
            with h5io.H5File(os.path.join(root, "sensor_data", name), "w") as f:
                f.write("rgb", np.zeros((len(recs), c["H"], c["W"], 3), dtype=np.uint8))
                f.write("segmentation_data", seg)
                f.write_strings("detection_data", recs)
