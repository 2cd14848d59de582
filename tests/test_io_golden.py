"""`data/mp3d.py` and `engine/eval_loop.py` against the reference's OWN loader and driver.

The fixtures `tests/golden/mp3d_loader.npz` / `mp3d_driver.json` are outputs of `Detic/SMNet/loader.py` (`SMNetDetectionLoader`,
`collate_smnet`) and `Detic/train_mp3d.py` (`map_mp3d_batch_to_coco`, `mp3d_inference_on_dataset` with a stub model and a stub
evaluator) run in the development container by `tests/golden/gen_golden_io.py` on the episode dataset that
`tests/golden/_inputs.py::write_mp3d_mini` writes (110 episode file pairs in 3 scenes; the JPEG bytes travel in the fixture).
Here the same dataset is written again and read by the product's mirror."""
import json
import os
import zlib

import numpy as np
import pytest
import torch

import _inputs as I
from embodied_object_detection_amd.data import h5io

pytestmark = pytest.mark.skipif(not h5io.available(), reason="no libhdf5 in this image")


def crc(a) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


@pytest.fixture(scope="module")
def fx(golden_dir):
    z = np.load(os.path.join(golden_dir, "mp3d_loader.npz"))
    with open(os.path.join(golden_dir, "mp3d_driver.json")) as fh:
        drv = json.load(fh)
    return z, drv


@pytest.fixture(scope="module")
def root(tmp_path_factory, fx):
    r = str(tmp_path_factory.mktemp("mp3d_mini"))
    I.write_mp3d_mini(r, [fx[0][f"jpeg_{i}"] for i in range(I.MP3D_MINI["n_jpeg"])])
    return r


def test_loader_matches_the_reference_loader(fx, root):
    """loader.py:97-117 (ordering, longterm duplication), :199-303 (datasets, GT parsing + class filter + XYXY, JPEG decode,
    reset rule, keys, dtypes, fallbacks)."""
    from embodied_object_detection_amd.data.mp3d import SMNetDetectionLoader, collate_smnet
    z, _ = fx
    for tt in ("default", "episodic", "longterm"):
        ld = SMNetDetectionLoader(data_path=root, test_type=tt, memory_type="implicit_memory", semmap_path="")
        assert ld.files == z[f"files_{tt}"].tolist(), tt
        resets, lengths = [], []
        for i in range(len(ld)):
            if tt != "default" and i >= 12 and not (tt == "longterm" and 45 <= i < 56):
                continue
            ep = ld[i]
            lengths.append(len(ep))
            resets += [bool(f["memory_reset"]) for f in ep]
        assert lengths == z[f"lengths_{tt}"].tolist(), tt              # incl. the cap at 20 frames (loader.py:74,246)
        assert resets == z[f"resets_{tt}"].tolist(), tt                # loader.py:289-293
    ld = SMNetDetectionLoader(data_path=root, test_type="default", memory_type="implicit_memory", semmap_path="")
    names, boxes, classes, nbox, img_crc, proj_crc, mem_crc = [], [], [], [], [], [], []
    f0 = fe = None
    for i in range(len(ld)):
        ep = collate_smnet([ld[i]])[0]
        mem_crc.append(crc(ep[0]["memory_features"]))
        for f in ep:
            assert set(f) == {"file_name", "sequence_name", "gt_boxes", "gt_classes", "image", "proj_indices", "memory_reset",
                              "memory_features", "observations"}
            assert f["observations"] is None and f["sequence_name"] == ld.files[i]
            f0 = f0 if f0 is not None else f
            if fe is None and len(f["gt_classes"]) == 0:
                fe = f
            names.append(f["file_name"])
            nbox.append(len(f["gt_classes"]))
            boxes += np.asarray(f["gt_boxes"], dtype=np.float64).reshape(-1, 4).tolist()
            classes += np.asarray(f["gt_classes"]).reshape(-1).tolist()
            img_crc.append(crc(f["image"]))
            proj_crc.append(crc(f["proj_indices"]))
    assert names == z["file_names"].tolist()
    assert nbox == z["n_boxes"].tolist() and classes == z["gt_classes"].tolist()
    assert np.array_equal(np.array(boxes, dtype=np.float64).reshape(-1, 4), z["gt_boxes"])          # XYWH -> XYXY, class filter
    assert img_crc == z["image_crc"].tolist(), "decoded JPEG pixels differ from the reference loader's"
    assert proj_crc == z["proj_crc"].tolist() and mem_crc == z["memory_crc"].tolist()
    meta = [str(f0["image"].dtype), str(f0["image"].shape), str(f0["proj_indices"].dtype), str(f0["proj_indices"].shape),
            str(f0["memory_features"].dtype), str(f0["memory_features"].shape), str(np.asarray(f0["gt_boxes"]).dtype),
            str(np.asarray(fe["gt_boxes"]).shape), str(np.asarray(fe["gt_boxes"]).dtype)]
    assert meta == z["frame_meta"].tolist()
    li = SMNetDetectionLoader(data_path=root, memory_type="", semmap_path="")
    assert crc(li[3][0]["memory_features"]) == int(z["memory_crc_image_only"][0])


def test_frame_dicts_match_the_reference_mapping(fx, root):
    """train_mp3d.py:452-507."""
    from embodied_object_detection_amd.data.mp3d import SMNetDetectionLoader, collate_smnet, map_mp3d_batch_to_coco
    z, _ = fx
    ld = SMNetDetectionLoader(data_path=root, test_type="default", memory_type="implicit_memory", semmap_path="")
    mapped = map_mp3d_batch_to_coco(collate_smnet([ld[0], ld[3]]))
    meta, img_crc, boxes, classes, nb = [], [], [], [], []
    for seq in mapped:
        for d in seq:
            assert set(d) == {"file_name", "sequence_name", "height", "width", "instances", "image", "memory", "proj_indices",
                              "memory_reset", "observations"}
            meta.append([d["height"], d["width"], int(d["memory_reset"]), *d["image"].shape])
            assert d["image"].dtype == torch.uint8
            img_crc.append(crc(d["image"].contiguous().numpy()))
            b = d["instances"].gt_boxes.tensor
            boxes += b.reshape(-1, 4).tolist()
            classes += d["instances"].gt_classes.reshape(-1).tolist()
            nb.append(int(b.shape[0]))
    assert np.array_equal(np.array(meta, dtype=np.int32), z["mapped_meta"])
    assert img_crc == z["mapped_image_crc"].tolist() and nb == z["mapped_n_boxes"].tolist()
    assert np.array_equal(np.array(boxes, dtype=np.float32).reshape(-1, 4), z["mapped_boxes"]) and classes == z["mapped_classes"].tolist()
    i0 = mapped[0][0]["instances"]
    assert [str(i0.gt_boxes.tensor.dtype), str(i0.gt_classes.dtype)] == z["mapped_box_dtype"].tolist()


class _StubModel:
    """What `tests/golden/gen_golden_io.py` used: records the calls, returns one tagged (empty) result per frame."""

    def __init__(self):
        self.calls = []

    def __call__(self, batched):
        from embodied_object_detection_amd.structures import Boxes, Instances
        assert len(batched) == 1
        seq = batched[0]
        self.calls.append({"sequence_name": seq[0]["sequence_name"], "n_frames": len(seq),
                           "memory_reset": [bool(f["memory_reset"]) for f in seq]})
        out = []
        for k, f in enumerate(seq):
            inst = Instances((f["height"], f["width"]))
            inst.pred_boxes = Boxes(torch.tensor([[0.0, 0.0, 1.0 + k, 1.0]]))          # one detection that names its frame
            inst.scores = torch.tensor([0.5])
            inst.pred_classes = torch.tensor([0])
            out.append({"instances": inst})
        return out


@pytest.mark.parametrize("test_type", ["default", "longterm"])
def test_eval_driver_matches_the_reference_driver(fx, root, test_type):
    """train_mp3d.py:85-363: episode order and reset flags reaching the model, every 5th frame of EACH episode reaching the
    evaluator (:187-188), GT rebuilt as integer-truncated XYWH (:232-238), quartiles by `idx % 100` of the dataloader index
    (:210-217) -- through `Mp3dScenes` + `inference_on_scenes` + `records_by_image`, single rank and sharded over 2 ranks."""
    from embodied_object_detection_amd.data.mp3d import Mp3dScenes, SMNetDetectionLoader
    from embodied_object_detection_amd.engine.eval_loop import (KIND_DET, KIND_GT, gather_records, inference_on_scenes, records_by_image,
                                                                 rows_needed)
    _, drv = fx
    ref = drv[test_type]
    ld = SMNetDetectionLoader(data_path=root, test_type=test_type, memory_type="implicit_memory", semmap_path="")
    ds = Mp3dScenes(ld)
    model = _StubModel()
    seen = []
    res = inference_on_scenes(model, ds.scenes, 0, max_rows=rows_needed([20] * len(ld)), scene_episode_offset=ds.episode_offsets(),
                              on_episode=lambda idx, inputs, outputs: seen.append((idx, inputs[0]["sequence_name"])))
    # the model sees the reference's episodes, in the reference's order, with the reference's reset flags
    assert model.calls == ref["model_calls"]
    assert [s[0] for s in seen] == list(range(len(ld))), "global episode index = the reference's dataloader index"
    buf = gather_records(res["records"], 0, 1, "cpu")
    dets, gts, quart = records_by_image(buf)
    uids = sorted(set(dets) | set(gts))
    # evaluated frames: the reference numbers them im_id = 0, 1, ... in dataloader order; ours sort the same way
    proc = [p for call in ref["processed"] for p in call]
    assert len(uids) == len(proc) == len(ref["images"])
    for uid, (im_id, file_name, seq_name, k) in zip(uids, proc):
        assert dets[uid]["boxes"][0, 2] == 1.0 + k, "not the every-5th frame the reference hands to the evaluator"
    # ground truth rows
    ref_gt = {}
    for (_id, im_id, cat, x, y, w, h, crowd, area) in ref["annotations"]:
        ref_gt.setdefault(im_id, []).append((cat, x, y, x + w, y + h))
    for n, uid in enumerate(uids):
        mine = [] if uid not in gts else [(int(c), *[int(v) for v in b]) for c, b in zip(gts[uid]["classes"], gts[uid]["boxes"])]
        assert mine == ref_gt.get(n, []), (n, uid)
    # quartiles
    for qi in range(4):
        assert [n for n, uid in enumerate(uids) if quart[uid] == qi] == ref["evaluate_calls"][qi], qi
    assert ref["evaluate_calls"][4] is None and len(ref["evaluate_calls"]) == 5
    # sharding scenes over two ranks changes nothing of the above
    merged = []
    for r in (0, 1):
        rr = inference_on_scenes(_StubModel(), ds.shard(r, 2), r, max_rows=rows_needed([20] * len(ld)),
                                 scene_episode_offset=ds.episode_offsets())
        merged.append(gather_records(rr["records"], 0, 1, "cpu")[0])
    d2, g2, q2 = records_by_image(np.stack(merged))
    assert sorted(set(d2) | set(g2)) == uids and q2 == quart
    assert all(np.array_equal(g2[u]["boxes"], gts[u]["boxes"]) for u in gts)


def test_do_train_matches_the_reference_loop(golden_dir, root):
    """`engine/train_loop.do_train` against the reference's OWN `do_train` (train_mp3d.py:509-659, run with a stub model / optimizer /
    checkpointer on this dataset by `gen_golden_io.py::gen_train_driver`): per iteration the episodes that reach `model(data)`
    (IMS_PER_BATCH of them, through collate_smnet + map_mp3d_batch_to_coco, in TrainingSampler's seeded order), the number the
    iteration is filed under, its learning rate; which saves happen under which name with which stored iteration and scheduler
    position; the periodic `do_test`; the writers' rhythm; a resumed run; SOLVER.TRAIN_ITER; MODEL.WEIGHTS with an iteration entry."""
    from embodied_object_detection_amd import setup_cfg
    from embodied_object_detection_amd.data.mp3d import SMNetDetectionLoader, collate_smnet, map_mp3d_batch_to_coco
    from embodied_object_detection_amd.engine import train_loop
    with open(os.path.join(golden_dir, "mp3d_train_driver.json")) as fh:
        ref = json.load(fh)
    loader = SMNetDetectionLoader(data_path=root, memory_type="implicit_memory", semmap_path="")

    class Model:
        def __init__(self):
            self.calls, self.training = [], False

        def train(self, mode=True):
            self.training = mode

        def eval(self):
            self.training = False

        def __call__(self, data):
            assert self.training
            self.calls.append([[ep[0]["sequence_name"], len(ep), [bool(f["memory_reset"]) for f in ep], type(ep[0]["instances"]).__name__]
                               for ep in data])
            n = len(self.calls)
            return {"loss_a": torch.tensor(2.0 / n), "loss_b": torch.tensor(1.0 / n)}

    class StubTrainer:
        def __init__(self):
            self.factors = []

        def optimizer_step(self, lr_factor=1.0):
            self.factors.append(lr_factor)

    for name, want in ref.items():
        if name == "sampler_seed":
            continue
        c = want["cfg"]
        cfg = setup_cfg(None, ["SOLVER.MAX_ITER", c["max_iter"], "SOLVER.TRAIN_ITER", c["train_iter"], "SOLVER.CHECKPOINT_PERIOD", c["period"],
                               "SOLVER.IMS_PER_BATCH", c["ims"], "SOLVER.BASE_LR", 0.01, "SOLVER.WARMUP_ITERS", 2, "SOLVER.WARMUP_FACTOR", 0.1,
                               "SOLVER.WARMUP_METHOD", "linear", "SOLVER.LR_SCHEDULER_NAME", "WarmupCosineLR", "TEST.EVAL_PERIOD", c["eval_period"]])
        model, tr = Model(), StubTrainer()
        saves, tests, logged = [], [], []
        sched = {"n": 0}
        resume_state = None
        if c["resume"] and want["found_iteration"] is not None:
            # what a run of `found_iteration` iterations leaves in its last checkpoint
            resume_state = {"iteration": want["found_iteration"], "optimizer": None, "scheduler": {"last_epoch": want["found_iteration"]}}
        rows = train_loop.do_train(cfg, model, tr, train_loop.training_batches(loader, c["ims"], seed=ref["sampler_seed"], collate=collate_smnet),
                                   resume_state=resume_state, map_batch=map_mp3d_batch_to_coco,
                                   on_save=lambda n, it: saves.append([n, it, (resume_state["scheduler"]["last_epoch"] if resume_state else 0) + len(tr.factors)]),
                                   do_test=lambda: tests.append(len(tr.factors)), log=lambda r: logged.append(len(tr.factors)))
        got_calls = [[[e[0], e[1], e[2]] for e in it] for it in model.calls]
        assert got_calls == [[[e[0], e[1], e[2]] for e in it] for it in want["model_calls"]], name
        assert all(e[3] == "Instances" for it in model.calls for e in it)
        assert [r["iteration"] for r in rows] == [r["storage_iter"] for r in want["rows"]], name
        for r, w in zip(rows, want["rows"]):
            assert abs(r["lr"] - w["lr"]) <= 1e-12 * max(abs(w["lr"]), 1e-12), (name, r["iteration"], r["lr"], w["lr"])
        assert [[n, it, ep] for n, it, ep in saves] == [[s[0], s[1]["iteration"], s[3]] for s in want["saves"]], (name, saves, want["saves"])
        assert all(s[2] == ["optimizer", "scheduler"] for s in want["saves"])
        assert tests == want["do_test_after_rows"], name
        assert logged == want["writer_after_rows"], name
        assert model.training == want["training"], name


def test_training_loader_worker_processes_give_the_same_batches(root):
    """`training_batches(..., workers=2)`: the reference's DataLoader arrangement (two forked worker processes, drop_last, collate_smnet,
    TrainingSampler; train_mp3d.py:552-572) yields the batches of the in-process iterator, in the same order."""
    from embodied_object_detection_amd.data.mp3d import SMNetDetectionLoader, collate_smnet
    from embodied_object_detection_amd.engine import train_loop
    loader = SMNetDetectionLoader(data_path=root, memory_type="implicit_memory", semmap_path="")
    a = train_loop.training_batches(loader, 2, seed=7, collate=collate_smnet)
    b = train_loop.training_batches(loader, 2, seed=7, collate=collate_smnet, workers=2)
    for _ in range(4):
        x, y = next(a), next(b)
        assert len(x) == len(y) == 2
        for ex, ey in zip(x, y):
            assert len(ex) == len(ey)
            for fx, fy in zip(ex, ey):
                assert fx["sequence_name"] == fy["sequence_name"] and fx["file_name"] == fy["file_name"]
                assert bool(fx["memory_reset"]) == bool(fy["memory_reset"])
                assert crc(np.asarray(fx["image"])) == crc(np.asarray(fy["image"]))
                assert crc(np.asarray(fx["proj_indices"])) == crc(np.asarray(fy["proj_indices"]))
    del b
