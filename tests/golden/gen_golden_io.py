#!/usr/bin/env python
"""Golden vectors of the on-disk loader and the eval driver  --  runs ONLY in the development container (needs /root/reference).

Executes the reference's OWN `SMNetDetectionLoader` / `collate_smnet` (`Detic/SMNet/loader.py:51-308`) and
`map_mp3d_batch_to_coco` / `mp3d_inference_on_dataset` (`Detic/train_mp3d.py:85-363,452-507`) on a small episode dataset written
by `tests/golden/_inputs.py::write_mp3d_mini` and stores what they produce:

* `mp3d_loader.npz`  -- file ordering (default / longterm), `memory_reset` of every frame for the three TEST_TYPEs, file names,
  filtered XYXY ground truth, CRCs of every decoded image / index image / memory table, dtypes and shapes, and the frame dicts
  `map_mp3d_batch_to_coco` builds from an episode;
* `mp3d_driver.json` -- with a stub model and a stub evaluator: which episodes reach `model([inputs])` in which order, which
  frames reach `evaluator.process` (every 5th), the COCO `images` / `annotations` the driver rebuilds (integer-truncated XYWH),
  the four quartile id lists and the order of the `evaluate` calls; for TEST_TYPE default and longterm.

Recipe as in gen_golden.py: absent packages are answered by stubs.  Here `h5py` is answered by a thin adapter
over the product's HDF5 binding (`data/h5io.py`, libhdf5 through ctypes -- h5py is not installed), PIL is the real Pillow, and the
detectron2-owned helpers the loader calls (`PathManager.open`, `_apply_exif_orientation`, `convert_PIL_to_numpy`) are restated
from their published semantics.  No reference source or bytecode is copied.

* `mp3d_train_driver.json` -- `do_train` (`Detic/train_mp3d.py:509-659`) with a stub model / optimizer / checkpointer: the episodes
  of every iteration, the iteration numbering, the lr of every iteration, the checkpoint rhythm (names, stored `iteration`), the
  periodic `do_test`, the writers' rhythm and the start of a resumed run.

    python tests/golden/gen_golden_io.py            # writes tests/golden/mp3d_loader.npz, mp3d_driver.json, mp3d_train_driver.json
"""
from __future__ import annotations

import io
import json
import os
import sys
import tempfile
import types
import zlib

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402   (the shim machinery)
import _inputs as I  # noqa: E402

from embodied_object_detection_amd.data import h5io  # noqa: E402

DETIC = G.DETIC
G.STUB_ROOTS = ("detectron2", "timm", "cv2", "fvcore", "mss", "tqdm", "torchvision", "pycocotools", "matplotlib", "tensorboardX")


def crc(a) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------
# h5py answered by data/h5io.py
# ------------------------------------------------------------------------------------------------
class _StrDataset:
    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


class _H5pyFile:
    def __init__(self, path, mode="r"):
        assert mode == "r"
        self.f = h5io.H5File(path)

    def __getitem__(self, name):
        try:
            return self.f.read(name)
        except h5io.H5Error:
            return _StrDataset(self.f.read_strings(name))

    def close(self):
        self.f.close()


def _install_h5py():
    m = types.ModuleType("h5py")
    m.File = _H5pyFile
    sys.modules["h5py"] = m


# detectron2-owned helpers of the loader (detectron2/data/detection_utils.py, utils/file_io.py), restated
def _apply_exif_orientation(image):
    from PIL import Image
    if not hasattr(image, "getexif"):
        return image
    try:
        exif = image.getexif()
    except Exception:
        exif = None
    if exif is None:
        return image
    method = {2: Image.FLIP_LEFT_RIGHT, 3: Image.ROTATE_180, 4: Image.FLIP_TOP_BOTTOM, 5: Image.TRANSPOSE, 6: Image.ROTATE_270,
              7: Image.TRANSVERSE, 8: Image.ROTATE_90}.get(exif.get(0x0112))
    return image.transpose(method) if method is not None else image


def _convert_PIL_to_numpy(image, format):
    assert format == "RGB"
    return np.asarray(image.convert("RGB"))


class _PathManager:
    @staticmethod
    def open(path, mode="r"):
        return open(path, mode)


def load_reference_io_modules():
    G._SPECIAL.update({"_apply_exif_orientation": _apply_exif_orientation, "convert_PIL_to_numpy": _convert_PIL_to_numpy,
                       "PathManager": _PathManager, "get_world_size": lambda: 1})
    G.install_shim()
    _install_h5py()
    G._pkg("SMNet", os.path.join(DETIC, "SMNet"))
    loader = G._load("SMNet.loader", os.path.join(DETIC, "SMNet", "loader.py"))
    for name in ("centernet", "centernet.config", "detic", "detic.config", "detic.data", "detic.data.custom_build_augmentation",
                 "detic.data.custom_dataset_dataloader", "detic.data.custom_dataset_mapper", "detic.custom_solver", "detic.evaluation",
                 "detic.evaluation.oideval", "detic.evaluation.custom_coco_eval", "detic.modeling", "detic.modeling.utils"):
        sys.modules[name] = G._StubModule(name)
    driver = G._load("eod_ref_train_mp3d", os.path.join(DETIC, "train_mp3d.py"))
    return loader, driver


# ------------------------------------------------------------------------------------------------
# cases
# ------------------------------------------------------------------------------------------------
def encode_jpegs():
    from PIL import Image
    out = []
    for px in I.mp3d_mini_jpeg_pixels():
        buf = io.BytesIO()
        Image.fromarray(px).save(buf, format="JPEG", quality=90)
        out.append(np.frombuffer(buf.getvalue(), dtype=np.uint8).copy())
    return out


def gen_loader(loader_mod, root, jpegs, out):
    L = loader_mod.SMNetDetectionLoader
    res = {f"jpeg_{i}": b for i, b in enumerate(jpegs)}
    for tt in ("default", "episodic", "longterm"):
        ld = L(data_path=root, test_type=tt, memory_type="implicit_memory", semmap_path="")
        res[f"files_{tt}"] = np.array(ld.files)
        resets, lengths = [], []
        for i in range(len(ld)):
            if tt != "default" and i >= 12 and not (tt == "longterm" and 45 <= i < 56):
                continue                       # the reset rule needs the names only: a prefix (and the longterm seam) is enough
            ep = ld[i]
            lengths.append(len(ep))
            resets += [bool(f["memory_reset"]) for f in ep]
        res[f"resets_{tt}"] = np.array(resets)
        res[f"lengths_{tt}"] = np.array(lengths, dtype=np.int32)
    ld = L(data_path=root, test_type="default", memory_type="implicit_memory", semmap_path="")
    names, boxes, classes, nbox, img_crc, proj_crc, mem_crc = [], [], [], [], [], [], []
    for i in range(len(ld)):
        ep = loader_mod.collate_smnet([ld[i]])[0]
        mem_crc.append(crc(ep[0]["memory_features"]))
        for f in ep:
            assert set(f) == {"file_name", "sequence_name", "gt_boxes", "gt_classes", "image", "proj_indices", "memory_reset",
                              "memory_features", "observations"}, sorted(f)
            assert f["observations"] is None and f["sequence_name"] == ld.files[i]
            names.append(f["file_name"])
            nbox.append(len(f["gt_classes"]))
            boxes += np.asarray(f["gt_boxes"], dtype=np.float64).reshape(-1, 4).tolist()
            classes += np.asarray(f["gt_classes"]).reshape(-1).tolist()
            img_crc.append(crc(f["image"]))
            proj_crc.append(crc(f["proj_indices"]))
    f0, fe = ld[0][0], None
    for i in range(len(ld)):
        for f in ld[i]:
            if len(f["gt_classes"]) == 0:
                fe = f
                break
        if fe is not None:
            break
    res.update(file_names=np.array(names), gt_boxes=np.array(boxes, dtype=np.float64).reshape(-1, 4),
               gt_classes=np.array(classes, dtype=np.int64), n_boxes=np.array(nbox, dtype=np.int32),
               image_crc=np.array(img_crc, dtype=np.uint32), proj_crc=np.array(proj_crc, dtype=np.uint32),
               memory_crc=np.array(mem_crc, dtype=np.uint32),
               frame_meta=np.array([str(f0["image"].dtype), str(f0["image"].shape), str(f0["proj_indices"].dtype),
                                    str(f0["proj_indices"].shape), str(f0["memory_features"].dtype), str(f0["memory_features"].shape),
                                    str(np.asarray(f0["gt_boxes"]).dtype), str(np.asarray(fe["gt_boxes"]).shape),
                                    str(np.asarray(fe["gt_boxes"]).dtype)]))
    # MEMORY_TYPE image_only / '' hands the offline map features over (loader.py:300-302)
    li = L(data_path=root, memory_type="", semmap_path="")
    res["memory_crc_image_only"] = np.array([crc(li[3][0]["memory_features"])], dtype=np.uint32)
    return res, ld


def gen_frame_dicts(driver_mod, loader_mod, ld, res):
    """map_mp3d_batch_to_coco on two episodes (train_mp3d.py:452-507)."""
    eps = [ld[0], ld[3]]
    mapped = driver_mod.map_mp3d_batch_to_coco(loader_mod.collate_smnet(eps))
    meta, img_crc, boxes, classes, nb = [], [], [], [], []
    for seq in mapped:
        for d in seq:
            assert set(d) == {"file_name", "sequence_name", "height", "width", "instances", "image", "memory", "proj_indices",
                              "memory_reset", "observations"}, sorted(d)
            meta.append([d["height"], d["width"], int(d["memory_reset"]), *d["image"].shape])
            assert d["image"].dtype == torch.uint8
            img_crc.append(crc(d["image"].contiguous().numpy()))
            b = d["instances"].gt_boxes.tensor
            boxes += b.reshape(-1, 4).tolist()
            classes += d["instances"].gt_classes.reshape(-1).tolist()
            nb.append(int(b.shape[0]))
    res.update(mapped_meta=np.array(meta, dtype=np.int32), mapped_image_crc=np.array(img_crc, dtype=np.uint32),
               mapped_boxes=np.array(boxes, dtype=np.float32).reshape(-1, 4), mapped_classes=np.array(classes, dtype=np.int64),
               mapped_n_boxes=np.array(nb, dtype=np.int32),
               mapped_box_dtype=np.array([str(mapped[0][0]["instances"].gt_boxes.tensor.dtype),
                                          str(mapped[0][0]["instances"].gt_classes.dtype)]))


class _StubModel:
    """Records what reaches `model([inputs])`; returns one tagged output per frame."""

    def __init__(self):
        self.calls = []

    def __call__(self, batched):
        assert len(batched) == 1
        seq = batched[0]
        self.calls.append({"sequence_name": seq[0]["sequence_name"], "n_frames": len(seq),
                           "memory_reset": [bool(f["memory_reset"]) for f in seq]})
        return [{"instances": (seq[0]["sequence_name"], k)} for k in range(len(seq))]


class _StubEvaluator:
    def __init__(self):
        self._metadata = types.SimpleNamespace()
        cats = [{"id": i, "name": f"c{i}"} for i in range(20)]
        self._coco_api = types.SimpleNamespace(dataset={"categories": cats}, createIndex=self._index)
        self.processed, self.evaluate_calls, self.indexed = [], [], 0

    def _index(self):
        self.indexed += 1

    def reset(self):
        self.processed = []

    def process(self, inputs, outputs):
        assert len(inputs) == len(outputs)
        self.processed.append([[int(i["image_id"]), i["file_name"], o["instances"][0], int(o["instances"][1])]
                               for i, o in zip(inputs, outputs)])

    def evaluate(self, img_ids=None):
        self.evaluate_calls.append(None if img_ids is None else [int(i) for i in img_ids])
        return {}


def gen_driver(driver_mod, loader_mod, root):
    from torch.utils.data import DataLoader
    out = {}
    for tt in ("default", "longterm"):
        ld = loader_mod.SMNetDetectionLoader(data_path=root, test_type=tt, memory_type="implicit_memory", semmap_path="")
        # do_test's loader (train_mp3d.py:399-410) in one process: batch 1, dataset order, collate_smnet
        dl = DataLoader(ld, batch_size=1, shuffle=False, num_workers=0, collate_fn=loader_mod.collate_smnet)
        model, ev = _StubModel(), _StubEvaluator()
        driver_mod.mp3d_inference_on_dataset(model, dl, ev)
        annos = ev._coco_api.dataset
        out[tt] = {
            "model_calls": model.calls,
            "processed": ev.processed,
            "images": annos["images"],
            "annotations": [[a["id"], a["image_id"], a["category_id"], *a["bbox"], a["iscrowd"], a["area"]] for a in annos["annotations"]],
            "evaluate_calls": ev.evaluate_calls,
            "thing_classes": list(ev._metadata.thing_classes),
        }
    return out


# ------------------------------------------------------------------------------------------------
# do_train (train_mp3d.py:509-659) with a stub model: the loop's own control flow
# ------------------------------------------------------------------------------------------------
# detectron2 / fvcore pieces the loop calls, restated from their published semantics (not in the reference tree: "unpinned"):
class _TrainingSampler:
    """detectron2 `TrainingSampler(size, shuffle=True, seed)`: an infinite stream of seeded permutations.  The reference passes no seed
    (`shared_random_seed()`: the order differs from run to run); the fixture fixes it so that the stream can be compared."""
    SEED = 20

    def __init__(self, size, shuffle=True, seed=None):
        self.size, self.shuffle, self.seed = size, shuffle, _TrainingSampler.SEED if seed is None else int(seed)

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed)
        while True:
            yield from (torch.randperm(self.size, generator=g).tolist() if self.shuffle else list(range(self.size)))


def _warmup_cosine_scheduler(cfg, optimizer):
    """detectron2 `build_lr_scheduler` for WarmupCosineLR: LRMultiplier(WarmupParamScheduler(CosineParamScheduler(1, 0), warmup_factor,
    min(warmup_iters / max_iter, 1), method), max_iter): the multiplier of iteration i is evaluated at where = i / max_iter."""
    import math
    s = cfg.SOLVER
    max_iter, wi, wf = int(s.MAX_ITER), int(s.WARMUP_ITERS), float(s.WARMUP_FACTOR)
    wlen = min(wi / max_iter, 1.0)

    def mult(i):
        where = i / max_iter if i < max_iter else 1.0 - 1e-12            # LRMultiplier clamps the last step
        cos = 0.5 * (1.0 + math.cos(math.pi * where))
        if where >= wlen:
            return cos
        end = 0.5 * (1.0 + math.cos(math.pi * wlen))                     # the warmup ends on the cosine's value there
        start = wf * 1.0                                                   # warmup_factor x the schedule's value at 0
        a = where / wlen
        return end * a + start * (1 - a) if str(s.WARMUP_METHOD) == "linear" else start
    return torch.optim.lr_scheduler.LambdaLR(optimizer, mult)


class _Checkpointer:
    """fvcore `Checkpointer` as `DetectionCheckpointer(model, dir, optimizer=, scheduler=)`: `save(name, **extra)` stores the model,
    every checkpointable's `state_dict()` and the extra state (recorded: name, extra, the checkpointables' names, the scheduler's
    `last_epoch`); `resume_or_load(path, resume=True)` loads the last checkpoint -- `load_state_dict` on every checkpointable -- and
    returns its extra state; with `resume=False` it loads `path` as weights only and returns what that file holds besides them."""
    found = {}          # the last checkpoint of the output directory: {"iteration": i, "optimizer": sd, "scheduler": sd} or {}
    saves = None

    def __init__(self, model, save_dir="", **checkpointables):
        self.objs = dict(checkpointables)

    def resume_or_load(self, path, *, resume=True):
        f = _Checkpointer.found
        if resume and f:
            for k, obj in self.objs.items():
                obj.load_state_dict(f[k])
            return {"iteration": f["iteration"]}
        return {"iteration": f["iteration"]} if f else {}       # MODEL.WEIGHTS with an 'iteration' entry: do_train must ignore it (:526-527)

    def save(self, name, **extra):
        _Checkpointer.saves.append([name + ".pth", {k: int(v) for k, v in extra.items()}, sorted(self.objs),
                                    int(self.objs["scheduler"].last_epoch)])


class _PeriodicCheckpointer:
    """fvcore `PeriodicCheckpointer.step`: save `model_{iteration:07d}` when (iteration + 1) % period == 0, `model_final` when
    iteration >= max_iter - 1; both with `iteration=` the number it was handed."""

    def __init__(self, checkpointer, period, max_iter=None, max_to_keep=None, file_prefix="model"):
        self.c, self.period, self.max_iter, self.prefix = checkpointer, int(period), max_iter, file_prefix

    def step(self, iteration, **kw):
        iteration = int(iteration)
        if (iteration + 1) % self.period == 0:
            self.c.save("{}_{:07d}".format(self.prefix, iteration), iteration=iteration, **kw)
        if self.max_iter is not None and iteration >= self.max_iter - 1:
            self.c.save(f"{self.prefix}_final", iteration=iteration, **kw)


class _EventStorage:
    """detectron2 `EventStorage(start_iter)`: `iter`, `step()`, `put_scalar(s)`; records the `lr` scalar with the iteration it is filed under."""
    rows = None

    def __init__(self, start_iter=0):
        self.iter = start_iter

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def step(self):
        self.iter += 1

    def put_scalars(self, **kw):
        if "total_loss" in kw:
            _EventStorage.rows.append({"storage_iter": self.iter, "total_loss": float(kw["total_loss"])})

    def put_scalar(self, name, value, smoothing_hint=True):
        if name == "lr":
            _EventStorage.rows[-1]["lr"] = float(value)


class _Writer:
    writes = None

    def __init__(self, *a, **k):
        pass

    def write(self):
        _Writer.writes.append(len(_EventStorage.rows))


class _StubTrainModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(()))
        self.calls = []

    def forward(self, data):
        assert self.training
        self.calls.append([[ep[0]["sequence_name"], len(ep), [bool(f["memory_reset"]) for f in ep], type(ep[0]["instances"]).__name__]
                           for ep in data])
        n = len(self.calls)
        return {"loss_a": self.w * (2.0 / n), "loss_b": self.w * self.w * (1.0 / n)}


def _ns(d):
    return types.SimpleNamespace(**{k: _ns(v) if isinstance(v, dict) else v for k, v in d.items()})


def gen_train_driver(driver_mod, loader_mod, root):
    """`do_train` (train_mp3d.py:509-659) on the written dataset, cfg.DATALOADER.SAMPLER_TRAIN 'MP3DLoader': which episodes reach
    `model(data)` per iteration (IMS_PER_BATCH of them through collate_smnet + map_mp3d_batch_to_coco), under which iteration number
    the losses / lr are filed, the lr of every iteration (optimizer.step before scheduler.step), which iterations save which
    checkpoint under which name with which `iteration`, when `do_test` runs (TEST.EVAL_PERIOD), when the writers write, and where a
    resumed run starts.  The DataLoader is the reference's own call with its worker processes switched off (num_workers 0)."""
    from torch.utils.data import DataLoader
    comm = sys.modules["detectron2.utils.comm"]
    comm.reduce_dict = lambda d: d
    comm.is_main_process = lambda: True
    comm.synchronize = lambda: None
    driver_mod.TrainingSampler = _TrainingSampler
    driver_mod.build_lr_scheduler = _warmup_cosine_scheduler
    driver_mod.build_custom_optimizer = lambda cfg, model: torch.optim.SGD(model.parameters(), lr=float(cfg.SOLVER.BASE_LR))
    driver_mod.DetectionCheckpointer = _Checkpointer
    driver_mod.PeriodicCheckpointer = _PeriodicCheckpointer
    driver_mod.EventStorage = _EventStorage
    driver_mod.CommonMetricPrinter = _Writer
    driver_mod.JSONWriter = lambda *a, **k: types.SimpleNamespace(write=lambda: None)
    driver_mod.TensorboardXWriter = lambda *a, **k: types.SimpleNamespace(write=lambda: None)
    driver_mod.DataLoader = lambda ds, **k: DataLoader(ds, **{**k, "num_workers": 0, "pin_memory": False, "multiprocessing_context": None})
    tests = []
    driver_mod.do_test = lambda cfg, model: tests.append(len(_EventStorage.rows))
    out = {"sampler_seed": _TrainingSampler.SEED}
    cases = {
        "plain": dict(max_iter=7, train_iter=-1, period=3, eval_period=0, ims=2, resume=False, found={}),
        "period2_eval2": dict(max_iter=6, train_iter=-1, period=2, eval_period=2, ims=1, resume=False, found={}),
        "train_iter_cap": dict(max_iter=9, train_iter=4, period=5, eval_period=0, ims=1, resume=False, found={}),
        "resumed": dict(max_iter=7, train_iter=-1, period=3, eval_period=0, ims=2, resume=True, found={"iteration": 3}),
        "weights_not_resumed": dict(max_iter=3, train_iter=-1, period=2, eval_period=0, ims=1, resume=False, found={"iteration": 3}),
        "logged": dict(max_iter=27, train_iter=-1, period=100, eval_period=0, ims=1, resume=False, found={}),
    }
    for name, c in cases.items():
        cfg = _ns({"SOLVER": {"USE_CUSTOM_SOLVER": True, "OPTIMIZER": "ADAMW", "MAX_ITER": c["max_iter"], "TRAIN_ITER": c["train_iter"],
                              "CHECKPOINT_PERIOD": c["period"], "IMS_PER_BATCH": c["ims"], "BASE_LR": 0.01, "WARMUP_ITERS": 2,
                              "WARMUP_FACTOR": 0.1, "WARMUP_METHOD": "linear", "CLIP_GRADIENTS": {"CLIP_TYPE": "value"},
                              "BACKBONE_MULTIPLIER": 1.0},
                   "OUTPUT_DIR": "/nonexistent", "WITH_IMAGE_LABELS": False, "INPUT": {"CUSTOM_AUG": ""},
                   "DATALOADER": {"SAMPLER_TRAIN": "MP3DLoader"}, "FP16": False, "TEST": {"EVAL_PERIOD": c["eval_period"]},
                   "MODEL": {"WEIGHTS": "", "MEMORY_TYPE": "implicit_memory", "SEMMAP_PATH": "", "TRAIN_DATA_PATH": root,
                             "ROI_BOX_HEAD": {"ZEROSHOT_WEIGHT_PATH": ""}}})
        found = {}
        if c["found"]:
            # the state a run of `it` iterations of this configuration leaves behind: optimizer and scheduler stepped `it` times
            it = c["found"]["iteration"]
            m0 = _StubTrainModel()
            o0 = driver_mod.build_custom_optimizer(cfg, m0)
            s0 = _warmup_cosine_scheduler(cfg, o0)
            for _ in range(it):
                o0.step()
                s0.step()
            found = {"iteration": it, "optimizer": o0.state_dict(), "scheduler": s0.state_dict()}
        _Checkpointer.found, _Checkpointer.saves = found, []
        _EventStorage.rows, _Writer.writes = [], []
        del tests[:]
        model = _StubTrainModel()
        driver_mod.do_train(cfg, model, resume=c["resume"])
        out[name] = {"cfg": {k: v for k, v in c.items() if k != "found"}, "found_iteration": c["found"].get("iteration"),
                     "model_calls": model.calls, "rows": _EventStorage.rows, "saves": list(_Checkpointer.saves),
                     "do_test_after_rows": list(tests), "writer_after_rows": list(_Writer.writes), "training": bool(model.training)}
    return out


def main():
    out = HERE
    loader_mod, driver_mod = load_reference_io_modules()
    jpegs = encode_jpegs()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        root = os.path.join(tmp, "mp3d_mini")
        I.write_mp3d_mini(root, jpegs)
        os.chdir(DETIC)             # the loader opens 'SMNet/semmap_GT_info.json' relative to the Detic directory (loader.py:81,124)
        try:
            res, ld = gen_loader(loader_mod, root, jpegs, out)
            gen_frame_dicts(driver_mod, loader_mod, ld, res)
            drv = gen_driver(driver_mod, loader_mod, root)
            trn = gen_train_driver(driver_mod, loader_mod, root)
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(out, "mp3d_loader.npz"), **res)
    with open(os.path.join(out, "mp3d_driver.json"), "w") as fh:
        json.dump(drv, fh, separators=(",", ":"), sort_keys=True)
        fh.write("\n")
    with open(os.path.join(out, "mp3d_train_driver.json"), "w") as fh:
        json.dump(trn, fh, separators=(",", ":"), sort_keys=True)
        fh.write("\n")
    for f in ("mp3d_loader.npz", "mp3d_driver.json", "mp3d_train_driver.json"):
        print(f, os.path.getsize(os.path.join(out, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
