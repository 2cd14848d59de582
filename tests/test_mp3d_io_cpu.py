"""On-disk formats: HDF5 container through libhdf5, the episode loader mirror and the frame-dict mapping.
The episode files are written here in the layout `build_data.py:276-286` / `build_memory_data.py:151-153` produce."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from embodied_object_detection_amd.data import h5io

pytestmark = pytest.mark.skipif(not h5io.available(), reason="no libhdf5 in this image")


def test_h5_roundtrip_and_h5dump_sees_h5py_compatible_types(tmp_path):
    a = np.arange(24, dtype=np.int32).reshape(2, 3, 4)
    b = np.random.RandomState(0).rand(5, 7).astype(np.float32)
    path = str(tmp_path / "x.h5")
    with h5io.H5File(path, "w") as f:
        f.write("proj_indices", a)
        f.write("memory_features", b.astype(np.float64), dtype=np.float32)
        f.write("masks_outliers", np.array([True, False, True]))
        f.write("empty", np.zeros((0, 4), dtype=np.float32))
        f.write_strings("detection_data", ["{'a': 1}", "second ü", ""])
    with h5io.H5File(path) as f:
        assert sorted(f.keys()) == ["detection_data", "empty", "masks_outliers", "memory_features", "proj_indices"]
        assert "proj_indices" in f and "nope" not in f and f.shape("proj_indices") == (2, 3, 4) and f.shape("detection_data") == (3,)
        got = f.read("proj_indices")
        assert got.dtype == np.int32 and np.array_equal(got, a)
        assert f.read("memory_features").dtype == np.float32 and np.array_equal(f.read("memory_features"), b)
        assert f.read("masks_outliers").tolist() == [1, 0, 1] and f.read("empty").shape == (0, 4)
        assert f.read_strings("detection_data") == [b"{'a': 1}", "second ü".encode(), b""]
        with pytest.raises(KeyError):
            f.read("nope")
        with pytest.raises(h5io.H5Error):
            f.read("detection_data")
    with pytest.raises(h5io.H5Error):
        h5io.H5File(str(tmp_path / "missing.h5"))
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):      # an independent HDF5 tool must agree on names and types (h5py's vlen str = variable UTF-8 C string)
        out = subprocess.run([h5dump, "-H", path], capture_output=True, text=True).stdout
        assert 'DATASET "proj_indices"' in out and "H5T_STD_I32LE" in out and "H5T_IEEE_F32LE" in out
        assert "STRSIZE H5T_VARIABLE" in out and "H5T_CSET_UTF8" in out and "( 2, 3, 4 )" in out


def _write_dataset(root, H=32, W=64, T=3, n_cells=50):
    """2 scenes ('sA_1' with episodes 0,1,10 and 'sB_2' with episode 0); returns what was written."""
    from PIL import Image
    rng = np.random.RandomState(1)
    for d in ("memory_data", "sensor_data", "JPEGImages"):
        os.makedirs(os.path.join(root, d))
    written = {}
    for name in ("sA_1_10.h5", "sA_1_0.h5", "sB_2_0.h5", "sA_1_1.h5"):
        proj = rng.randint(0, n_cells, size=(T, H, W, 1)).astype(np.int32)
        mem = rng.rand(n_cells, 256).astype(np.float32)
        sem = rng.randint(0, 13, size=(n_cells,)).astype(np.int32)
        with h5io.H5File(os.path.join(root, "memory_data", name), "w") as f:
            f.write("memory_features", mem)
            f.write("proj_indices", proj)
            f.write("semmap_gt", sem)
        recs, imgs = [], []
        for i in range(T):
            img = rng.randint(0, 255, size=(H, W, 3)).astype(np.uint8)
            fn = f"{name[:-3]}_{i}.jpg"
            Image.fromarray(img).save(os.path.join(root, "JPEGImages", fn), quality=95)
            imgs.append(fn)
            # str(dict) exactly as build_data.py:245 writes it: XYWH boxes, classes incl. ids outside the evaluated subset (1, 8)
            recs.append(str({"file_name": fn, "image": "x", "gt_boxes": [[1, 2, 10, 20], [5, 6, 7, 8], [0, 0, 3, 3]],
                             "gt_classes": [0, 1, 19]}))
        with h5io.H5File(os.path.join(root, "sensor_data", name), "w") as f:
            f.write("rgb", np.zeros((T, H, W, 3), dtype=np.uint8))
            f.write("segmentation_data", rng.randint(0, 21, size=(T, H, W)).astype(np.uint8))
            f.write_strings("detection_data", recs)
        written[name] = dict(proj=proj, mem=mem, sem=sem, imgs=imgs)
    return written


def test_loader_mirrors_reference_semantics(tmp_path):
    from PIL import Image
    from embodied_object_detection_amd.data.mp3d import (Mp3dScenes, SMNetDetectionLoader, collate_smnet, episode_sort_key,
                                                            longterm_file_list, map_mp3d_batch_to_coco)
    root = str(tmp_path / "ds")
    w = _write_dataset(root)
    ld = SMNetDetectionLoader(data_path=root, test_type="default", memory_type="implicit_memory", semmap_path="")
    # loader.py:97-105: numeric order of the last token, not lexicographic ("10" after "1")
    assert ld.files == ["sA_1_0.h5", "sA_1_1.h5", "sA_1_10.h5", "sB_2_0.h5"] and len(ld) == 4
    ep = ld[2]
    assert len(ep) == 3 and set(ep[0]) == {"file_name", "sequence_name", "gt_boxes", "gt_classes", "image", "proj_indices",
                                           "memory_reset", "memory_features", "observations"}
    r = ep[1]
    assert r["sequence_name"] == "sA_1_10.h5" and r["file_name"] == "sA_1_10_1.jpg"
    # class 1 is not in the evaluated subset (loader.py:133,256-257); XYWH -> XYXY (:253)
    assert r["gt_classes"].tolist() == [0, 19] and r["gt_boxes"].tolist() == [[1, 2, 11, 22], [0, 0, 3, 3]]
    assert np.array_equal(r["proj_indices"], w["sA_1_10.h5"]["proj"][1]) and r["proj_indices"].shape == (32, 64, 1)
    assert np.array_equal(r["memory_features"], w["sA_1_10.h5"]["mem"]) and r["observations"] is None
    assert np.array_equal(r["image"], np.asarray(Image.open(os.path.join(root, "JPEGImages", "sA_1_10_1.jpg")).convert("RGB")))
    # memory_reset: default = first frame of episode 0 of a scene only; episodic = first frame of every episode (:289-293)
    assert [f["memory_reset"] for f in ld[0]] == [True, False, False] and not any(f["memory_reset"] for f in ld[1] + ld[2])
    assert [f["memory_reset"] for f in ld[3]] == [True, False, False]
    le = SMNetDetectionLoader(data_path=root, test_type="episodic", memory_type="implicit_memory", semmap_path="")
    assert [f["memory_reset"] for f in le[1]] == [True, False, False]
    # image_only: memory is the offline map features (loader.py:300-302)
    li = SMNetDetectionLoader(data_path=root, memory_type="", semmap_path="")
    assert np.array_equal(li[0][0]["memory_features"], w["sA_1_0.h5"]["mem"])
    # a broken memory file degrades to the reference's fallback shapes (:205-208)
    open(os.path.join(root, "memory_data", "sB_2_0.h5"), "wb").write(b"not hdf5")
    assert ld[3][0]["memory_features"].shape == (1, 256) and ld[3][0]["proj_indices"].shape == (480, 640, 1)
    # longterm duplication (:108-117)
    files = [f"s_0_{i}.h5" for i in range(120)]
    lt = longterm_file_list(sorted(files, key=episode_sort_key))
    assert len(lt) == 240 and lt[:50] == lt[50 - 0:100][:0] + lt[:50] and lt[50] == lt[49] and lt[51:100] == lt[1:50]
    assert lt[100:150] == lt[150 + 0:200][:0] + lt[100:150] and lt[150] == lt[149]
    # frame-dict mapping (train_mp3d.py:452-507)
    frames = map_mp3d_batch_to_coco(collate_smnet([ld[0]]))[0]
    f0 = frames[0]
    assert f0["height"] == 32 and f0["width"] == 64 and tuple(f0["image"].shape) == (3, 32, 64) and f0["image"].dtype == torch.uint8
    assert f0["instances"].gt_boxes.tensor.tolist() == [[1, 2, 11, 22], [0, 0, 3, 3]] and f0["instances"].gt_classes.tolist() == [0, 19]
    assert f0["memory"].shape == (50, 256) and f0["proj_indices"].dtype == np.int32 and f0["memory_reset"] is True
    # scenes for the sharded eval driver: episodes of a scene stay together, global episode offsets follow the file order
    ds = Mp3dScenes(ld)
    assert [s.name for s in ds.scenes] == ["sA_1_", "sB_2_"] and [s.indices for s in ds.scenes] == [[0, 1, 2], [3]]
    assert ds.episode_offsets() == {0: 0, 1: 3} and [s.seq_id for s in ds.shard(1, 2)] == [1]
    assert len(list(ds.scenes[0].episodes())) == 3


def test_snapshot_is_hdf5_with_the_reference_names(tmp_path):
    from embodied_object_detection_amd.data import snapshot as S
    sem = np.arange(-1, 9, dtype=np.int64)
    path = S.write_snapshot(str(tmp_path), "scene0_1.h5", sem, np.ones((10, 512)), np.arange(10.0))
    assert path.endswith(os.path.join("memory", "scene0_1.h5"))
    with h5io.H5File(path) as f:
        assert sorted(f.keys()) == ["impicit_memory", "observations", "semmap"] and f.read("semmap").dtype == np.int32
    got = S.read_snapshot(os.path.join(str(tmp_path), "memory"), "scene0_1.h5")
    assert np.array_equal(got["semmap_real"], sem + 1) and got["implicit_memory"].dtype == np.float32


def test_robot_run_reads_a_recorded_run(tmp_path):
    """robot_demo.py:485-517 on disk: every second image, nearest depth / pose by timestamp, 16-bit depth in millimetres."""
    from PIL import Image
    from embodied_object_detection_amd.data import robot as R
    from oracle import projector as OP
    root = tmp_path / "run7"
    for d in ("images", "depth", "pose"):
        os.makedirs(root / d)
    rng = np.random.RandomState(3)
    H, W = 64, 96
    stamps = [1000, 1050, 1100, 1150, 1200]
    for t in stamps:
        Image.fromarray(rng.randint(0, 255, size=(H, W, 3)).astype(np.uint8)).save(root / "images" / f"{t}.png")
    depths = {}
    for t in (990, 1110, 1190):
        depths[t] = rng.randint(500, 9000, size=(H, W)).astype(np.uint16)
        Image.fromarray(depths[t]).save(root / "depth" / f"{t}.png")           # 16-bit single-channel PNG
    for t, pose in ((1001, [1.0, 2.0, 0.1]), (1125, [1.5, 2.5, 0.2]), (1210, [2.0, 3.0, 0.3])):
        np.save(root / "pose" / f"{t}.npy", np.array(pose, dtype=np.float32))
    fe = R.RobotFrontEnd(projector=lambda d, T_, intr, ps, ms, cell, mw, mh, order=0: OP.depth_to_proj_indices(d, T_, intr, ps, ms, cell, mw, mh, order),
                         sequence_name="run7")
    run = R.RobotRun(str(root), fe)
    frames = list(run)
    assert len(run) == 3 and [f["file_name"] for f in frames] == ["1000.png", "1100.png", "1200.png"]
    assert [f["depth_file"] for f in frames] == ["990.png", "1110.png", "1190.png"]
    assert [f["pose_file"] for f in frames] == ["1001.npy", "1125.npy", "1210.npy"]
    assert [f["memory_reset"] for f in frames] == [True, False, False] and frames[0]["sequence_name"] == "run7"
    assert np.array_equal(R.read_depth_mm(str(root / "depth" / "1110.png")), depths[1110])
    f1 = frames[1]
    assert tuple(f1["image"].shape) == (3, H, W) and f1["proj_indices"].shape == (H, W, 1) and f1["proj_indices"].dtype == np.int32
    depth_m = (depths[1110].astype(np.float64) / 1000).astype(np.float32)
    ref = OP.depth_to_proj_indices(depth_m, R.robot_transform([1.5, 2.5, 0.2]), R.ROBOT_INTRINSICS, (0.0, 0.0, 0.0),
                                   np.asarray(R.ROBOT_MAP_SHIFT, dtype=np.float32), R.ROBOT_RES, R.ROBOT_MAP_W, R.ROBOT_MAP_H, 1)
    assert np.array_equal(f1["proj_indices"][..., 0], ref)


def test_loader_semantic_gt_map_gt_and_realtime_snapshot(tmp_path):
    """The baseline memory types of the loader (loader.py:229-243,262-269) and the real-time snapshot path (:214-223)."""
    from embodied_object_detection_amd.data import snapshot as S
    from embodied_object_detection_amd.data.mp3d import SMNET_CLASS_MAPPING, SMNetDetectionLoader
    root = str(tmp_path / "ds")
    w = _write_dataset(root)
    clip = np.random.RandomState(5).rand(20, 512).astype(np.float32)
    clip_path = str(tmp_path / "clip.npy")
    np.save(clip_path, clip)
    name = "sA_1_0.h5"
    # semantic_gt: memory = [zero row; clip rows], proj_indices = the segmentation image of the frame
    ld = SMNetDetectionLoader(data_path=root, clip_path=clip_path, memory_type="semantic_gt", semmap_path="")
    ep = ld[0]
    with h5io.H5File(os.path.join(root, "sensor_data", name)) as f:
        seg = f.read("segmentation_data")
    assert ep[0]["memory_features"].shape == (21, 512) and not ep[0]["memory_features"][0].any()
    assert np.array_equal(ep[0]["memory_features"][1:], clip)
    assert np.array_equal(ep[1]["proj_indices"][..., 0], seg[1]) and ep[1]["observations"] is None
    # map_gt without a snapshot: class-mapped clip rows, indices looked up in the GT semantic map
    lm = SMNetDetectionLoader(data_path=root, clip_path=clip_path, memory_type="map_gt", semmap_path="")
    em = lm[0]
    full = np.insert(clip, 0, np.zeros((1, 512)), axis=0)
    assert np.array_equal(em[0]["memory_features"], full[SMNET_CLASS_MAPPING])
    assert np.array_equal(em[2]["proj_indices"], w[name]["sem"][w[name]["proj"]][2])
    # implicit_memory with a real-time snapshot directory: memory / observations come from the dump, labels are shifted by +1
    snap_dir = str(tmp_path / "out")
    mem = np.random.RandomState(6).rand(50, 512).astype(np.float32)
    obs = np.arange(50, dtype=np.float32)
    S.write_snapshot(snap_dir, name, np.full((50,), -1, dtype=np.int32), mem, obs)
    ls = SMNetDetectionLoader(data_path=root, memory_type="implicit_memory", semmap_path=os.path.join(snap_dir, "memory"))
    es = ls[0]
    assert np.array_equal(es[0]["memory_features"], mem) and np.array_equal(es[0]["observations"], obs)
    # map_gt WITH the snapshot: indices go through the (+1 shifted) real-time semantic map
    lms = SMNetDetectionLoader(data_path=root, clip_path=clip_path, memory_type="map_gt", semmap_path=os.path.join(snap_dir, "memory"))
    assert int(lms[0][0]["proj_indices"].max()) == 0 and lms[0][0]["memory_features"].shape == (21, 512)
