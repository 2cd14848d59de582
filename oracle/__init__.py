"""CPU oracle of the per-frame recurrent inference path  --  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch fp32 on the host, plus a plain-C restatement of the
integer grid-cell indexing) of the algorithm of nhcha6/embodied-object-detection's hot path.
It exists to CHECK the HIP product and to be timed as the `cpu_baseline` leg of `bench.py`.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.  The
product package (`embodied_object_detection_amd`) never imports, calls or links anything in here
and fails loudly when its HIP extension is missing.

Pinning status
--------------
* Detic-owned code (memory read/fusion in `CustomRecurrentFPN.forward`, `CenterNetHead`,
  `ZeroShotClassifier`, `create_implicit_memory`, `box_to_image_features`,
  `project_image_features`, `LastLevelP6P7_P5`, the SMNet projector, `robot_demo.py`'s projector;
  since round 2 also `CenterNet.compute_grids / inference / predict_single_level / nms_and_topK`
  (`centernet_decode.npz`), `DeticCascadeROIHeads._forward_box / _run_stage /
  _create_proposals_from_boxes` incl. the score merge with three real `DeticFastRCNNOutputLayers`
  (`cascade.npz`), and the eval branch of `CustomRCNNRecurrent.forward` with `update_implicit_memory`,
  `inference_with_proposals`, `preprocess_spatial_memory`, `visualise_clip_image_features`
  (`memory_update.npz`; TEST_TYPE default and longterm)): pinned by golden vectors produced by
  running the reference's own functions in the development container
  (`tests/golden/gen_golden.py` -> `tests/golden/*.npz`).  Where those functions call into
  detectron2 (batched_nms, fast_rcnn_inference, paste_masks_in_image, Box2BoxTransform, ROIPooler)
  the generator injects this oracle's restatement, so the fixtures pin the reference's OWN arithmetic
  and control flow around those calls.
* detectron2 / timm / torchvision owned arithmetic (ROIAlignV2, batched NMS, Box2BoxTransform,
  fast_rcnn_inference, paste_masks_in_image, FrozenBN, timm ResNet-50 topology): those packages are
  not vendored in the reference tree and not installed here, and the reference holds no test or
  golden vector at those boundaries: **parity unpinned** there; the restatement follows the
  published upstream semantics written down in SURVEY.md Appendix A.

Every function cites the reference file:line it follows (paths relative to the reference root).
"""
