import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from embodied_object_detection_amd import build_model, setup_cfg
from embodied_object_detection_amd.checkpoint import synthetic_state_dict
from embodied_object_detection_amd.data.synthetic import SyntheticSequence
dev = torch.device("cuda:0")
cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5, "MODEL.DEVICE", "cuda:0"])
m = build_model(cfg, synthetic_state_dict(0))
N = 30
seq = SyntheticSequence(0, H=640, W=640, n_frames=N, map_w=200, map_h=200, cell=0.2)
frames = []
for i in range(N):
    f = seq.frame(i); f["image"] = f["image"].to(dev); f["proj_indices"] = torch.from_numpy(f["proj_indices"][..., 0]).to(dev); frames.append(f)
def step(i):
    if frames[i]["memory_reset"]: m.reset_memory(seq.n_cells)
    m.inference_frame(frames[i], materialize=False, next_frame=frames[i + 1] if i + 1 < N else None)
for i in range(5): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter()
for i in range(5, N - 1): step(i)
host = time.perf_counter() - t
pr.disable()
torch.cuda.synchronize()
print("host ms/frame", host / (N - 6) * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
