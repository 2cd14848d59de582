import sys, json
sys.path.insert(0, '.')
import bench, torch
from embodied_object_detection_amd.checkpoint import synthetic_state_dict
r = bench.train_step_variant(synthetic_state_dict(0))
print(json.dumps({k: r[k] for k in ("ms_per_step", "achieved_tflops", "frac_of_fp32_mfma_peak")}))
