"""The eval collective on the real RCCL backend.  Only one GPU is available to the tests, so the process group has ONE rank:
the same `gather_records` code path (`all_reduce(SUM)` of the fixed-shape record buffer, `engine/eval_loop.py`) that the 8-GPU
run uses executes through RCCL here; the multi-rank arithmetic is covered by the world-2 gloo test on the CPU
(`tests/test_host_cpu.py::test_sharded_eval_world2_equals_single_process`)."""
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gather_records_through_rccl_world1():
    import torch.distributed as dist
    from embodied_object_detection_amd.engine.eval_loop import KIND_DET, KIND_GT, RecordBuffer, evaluate_gathered, gather_records
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        rec = RecordBuffer(64)
        rng = np.random.RandomState(0)
        for i in range(20):
            x, y = rng.uniform(0, 500, 2)
            rec.add([KIND_DET, 0, i % 4, float(i % 3), float(rng.rand()), x, y, x + 50, y + 40, 0])
            rec.add([KIND_GT, 0, i % 4, float(i % 3), 0.0, x + 2, y + 1, x + 50, y + 40, 0])
        local = rec.to_tensor("cpu").numpy()
        buf = gather_records(rec, 0, 1, dev)                       # all_reduce(SUM) over RCCL, world size 1
        assert buf.shape == (1, 64, 10) and np.array_equal(buf[0], local[:-1])      # the last row of the message is the status row
        # a second, larger message of the size class a real eval produces (~2.6 MB)
        t = torch.arange(1 << 16, dtype=torch.float32, device=dev).repeat(10)
        ref = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        assert torch.equal(t, ref)
        ap = evaluate_gathered(buf, 20)["all"]
        assert ap["AP50"] > 50.0
    finally:
        dist.barrier()
        dist.destroy_process_group()
