import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def _available_cores() -> int:
    """Cores this process may really use (affinity mask and cgroup quota), not the host's total: the CPU oracle is torch on the host
    cores, and with torch's default of one thread per LOGICAL core of the host a job confined to a 16-core share oversubscribes
    itself several times over (the oracle's 640x640 frame then takes ~20 s instead of ~3.5 s)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh2:
                        n = min(n, max(1, q // int(fh2.read().split()[0])))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, 32))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch
    torch.set_num_threads(_available_cores())


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def synthetic_sd():
    from embodied_object_detection_amd.checkpoint import synthetic_state_dict
    return synthetic_state_dict(0)
