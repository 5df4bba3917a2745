import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# The drivers and the Sampler default to the fp32-accurate bf16x3 arithmetic since round 5 (t2ms_amd/sampler.py DEFAULT_MATH).
# The suite pins the EXACT f32 MFMA arithmetic as that default so the headline kernels -- and the bitwise class-API == fused
# sampler properties -- stay covered; bf16x3 is selected explicitly by its own tests (chains, chain1000, forwards, weight
# updates), and tests/test_host_logic.py asserts the shipped default.
os.environ.setdefault("T2S_DEFAULT_MATH", "f32")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
