import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "az-net_amd", "lib")
for p in (LIB, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# The full-head parity tests (head at launch sizes, whole searches against the pure-CPU oracle) run twice: on the default
# fp32-input MFMA and with int6 as three bf16 terms per fp32 operand (az_set_gemm_mode 3: all 24 mantissa bits, fp32's
# exponent range, fp32 accumulation) -- same tolerances, same trees.  Chosen inside pytest, not by an environment variable.
@pytest.fixture(scope="session", params=[0, 3], ids=["fp32_mfma", "bf16x3_exact_split"])
def gemm_mode(request):
    return request.param
