import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


GEMM_MODES = ["f32", "bf16x6"]


@pytest.fixture(params=GEMM_MODES)
def gemm_mode(request):
    """Both arithmetic modes of the dense products under the SAME bounds: "f32" (v_mfma_f32_16x16x4_f32) and "bf16x6" (every fp32
    operand split exactly into three bf16 terms, six bf16 MFMA products per fp32 product, fp32 accumulate).  The learner-level parity
    tests (golden cases, full-size configurations, ranks == one process) take this fixture, so the driver's `pytest -m gpu` runs and
    prints the margins of both."""
    return request.param


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter):
    """achieved parity margins (tests/parity.py): worst error / bound per case"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import parity
    except ImportError:
        return
    lines = parity.summary_lines()
    if lines:
        terminalreporter.write_sep("-", "parity margins: max|got-ref| vs tol * max|ref| (north-star 1e-4 fp32)")
        for l in lines:
            terminalreporter.write_line(l)
        out = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out):
            with open(os.path.join(out, "parity_margins.txt"), "w") as f:
                f.write("\n".join(lines) + "\n")
