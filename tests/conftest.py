import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # MARL_TEST_GEMM_MODE=bf16x6: the whole suite with the opt-in split kernels where they exist (same bounds; the margins
    # table of such a run is committed beside the fp32 one)
    mode = os.environ.get("MARL_TEST_GEMM_MODE")
    if mode:
        from marl_amd.network import mixer
        mixer.DEFAULT_GEMM_MODE = mode


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
