"""Driver entry points: build() compiles every HIP source for gfx950; smoke() runs one tiny
QMIX train step + batched rollout on cuda:0 and checks it against the CPU oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build() -> None:
    subprocess.check_call(["make", "-j8", "-C", os.path.join(ROOT, "marl_amd", "csrc")])
    import marl_amd  # noqa: F401
    from marl_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libmarl_hip.so was not produced"


def smoke() -> None:
    from tests import smoke_impl
    smoke_impl.run()


if __name__ == "__main__":
    build()
    print("build ok")
