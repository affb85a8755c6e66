import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # build what is missing (normally __graft_entry__.build() has run already; hipcc cross-compiles without a GPU)
    import subprocess
    pkg = os.path.join(ROOT, "humanoid-navigation-using-mpc-ldcbf_amd")
    if not os.path.exists(os.path.join(pkg, "liblipmpc.so")):
        subprocess.check_call(["make", "-C", os.path.join(pkg, "csrc"), "-j", "8"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liblipmpc_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
