import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build (or reuse) the native artefacts once per session.  On the GPU box everything is prebuilt and
    travels with the snapshot; here it compiles in well under a minute."""
    from figbird_amd import build as fbuild
    # build_lib() is a no-op when libfighip.so is newer than every source under csrc/; a stale or failing build fails the
    # session, so a green run always means HEAD's kernels were the ones tested
    fbuild.build_lib()
    fbuild.build_figfill()
    from tools import build_test_infra
    build_test_infra.build()
    yield
