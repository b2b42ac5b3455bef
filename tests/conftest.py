import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'quadtree-mpnnlstm_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """The shared library is a build artefact (git-ignored): build it when a fresh checkout has none, so that the ABI /
    loader tests do not depend on __graft_entry__.build() having run first (hipcc cross-compiles gfx950 without a GPU)."""
    lib = os.path.join(PKG, 'qtmpnn', 'libqtmpnn_hip.so')
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(['make', '-C', os.path.join(PKG, 'csrc')], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
