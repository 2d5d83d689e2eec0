"""Test configuration: registers the ``gpu`` marker and puts the package directory
(``jittor-clip-fewshot_amd/``: the hyphen makes it a path entry, not an import name --
exactly how the reference is used: ``from jclip import clip``) and the repo root on
``sys.path``."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "jittor-clip-fewshot_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
