import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    from regengo_amd import build
    return build.build_all()


@pytest.fixture(scope="session")
def kats():
    return json.load(open(os.path.join(GOLDEN, "kats.json")))


@pytest.fixture(scope="session")
def progs():
    return json.load(open(os.path.join(GOLDEN, "progs.json")))


@pytest.fixture(scope="session")
def corpus():
    return json.load(open(os.path.join(GOLDEN, "e2e_corpus.json")))


@pytest.fixture(scope="session")
def hostlib(built):
    from tests import _hosttest
    return _hosttest
