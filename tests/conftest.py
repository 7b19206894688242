import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_library():
    """The HIP library is built in-tree (hipcc cross-compiles gfx950 without a GPU) when it is missing or older than its
    sources — e.g. in a fresh checkout, where the git-ignored .so does not exist.  The product itself never builds or
    falls back on its own: overcooked_ai_amd._lib.load() raises when the library is absent."""
    from overcooked_ai_amd import build

    build.build_extension()


@pytest.fixture(scope="session")
def manifest():
    import json

    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def small_fixtures():
    import json

    with open(os.path.join(GOLDEN, "ref_small_fixtures.json")) as f:
        return json.load(f)
