import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU should fail loudly, not skip: the product has no CPU path.
    # Plain runs without a GPU skip the gpu-marked tests.
    if _has_gpu():
        return
    selected = config.getoption("-m") or ""
    if "gpu" in selected and "not gpu" not in selected:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (gpu-marked tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def mm():
    import mmiss_amd  # noqa: F401  (registers the package alias)
    return mmiss_amd


@pytest.fixture(scope="session", autouse=True)
def _native_artifacts():
    """The tests need libmmiss.so (the product) and oracle/libmmiss_oracle.so (the checker). They are git-ignored build
    products: build them once if a fresh checkout has not run __graft_entry__.build() yet."""
    lib = os.path.join(ROOT, "multimodal-image-similarity-search_amd", "libmmiss.so")
    ora = os.path.join(ROOT, "oracle", "libmmiss_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        import __graft_entry__

        __graft_entry__.build()
    yield
