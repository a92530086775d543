import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the HIP library and the C oracle
    once (hipcc cross-compiles gfx950 without a GPU) so that neither suite depends on a previous build step."""
    need = [os.path.join(ROOT, "deepviewagg_amd", "csrc", "libdva_hip.so"),
            os.path.join(ROOT, "oracle", "liboracle_mapping.so"),
            os.path.join(ROOT, "oracle", "liboracle_deepset.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


def pytest_collection_modifyitems(config, items):
    # gpu-marked tests are skipped (not failed) when collected on a box without a HIP device
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _poison_freed_device_memory(request):
    """DVA_TEST_POISON=1: before every GPU test, fill 2 GiB of the caching allocator's free list with 0xFF bytes (NaN as
    fp32 / bf16, -1 as an index), so that a kernel reading memory it was supposed to write first fails loudly instead of
    passing on whatever the previous test left there."""
    if os.environ.get("DVA_TEST_POISON") == "1" and "gpu" in request.keywords and torch.cuda.is_available():
        big = torch.empty(1 << 31, dtype=torch.uint8, device="cuda:0").fill_(0xFF)
        small = [torch.empty(1 << 16, dtype=torch.uint8, device="cuda:0").fill_(0xFF) for _ in range(64)]
        del big, small
    yield


def load_golden(name):
    """Committed golden vector produced by oracle/gen_golden.py from the reference's own source."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def t(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def state_dict_from(gold, prefix="sd/"):
    return {k[len(prefix):]: torch.from_numpy(np.array(v)) for k, v in gold.items() if k.startswith(prefix)}


@pytest.fixture
def golden():
    return load_golden
