import os
import sys

import pytest

os.environ.setdefault("MSCOMP_AMD_TEST_HOOKS", "1")      # the mscomp_amd_debug_set_* kernel switches work only in processes that ask for them before the library loads
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import loader
    loader.build()
    loader.load_oracle()
    return loader


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ms_compress_amd as m
    ctx = m.Context()
    yield ctx
    ctx.close()
