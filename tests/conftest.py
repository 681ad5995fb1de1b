import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(params=[pytest.param("hip", marks=pytest.mark.gpu), "cpu_abi"])
def abi_dev(request):
    """Operator-level tests of the rows either side of the pre-training step (decode, input transform, fine-tune kernels) run twice: through
    libdig_hip.so on the MI355X (`hip`, marked gpu) and through cpu_abi/libdig_cpu.so in the GPU-less container (`cpu_abi`; the same host code
    of dig_amd above the C ABI).  Function-scoped: the CPU library is swapped in for one test only (other tests of the same module bind the HIP
    library)."""
    import torch
    if request.param == "hip":
        assert torch.cuda.is_available(), "GPU tests need an MI355X"
        from dig_amd import _lib
        _lib.lib()
        yield torch.device("cuda:0")
    else:
        from cpu_abi_util import cpu_abi_backend
        with cpu_abi_backend() as d:
            yield d


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
