import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    from vdm4cdm_amd import _lib
    return _lib.lib()


@pytest.fixture(autouse=True, scope="session")
def _quiesce_gpu_before_freeing():
    """The GPU suite runs ~360 tests in ONE process; many of them end with `torch.cuda.empty_cache()` right after deleting a network whose
    last kernels (side streams, replayed hipGraphs, rocFFT) may still be in flight.  In one of two otherwise green runs of round 4 the
    process died with a segmentation fault inside that call (`hipFree`, a runtime thread without Python frame; same signature as the
    round-3 report in DESIGN.md section 7).  The product never calls empty_cache; here every call first drains the device and collects
    garbage, so that memory is only returned to the driver while nothing runs.  (tests/_stream_hazard_worker.py keeps exercising frees
    under load on purpose - with PYTORCH_NO_CUDA_MEMORY_CACHING=1 every free is a hipFree - in a process of its own.)"""
    import gc
    try:
        import torch
    except Exception:
        yield
        return
    if not torch.cuda.is_available():
        yield
        return
    orig = torch.cuda.empty_cache

    def drained_empty_cache():
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()
        orig()

    torch.cuda.empty_cache = drained_empty_cache
    try:
        yield
    finally:
        torch.cuda.empty_cache = orig


@pytest.fixture(autouse=True)
def _release_gpu_state_between_tests():
    """After every GPU test: drain the device, drop the library's per-stream scratch caches and return the cached blocks to the driver, so
    that no test inherits device memory (or a graph-pool remnant) of the ~360 tests that ran before it in the same process.  The round-4
    full-suite runs died twice (of seven) inside the hipGraph sampler tests at the end of test_unet_gpu.py - a segfault in hipFree and an
    abort at a device-to-host copy - and never in a test file run on its own."""
    yield
    try:
        import torch
    except Exception:
        return
    if not torch.cuda.is_available():
        return
    import gc
    from vdm4cdm_amd import hip_ops as ops
    torch.cuda.synchronize()
    for cache in (ops.Conv._ws, ops._gn_ws, ops._skip_ws, ops._red_ws):
        cache.clear()
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
