import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GPU_DEFAULT_TIMEOUT = 300  # seconds; tests that need less or more carry their own @pytest.mark.timeout


try:  # PyTorch brings a HIP runtime of its own (a second copy in this process, beside the one the product links).  It finds the
    # GPU when it is loaded before the product has initialised the device, and may not when loaded after (seen on the GPU box:
    # "No HIP GPUs are available" where tests/test_gpu_spectral.py, the one GPU test that hands a torch tensor to the library,
    # ran without the CPU test modules that import torch at collection).  Loaded here, the order is the same in every session.
    import torch  # noqa: F401
except Exception:  # (a session without PyTorch still runs everything that does not need it)
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")
    # (pytest-timeout registers this itself where it is installed; declared here so that the marks are known without it)
    config.addinivalue_line("markers", "timeout: per-test time limit (pytest-timeout): a kernel that never ends fails its test")
    config.addinivalue_line("markers", "allow_bad_photons: the test drives a kernel into one of its loop bounds on purpose")


def pytest_collection_modifyitems(config, items):
    """EVERY GPU test has a time limit: those without a mark of their own get the default, so that a kernel that never
    ends fails its test instead of stalling the run (the kernels are bounded by construction, DESIGN.md section 4.7 --
    this is the second line of defence)."""
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(GPU_DEFAULT_TIMEOUT, method="thread"))


@pytest.fixture(autouse=True, scope="session")
def _one_hip_runtime():
    """ONE HIP runtime per process, asserted (mcbrat3d_amd/_capi.py: hip_runtimes): PyTorch is imported above, before anything
    loads libmcbrat_hip.so, so the library binds to PyTorch's copy of libamdhip64.  Two copies -- the product loaded first,
    PyTorch after it -- would make every stream and device pointer the tests pass between the two a handle of another library.
    Checked when the session starts (nothing loaded yet, or one) and when it ends (everything loaded)."""
    from mcbrat3d_amd import _capi
    _capi.assert_single_hip_runtime()
    yield
    _capi.assert_single_hip_runtime()


@pytest.fixture(autouse=True)
def _no_photon_dropped_by_a_loop_bound(request):
    """Every integrator a GPU test finalises must report badPhotons == 0 (include/mcbrat.h: photons dropped because a
    loop bound of the kernels was reached), unless the test provokes a bound on purpose (allow_bad_photons)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from mcbrat3d_amd import integrator as I
    orig, seen = I.Integrator.finalize, []

    def finalize(self):
        if getattr(self, "_ctx", None):
            bad = int(self.counters()["badPhotons"])
            seen.append((bad, self.firstDrop() if bad else ""))
        orig(self)

    I.Integrator.finalize = finalize
    try:
        yield
    finally:
        I.Integrator.finalize = orig
    if request.node.get_closest_marker("allow_bad_photons") is None:
        assert all(b == 0 for b, _ in seen), "photons dropped by a loop bound: %r" % ([s for s in seen if s[0]],)
