import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _fresh_float32_guard():
    """The float32 guard of the SVGP module is process-wide and sticky (modules/gp_modules/_fused.py): every test starts with it cleared."""
    try:
        import torch
        if torch.cuda.is_available():
            from mxfusion_amd.modules.gp_modules._fused import Float32Guard
            from mxfusion_amd import ops
            Float32Guard.reset()
            ops.svgp_cond_nowait(reset=True)
    except Exception:       # noqa: BLE001 -- CPU-only runs, library absent: nothing to reset
        pass
    yield


@pytest.fixture
def float32_forms_on_small_problems():
    """Since r05 a float32 SVGP call below SVGPRegressionLogPdf.SMALL_F64_ELEMS covariances is evaluated in float64 inside.  The tests of the
    float32 forms THEMSELVES (explicit / whitened tiers, the guard's moves, the padded shapes) use small shapes to stay fast: they switch
    that rule off and exercise the forms they were written for."""
    from mxfusion_amd.modules.gp_modules.svgp_regression import SVGPRegressionLogPdf
    old = SVGPRegressionLogPdf.SMALL_F64_ELEMS
    SVGPRegressionLogPdf.SMALL_F64_ELEMS = 0
    yield
    SVGPRegressionLogPdf.SMALL_F64_ELEMS = old
