"""Normal distribution through the API on the MI355X, on the sample-axis combinations of
testing/components/distributions/normal_test.py:25-110 (log_pdf against scipy.stats.norm, draw_samples with injected noise)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

_R = np.random.RandomState(0)


def _reshape(a, is_samples, n_dim):
    """testing/components/distributions (numpy_array_reshape): put the sample axis first, pad to n_dim."""
    if is_samples:
        return a.reshape((a.shape[0],) + (1,) * (n_dim - a.ndim) + a.shape[1:])
    return a.reshape((1,) * (n_dim - a.ndim) + a.shape)


def _dev(a, dtype, is_samples):
    t = torch.as_tensor(a, dtype=dtype).cuda()
    return t if is_samples else t[None]


@pytest.mark.parametrize("dtype, mean, mean_s, var, var_s, rv, rv_s, num_samples", [
    (torch.float64, _R.rand(5, 3, 2), True, _R.rand(3, 2) + 0.1, False, _R.rand(5, 3, 2), True, 5),
    (torch.float64, _R.rand(3, 2), False, _R.rand(5, 3, 2) + 0.1, True, _R.rand(5, 3, 2), True, 5),
    (torch.float64, _R.rand(3, 2), False, _R.rand(3, 2) + 0.1, False, _R.rand(5, 3, 2), True, 5),
    (torch.float64, _R.rand(3, 2), False, _R.rand(3, 2) + 0.1, False, _R.rand(3, 2), False, 1),
    (torch.float32, _R.rand(5, 3, 2), True, _R.rand(3, 2) + 0.1, False, _R.rand(5, 3, 2), True, 5),
])
def test_normal_log_pdf_sample_axis_combinations(dtype, mean, mean_s, var, var_s, rv, rv_s, num_samples):
    from scipy.stats import norm
    from mxfusion_amd.components.distributions import Normal
    any_s = mean_s or var_s or rv_s
    rv_shape = rv.shape[1:] if rv_s else rv.shape
    n_dim = 1 + rv.ndim if (any_s and not rv_s) else rv.ndim
    ref = norm.logpdf(_reshape(rv, rv_s, n_dim), _reshape(mean, mean_s, n_dim), np.sqrt(_reshape(var, var_s, n_dim)))
    normal = Normal.define_variable(shape=rv_shape, dtype='float64' if dtype == torch.float64 else 'float32').factor
    variables = {normal.mean.uuid: _dev(mean, dtype, mean_s), normal.variance.uuid: _dev(var, dtype, var_s),
                 normal.random_variable.uuid: _dev(rv, dtype, rv_s)}
    got = normal.log_pdf(F=None, variables=variables)
    assert got.dtype == dtype
    assert (got.shape[0] > 1) == any_s
    if any_s:
        assert got.shape[0] == num_samples
    rtol, atol = (1e-7, 1e-10) if dtype == torch.float64 else (1e-4, 1e-5)
    assert np.allclose(ref, got.double().cpu().numpy().reshape(ref.shape), rtol=rtol, atol=atol)


@pytest.mark.parametrize("dtype, mean, mean_s, var, var_s, rv_shape, num_samples", [
    (torch.float64, _R.rand(5, 3, 2), True, _R.rand(3, 2) + 0.1, False, (3, 2), 5),
    (torch.float64, _R.rand(3, 2), False, _R.rand(5, 3, 2) + 0.1, True, (3, 2), 5),
    (torch.float64, _R.rand(3, 2), False, _R.rand(3, 2) + 0.1, False, (3, 2), 5),
    (torch.float64, _R.rand(5, 3, 2), True, _R.rand(5, 3, 2) + 0.1, True, (3, 2), 5),
    (torch.float32, _R.rand(5, 3, 2), True, _R.rand(3, 2) + 0.1, False, (3, 2), 5),
])
def test_normal_draw_samples_with_injected_noise(dtype, mean, mean_s, var, var_s, rv_shape, num_samples):
    from mxfusion_amd.components.distributions import Normal
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    n_dim = 1 + len(rv_shape)
    rand = np.random.RandomState(1).randn(num_samples, *rv_shape)
    ref = _reshape(mean, mean_s, n_dim) + rand * np.sqrt(_reshape(var, var_s, n_dim))
    normal = Normal.define_variable(shape=rv_shape, dtype='float64' if dtype == torch.float64 else 'float32',
                                    rand_gen=MockRandomGenerator(torch.as_tensor(rand.flatten(), dtype=dtype).cuda())).factor
    variables = {normal.mean.uuid: _dev(mean, dtype, mean_s), normal.variance.uuid: _dev(var, dtype, var_s)}
    got = normal.draw_samples(F=None, variables=variables, num_samples=num_samples)
    assert got.dtype == dtype and got.shape[0] == num_samples
    rtol, atol = (1e-7, 1e-10) if dtype == torch.float64 else (1e-4, 1e-5)
    assert np.allclose(ref, got.double().cpu().numpy(), rtol=rtol, atol=atol)


def test_normal_draw_samples_device_generator_moments():
    """normal_test.py:112-135: the real generator (no mock): 100 000 draws of N(0.5, 2) have the right mean and variance."""
    from mxfusion_amd.components.distributions import Normal
    torch.manual_seed(0)
    normal = Normal.define_variable(shape=(1,), dtype='float32').factor
    variables = {normal.mean.uuid: torch.tensor([[0.5]], device='cuda'), normal.variance.uuid: torch.tensor([[2.0]], device='cuda')}
    s = normal.draw_samples(F=None, variables=variables, num_samples=100000)
    assert abs(float(s.mean()) - 0.5) < 0.02 and abs(float(s.var()) - 2.0) < 0.05
