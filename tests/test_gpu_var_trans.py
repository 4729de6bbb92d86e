"""PositiveTransformation (Softplus) through the API on the MI355X: testing/components/variables/var_trans_test.py:28-56."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_softplus_round_trip_of_a_negative_value():
    from mxfusion_amd.components.variables import PositiveTransformation
    v = torch.tensor([-10.], dtype=torch.float64, device='cuda')
    p = PositiveTransformation()
    pos = p.transform(v)
    inv = p.inverseTransform(pos)
    assert float(v) < 0 and float(pos) > 0 and float(inv) < 0
    np.testing.assert_allclose(inv.cpu().numpy()[0], -10., rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("x, dtype, rtol, atol", [(10., torch.float64, 1e-7, 1e-10), (1e-30, torch.float64, 1e-7, 1e-10),
                                                 (5., torch.float32, 1e-4, 1e-5), (1e-6, torch.float32, 1e-4, 1e-5)])
def test_softplus_numerical(x, dtype, rtol, atol):
    from mxfusion_amd.components.variables import PositiveTransformation
    p = PositiveTransformation()
    xt = torch.tensor([x], dtype=dtype, device='cuda')
    pos = p.transform(xt)
    inv = p.inverseTransform(pos)
    xn = xt.cpu().numpy()
    np_pos = np.log1p(np.exp(xn))
    np_inv = np.log(np.expm1(np_pos))
    np.testing.assert_allclose(pos.cpu().numpy(), np_pos, rtol=rtol, atol=atol)
    np.testing.assert_allclose(inv.cpu().numpy(), np_inv, rtol=rtol, atol=atol)
    np.testing.assert_allclose(inv.cpu().numpy(), xn, rtol=rtol, atol=atol)
    assert pos.dtype == dtype
