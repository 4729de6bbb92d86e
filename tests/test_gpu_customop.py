"""The reference's custom operators on the MI355X, on the exact-integer cases of testing/util/customop_test.py:37-82 (SURVEY 8(c) KAT
list) plus make_diagonal (util/customop.py:22-81)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("data, isSamples, shape", [
    (np.array([[2, 3, 4], [3, 4, 5]]), False, (5, 4, 2, 3)),
    (np.array([[2, 3, 4], [3, 4, 5]]), True, (2, 4, 5, 3)),
    (np.array([2, 3, 4]), False, (2, 4, 5, 3)),
])
def test_broadcast_to_w_samples_forward_and_backward(data, isSamples, shape):
    from mxfusion_amd.util.customop import broadcast_to_w_samples
    x = torch.as_tensor(data, dtype=torch.float64).cuda().requires_grad_(True)
    res = broadcast_to_w_samples(None, x, shape, isSamples)
    res_np = np.empty(shape)
    if isSamples:
        res_np[:] = data.reshape(*((data.shape[0],) + (1,) * (len(shape) - len(data.shape)) + data.shape[1:]))
    else:
        res_np[:] = data
    assert tuple(res.shape) == shape
    assert np.all(res_np == res.detach().cpu().numpy())
    w = np.random.RandomState(0).rand(*shape)
    (res * torch.as_tensor(w).cuda()).sum().backward()
    if isSamples:
        grad_np = w.reshape(*((data.shape[0], -1) + data.shape[1:])).sum(1)
    else:
        grad_np = w.reshape(*((-1,) + data.shape)).sum(0)
    assert tuple(x.grad.shape) == data.shape
    assert np.allclose(x.grad.cpu().numpy(), grad_np)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
@pytest.mark.parametrize('shape', [(3,), (2, 5), (4, 1, 7), (2, 3, 64)])
def test_make_diagonal_forward_and_backward(dtype, shape):
    from mxfusion_amd.util.customop import make_diagonal
    rng = np.random.RandomState(sum(shape))
    a = rng.randint(-5, 6, size=shape).astype(np.float64)
    x = torch.as_tensor(a, dtype=dtype).cuda().requires_grad_(True)
    out = make_diagonal(None, x)
    n = shape[-1]
    ref = a[..., None] * np.eye(n)
    assert tuple(out.shape) == shape + (n,)
    assert np.all(out.detach().double().cpu().numpy() == ref)                  # exact
    w = rng.randint(-3, 4, size=shape + (n,)).astype(np.float64)
    (out * torch.as_tensor(w, dtype=dtype).cuda()).sum().backward()
    assert np.all(x.grad.double().cpu().numpy() == np.diagonal(w, axis1=-2, axis2=-1))
