"""GPU parity of the GaussianProcess / ConditionalGaussianProcess distributions against the oracle, on the
reference's own test matrix of sample / broadcast combinations (testing/components/distributions/gp/gp_test.py:37-170,
cond_gp_test.py), with a mean, with injected noise for the draws, and for the reverse mode."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gp_oracle as O  # noqa: E402

DT = 'float64'
_r = np.random.RandomState(0)
CASES = [  # X, X_isSamples, lengthscale, ls_isSamples, variance, var_isSamples, rv, rv_isSamples, num_samples   (gp_test.py:37-42)
    (_r.rand(5, 2), False, _r.rand(2) + 0.1, False, _r.rand(1) + 0.1, False, _r.rand(3, 5, 1), True, 3),
    (_r.rand(3, 5, 2), True, _r.rand(2) + 0.1, False, _r.rand(1) + 0.1, False, _r.rand(5, 1), False, 3),
    (_r.rand(3, 5, 2), True, _r.rand(3, 2) + 0.1, True, _r.rand(3, 1) + 0.1, True, _r.rand(3, 5, 1), True, 3),
    (_r.rand(5, 2), False, _r.rand(2) + 0.1, False, _r.rand(1) + 0.1, False, _r.rand(5, 1), False, 1),
    (_r.rand(7, 2), False, _r.rand(2) + 0.1, False, _r.rand(1) + 0.1, False, _r.rand(2, 7, 3), True, 2),     # D = 3 outputs
]


def _prep(a, is_samples):
    a = np.asarray(a, dtype=np.float64)
    return a if is_samples else a[None]


def _t(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64).cuda()


def _close(got, ref, tol=1e-9):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.allclose(got, ref, rtol=tol, atol=tol * max(1.0, float(np.abs(ref).max()))), np.abs(got - ref).max()


@pytest.mark.parametrize('case', range(len(CASES)))
@pytest.mark.parametrize('with_mean', [False, True])
def test_gp_dist_log_pdf_and_gradients(case, with_mean):
    from mxfusion_amd import Variable
    from mxfusion_amd.components.distributions import GaussianProcess
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    X, Xs, ls, lss, var, vs, rv, rvs, S = CASES[case]
    N, D = rv.shape[-2:]
    Xp, lsp, varp, rvp = _prep(X, Xs), _prep(ls, lss), _prep(var, vs), _prep(rv, rvs)
    mean = np.tanh(Xp @ _r.randn(2, D)) if with_mean else None
    rbf = RBF(2, True, 1., 1., 'rbf', None, DT)
    X_var = Variable(shape=(N, 2))
    mean_var = Variable(shape=(N, D)) if with_mean else None
    gp = GaussianProcess.define_variable(X=X_var, kernel=rbf, shape=(N, D), mean=mean_var, dtype=DT).factor
    dev = {n: _t(v).requires_grad_(True) for n, v in dict(X=Xp, ls=lsp, var=varp, rv=rvp).items()}
    variables = {gp.X.uuid: dev['X'], gp.rbf_lengthscale.uuid: dev['ls'], gp.rbf_variance.uuid: dev['var'], gp.random_variable.uuid: dev['rv']}
    if with_mean:
        variables[gp.mean.uuid] = _t(mean)
    got = gp.log_pdf(F=None, variables=variables)
    ora = {n: O.T(v).clone().requires_grad_(True) for n, v in dict(X=Xp, ls=lsp, var=varp, rv=rvp).items()}
    ref = O.gp_dist_log_pdf(O.RBF(2, ARD=True), ora['X'], ora['rv'], {'rbf_lengthscale': ora['ls'], 'rbf_variance': ora['var']},
                            mean=None if mean is None else O.T(mean))
    assert got.shape == (S,)
    _close(got, ref)
    got.sum().backward()
    ref.sum().backward()
    for n in ('X', 'ls', 'var', 'rv'):
        _close(dev[n].grad, ora[n].grad, 1e-7)


@pytest.mark.parametrize('case', range(4))
@pytest.mark.parametrize('with_mean', [False, True])
def test_gp_dist_draw_samples_with_injected_noise(case, with_mean):
    """gp_test.py:128-217: L eps (+ mean) with the noise replayed through the rand_gen seam."""
    from mxfusion_amd import Variable
    from mxfusion_amd.components.distributions import GaussianProcess
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    X, Xs, ls, lss, var, vs, _, _, S = CASES[case]
    N, D = 5, 1
    Xp, lsp, varp = _prep(X, Xs), _prep(ls, lss), _prep(var, vs)
    mean = np.tanh(Xp @ _r.randn(2, D)) if with_mean else None
    eps = _r.randn(S, N, D)
    rbf = RBF(2, True, 1., 1., 'rbf', None, DT)
    X_var = Variable(shape=(N, 2))
    mean_var = Variable(shape=(N, D)) if with_mean else None
    gp = GaussianProcess.define_variable(X=X_var, kernel=rbf, shape=(N, D), mean=mean_var, dtype=DT,
                                         rand_gen=MockRandomGenerator(_t(eps.reshape(-1)))).factor
    variables = {gp.X.uuid: _t(Xp), gp.rbf_lengthscale.uuid: _t(lsp), gp.rbf_variance.uuid: _t(varp)}
    if with_mean:
        variables[gp.mean.uuid] = _t(mean)
    got = gp.draw_samples(F=None, variables=variables, num_samples=S)
    ref = O.gp_dist_draw(O.RBF(2, ARD=True), O.T(Xp), {'rbf_lengthscale': O.T(lsp), 'rbf_variance': O.T(varp)}, O.T(eps),
                         mean=None if mean is None else O.T(mean))
    assert got.shape == (S, N, D)
    _close(got, ref)


@pytest.mark.parametrize('D', [1, 2])
@pytest.mark.parametrize('S', [1, 3])
@pytest.mark.parametrize('with_means', [False, True])
def test_cond_gp_dist_log_pdf_draws_and_gradients(D, S, with_means):
    """cond_gp.py:124-223 (cond_gp_test.py matrix in miniature; D = 2 also pins the reference's sum-over-outputs-before-squaring)."""
    from mxfusion_amd import Variable
    from mxfusion_amd.components.distributions import ConditionalGaussianProcess
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    rng = np.random.RandomState(10 * D + S)
    N, Nc = 6, 8
    X, Xc, Yc, rv = rng.rand(S, N, 2), rng.rand(1, Nc, 2), rng.randn(1, Nc, D), rng.randn(S, N, D)
    ls, var = rng.rand(1, 2) * 0.3 + 0.3, rng.rand(1, 1) + 0.5
    mean = np.tanh(X @ rng.randn(2, D)) if with_means else None
    mean_c = np.tanh(Xc @ rng.randn(2, D)) if with_means else None
    eps = rng.randn(S, N, D)
    rbf = RBF(2, True, 1., 1., 'rbf', None, DT)
    Xv, Xcv, Ycv = Variable(shape=(N, 2)), Variable(shape=(Nc, 2)), Variable(shape=(Nc, D))
    mv, mcv = (Variable(shape=(N, D)), Variable(shape=(Nc, D))) if with_means else (None, None)
    gp = ConditionalGaussianProcess.define_variable(X=Xv, X_cond=Xcv, Y_cond=Ycv, kernel=rbf, shape=(N, D), mean=mv, mean_cond=mcv, dtype=DT,
                                                    rand_gen=MockRandomGenerator(_t(eps.reshape(-1)))).factor
    vals = dict(X=X, Xc=Xc, Yc=Yc, ls=ls, var=var, rv=rv)
    dev = {n: _t(v).requires_grad_(True) for n, v in vals.items()}
    variables = {gp.X.uuid: dev['X'], gp.X_cond.uuid: dev['Xc'], gp.Y_cond.uuid: dev['Yc'], gp.rbf_lengthscale.uuid: dev['ls'],
                 gp.rbf_variance.uuid: dev['var'], gp.random_variable.uuid: dev['rv']}
    if with_means:
        variables[gp.mean.uuid] = _t(mean)
        variables[gp.mean_cond.uuid] = _t(mean_c)
    got = gp.log_pdf(F=None, variables=variables)
    ora = {n: O.T(v).clone().requires_grad_(True) for n, v in vals.items()}
    kp = {'rbf_lengthscale': ora['ls'], 'rbf_variance': ora['var']}
    om, omc = (O.T(mean), O.T(mean_c)) if with_means else (None, None)
    ref = O.cond_gp_dist_log_pdf(O.RBF(2, ARD=True), ora['X'], ora['Xc'], ora['Yc'], ora['rv'], kp, mean=om, mean_cond=omc)
    assert got.shape == (S,)
    _close(got, ref, 1e-8)
    got.sum().backward()
    ref.sum().backward()
    for n in vals:
        _close(dev[n].grad, ora[n].grad, 1e-6)
    draws = gp.draw_samples(F=None, variables={k: v.detach() for k, v in variables.items()}, num_samples=S)
    with torch.no_grad():
        ref_d = O.cond_gp_dist_draw(O.RBF(2, ARD=True), O.T(X), O.T(Xc), O.T(Yc), {'rbf_lengthscale': O.T(ls), 'rbf_variance': O.T(var)},
                                    O.T(eps), mean=om, mean_cond=omc)
    _close(draws, ref_d, 1e-8)


def test_reverse_mode_cholesky_matches_torch():
    """_linalg.CholFn (mxf_potrf + the closed-form reverse mode through mxf_trtri / mxf_gemm) against torch.linalg.cholesky's autograd on
    the CPU, with a gradient seed that also has upper-triangle entries (ignored: L is lower)."""
    from mxfusion_amd.components.distributions.gp._linalg import CholFn
    rng = np.random.RandomState(0)
    S, N = 3, 37
    A = rng.randn(S, N, N)
    K = A @ np.swapaxes(A, 1, 2) / N + np.eye(N) * 0.5
    W = rng.randn(S, N, N)
    Kd = _t(K).requires_grad_(True)
    L, info = CholFn.apply(Kd)
    (L * _t(W)).sum().backward()
    Kc = O.T(K).clone().requires_grad_(True)
    (torch.linalg.cholesky(Kc) * O.T(W)).sum().backward()
    assert int(info.abs().sum()) == 0
    ref = 0.5 * (Kc.grad + Kc.grad.transpose(-1, -2))             # torch returns a symmetrised gradient as well; compare the symmetric parts
    got = 0.5 * (Kd.grad + Kd.grad.transpose(-1, -2))
    _close(got, ref, 1e-9)


@pytest.mark.parametrize('S', [1, 3])
def test_gp_and_cond_gp_draws_are_differentiable(S):
    """The reparameterised draws L eps of gp.py:124-153 / cond_gp.py:185-223 keep their gradients w.r.t. the inputs and the kernel parameters
    (the reference differentiates through linalg.potrf): oracle autograd on the same injected noise."""
    from mxfusion_amd import Variable
    from mxfusion_amd.components.distributions import GaussianProcess, ConditionalGaussianProcess
    from mxfusion_amd.components.distributions.random_gen import MockRandomGenerator
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    rng = np.random.RandomState(20 + S)
    N, Nc, D = 6, 8, 2
    X, Xc, Yc = rng.rand(1, N, 2), rng.rand(1, Nc, 2), rng.randn(1, Nc, D)
    ls, var = rng.rand(1, 2) * 0.3 + 0.3, rng.rand(1, 1) + 0.5
    eps, w = rng.randn(S, N, D), rng.randn(S, N, D)
    rbf = RBF(2, True, 1., 1., 'rbf', None, DT)
    # GaussianProcess
    gp = GaussianProcess.define_variable(X=Variable(shape=(N, 2)), kernel=rbf, shape=(N, D), dtype=DT,
                                         rand_gen=MockRandomGenerator(_t(eps.reshape(-1)))).factor
    dev = {n: _t(v).requires_grad_(True) for n, v in dict(X=X, ls=ls, var=var).items()}
    got = gp.draw_samples(F=None, variables={gp.X.uuid: dev['X'], gp.rbf_lengthscale.uuid: dev['ls'], gp.rbf_variance.uuid: dev['var']},
                          num_samples=S)
    (got * _t(w)).sum().backward()
    ora = {n: O.T(v).clone().requires_grad_(True) for n, v in dict(X=X, ls=ls, var=var).items()}
    ref = O.gp_dist_draw(O.RBF(2, ARD=True), ora['X'], {'rbf_lengthscale': ora['ls'], 'rbf_variance': ora['var']}, O.T(eps))
    (ref * O.T(w)).sum().backward()
    _close(got, ref, 1e-9)
    for n in ('X', 'ls', 'var'):
        _close(dev[n].grad, ora[n].grad, 1e-6)
    # ConditionalGaussianProcess
    cgp = ConditionalGaussianProcess.define_variable(X=Variable(shape=(N, 2)), X_cond=Variable(shape=(Nc, 2)), Y_cond=Variable(shape=(Nc, D)),
                                                     kernel=rbf, shape=(N, D), dtype=DT, rand_gen=MockRandomGenerator(_t(eps.reshape(-1)))).factor
    vals = dict(X=X, Xc=Xc, Yc=Yc, ls=ls, var=var)
    dev = {n: _t(v).requires_grad_(True) for n, v in vals.items()}
    got = cgp.draw_samples(F=None, variables={cgp.X.uuid: dev['X'], cgp.X_cond.uuid: dev['Xc'], cgp.Y_cond.uuid: dev['Yc'],
                                              cgp.rbf_lengthscale.uuid: dev['ls'], cgp.rbf_variance.uuid: dev['var']}, num_samples=S)
    (got * _t(w)).sum().backward()
    ora = {n: O.T(v).clone().requires_grad_(True) for n, v in vals.items()}
    ref = O.cond_gp_dist_draw(O.RBF(2, ARD=True), ora['X'], ora['Xc'], ora['Yc'], {'rbf_lengthscale': ora['ls'], 'rbf_variance': ora['var']},
                              O.T(eps))
    (ref * O.T(w)).sum().backward()
    _close(got, ref, 1e-8)
    for n in vals:
        _close(dev[n].grad, ora[n].grad, 1e-5)
