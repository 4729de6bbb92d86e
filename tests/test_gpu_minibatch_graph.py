"""MinibatchInferenceLoop(use_graph=True) (r05): the minibatch step's forward + reverse pass replayed as a hipGraph over static minibatch buffers
must walk the same trajectory as the eager loop (minibatch_loop.py:65-93 semantics: shuffles, rollover, Trainer.step(batch_size)).  The model is
the reference's svgp_regression notebook (N = 1000, 20 inducing points, minibatches of 10: examples/notebooks/svgp_regression.ipynb:100-121, 250),
the regime where the step is paced by the host's launches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(use_graph, dtype, epochs=2):
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions.gp.kernels import RBF
    from mxfusion_amd.modules.gp_modules import SVGPRegression
    from mxfusion_amd.inference import GradBasedInference, MAP, MinibatchInferenceLoop
    rng = np.random.RandomState(0)
    N, M, B = 200, 20, 10
    X = rng.rand(N, 1) * 6 - 3
    Y = np.sin(X) + 0.05 * rng.randn(N, 1)
    td = torch.float64 if dtype == 'float64' else torch.float32
    m = Model()
    m.N = Variable()
    m.X = Variable(shape=(m.N, 1))
    m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
    m.kernel = RBF(input_dim=1, variance=1., lengthscale=1., dtype=dtype)
    m.Y = SVGPRegression.define_variable(X=m.X, kernel=m.kernel, noise_var=m.noise_var, num_inducing=M, shape=(m.N, 1), dtype=dtype)
    m.Y.factor.svgp_log_pdf.jitter = 1e-6
    loop = MinibatchInferenceLoop(batch_size=B, rv_scaling={m.Y: N / B}, use_graph=use_graph)
    infr = GradBasedInference(MAP(model=m, observed=[m.X, m.Y]), grad_loop=loop, dtype=dtype)
    infr.initialize(X=(B, 1), Y=(B, 1))
    gp = m.Y.factor
    post = gp._extra_graphs[0]
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=td).cuda()
    infr.params[gp.inducing_inputs] = t(np.linspace(-3, 3, M)[:, None])
    infr.params[post.qU_mean], infr.params[post.qU_cov_W], infr.params[post.qU_cov_diag] = t(np.zeros((M, 1))), t(0.1 * rng.randn(M, M)), t(np.ones(M))
    perms = [rng.permutation(N) for _ in range(epochs)]
    infr.run(X=t(X), Y=t(Y), learning_rate=0.05, max_iter=epochs, permutations=perms)
    torch.cuda.synchronize()
    return infr.params.flat.detach().double().cpu().numpy(), [float(l) for l in loop.epoch_losses], loop


@pytest.mark.parametrize('dtype,tol', [('float64', 1e-9), ('float32', 2e-4)])
def test_replayed_minibatch_steps_walk_the_eager_trajectory(dtype, tol):
    ref, ref_losses, _ = _run(False, dtype)
    got, got_losses, loop = _run(True, dtype)
    assert loop._gstate is not None and 'graph' in loop._gstate          # the steps really were replayed
    assert np.abs(ref - np.asarray(ref)[0]).max() > 0                     # (sanity: parameters are not all equal)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= tol * scale, np.abs(got - ref).max() / scale
    assert np.allclose(got_losses, ref_losses, rtol=max(tol, 1e-9))
    assert len(ref_losses) == 2 and ref_losses[1] < ref_losses[0]        # and the optimiser moved
