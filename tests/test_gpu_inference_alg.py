"""Inference drivers on the MI355X: testing/inference/inference_alg_test.py:32-90 (the SET_<uuid> parameter side channel of
inference_alg.py:236-251; the global default dtype of common/config.py:18) and testing/inference/map_test.py-style MAP of a Normal."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_set_parameters_side_channel():
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.inference import Inference, InferenceAlgorithm

    class SetValue(InferenceAlgorithm):
        def __init__(self, x, y, model, observed, extra_graphs=None):
            self.x_val, self.y_val = x, y
            super(SetValue, self).__init__(model=model, observed=observed, extra_graphs=extra_graphs)

        def compute(self, F, variables):
            self.set_parameter(variables, self.model.x, self.x_val)
            self.set_parameter(variables, self.model.y, self.y_val)

    m = Model()
    m.x = Variable(shape=(2,))
    m.y = Variable(shape=(3, 4))
    np.random.seed(0)
    x_np, y_np = np.random.rand(2), np.random.rand(3, 4)
    infr = Inference(SetValue(torch.as_tensor(x_np).cuda(), torch.as_tensor(y_np).cuda(), m, []), dtype='float64')
    infr.run()
    assert np.allclose(infr.params[m.x].cpu().numpy(), x_np)
    assert np.allclose(infr.params[m.y].cpu().numpy(), y_np)


def test_change_default_dtype_and_map_of_a_normal():
    """MAP of (mu, s) for Y ~ N(mu, s) with the library-wide default dtype switched to float64: parameters come out float64 and the
    loss after a few Adam steps equals the oracle's (MXNet Adam, var_trans softplus)."""
    from mxfusion_amd.common import config
    from mxfusion_amd import Model, Variable
    from mxfusion_amd.components.variables import PositiveTransformation
    from mxfusion_amd.components.distributions import Normal
    from mxfusion_amd.inference import GradBasedInference, MAP
    from oracle import gp_oracle as O
    old = config.DEFAULT_DTYPE
    config.DEFAULT_DTYPE = 'float64'
    try:
        np.random.seed(0)
        data = np.random.randn(100) * np.sqrt(5.) + 3.
        m = Model()
        m.mu = Variable(initial_value=0.1)
        m.s = Variable(transformation=PositiveTransformation(), initial_value=1.5)
        m.Y = Normal.define_variable(mean=m.mu, variance=m.s, shape=(100,))
        infr = GradBasedInference(inference_algorithm=MAP(model=m, observed=[m.Y]))
        infr.run(Y=torch.as_tensor(data, dtype=torch.float64).cuda(), learning_rate=0.1, max_iter=5)
        assert infr.params[m.mu].dtype == torch.float64 and infr.params[m.s].dtype == torch.float64
        # the same five steps through the oracle
        raw = {'mu': O.T([0.1]), 's': O.inv_softplus(O.T([1.5]))}
        opt = O.MXNetAdam(0.1)
        Y = O.T(data)
        for _ in range(5):
            leaves = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
            loss = -O.factor_sum(O.normal_log_pdf(leaves['mu'][None], O.softplus(leaves['s'])[None], Y[None]))
            loss.backward()
            raw = opt.step({k: v.detach() for k, v in leaves.items()}, {k: v.grad for k, v in leaves.items()}, batch_size=1)
        assert np.allclose(infr.params[m.mu].cpu().numpy().ravel(), raw['mu'].numpy(), rtol=1e-9)
        assert np.allclose(infr.params[m.s].cpu().numpy().ravel(), O.softplus(raw['s']).numpy(), rtol=1e-9)
    finally:
        config.DEFAULT_DTYPE = old
