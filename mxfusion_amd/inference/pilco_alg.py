"""PILCOAlgorithm (mxfusion/inference/pilco_alg.py:21-90): roll a learnt GP dynamics model forward under a policy and return the
accumulated cost, differentiable w.r.t. the policy's parameters.

Every time step is one call of the model's prediction algorithm (pilco_alg.py:80) -- mxf_gram (cross Gram against the conditioning
inputs), mxf_trsm, mxf_gemm, mxf_coldot -- whose test inputs are the previous step's prediction and action; the reverse pass goes
through the same kernels (mxf_gram_bwd for dK -> dX*, mxf_trsm with the transposed factor).  `policy`, `cost_function` and
`initial_state_generator` are caller-supplied torch callables (gluon HybridBlocks in the reference)."""
import numpy as np
import torch

from ..common import config
from .inference_alg import SamplingAlgorithm


class PILCOAlgorithm(SamplingAlgorithm):
    def __init__(self, model, observed, cost_function, policy, n_time_steps, initial_state_generator, extra_graphs=None, num_samples=3,
                 ctx=None, dtype=None):
        super(PILCOAlgorithm, self).__init__(model, observed, extra_graphs=extra_graphs)
        self.cost_function = cost_function
        self.policy = policy
        self.initial_state_generator = initial_state_generator
        self.n_time_steps = n_time_steps
        self.num_samples = num_samples
        self.dtype = dtype if dtype is not None else config.DEFAULT_DTYPE
        self.mxnet_context = ctx            # resolved by the inference driver (None = the current HIP device)

    def _tensor(self, a, like):
        if not isinstance(a, torch.Tensor):
            a = torch.as_tensor(np.asarray(a))
        return a.to(device=like.device, dtype=like.dtype)

    def compute(self, F, variables):
        """pilco_alg.py:55-90.  States travel as (num_samples, 1, state_dim): one test point per sampled trajectory, i.e. the S axis of
        the prediction call is the trajectory axis.  The action is kept (num_samples, 1, action_dim) from the first step on (the
        reference's first cost evaluation sees the un-expanded (num_samples, action_dim) action, pilco_alg.py:74,84)."""
        like = variables[self.model.X]
        S = self.num_samples
        s_0 = self._tensor(self.initial_state_generator(S), like)
        a_t = self.policy(s_0).reshape(S, 1, -1)
        x_t = torch.cat([s_0.reshape(S, 1, -1), a_t], dim=2)
        cost = 0
        for t in range(self.n_time_steps):
            variables[self.model.X] = x_t
            res = self.model.Y.factor.predict(F, variables, targets=[self.model.Y], num_samples=S)[0]
            s_next = res[0] if isinstance(res, (tuple, list)) else res          # (mean, variance) -> the mean; samples as they are
            cost = cost + self.cost_function(s_next, a_t)
            a_t = self.policy(s_next).reshape(S, 1, -1)
            x_t = torch.cat([s_next, a_t], dim=2)
        total_cost = torch.sum(cost)
        return total_cost, total_cost
