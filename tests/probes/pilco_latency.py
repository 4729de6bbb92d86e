"""Latency of the PILCO rollout (GP predict in a loop, pilco_alg.py:72-90): N conditioning points, S trajectories, T time steps.
usage: pilco_latency.py [N] [S] [T] [dtype] [graph]"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
DT = sys.argv[4] if len(sys.argv) > 4 else 'float64'
GRAPH = int(sys.argv[5]) if len(sys.argv) > 5 else 0
from mxfusion_amd import Model, Variable
from mxfusion_amd.components.variables import PositiveTransformation
from mxfusion_amd.components.distributions.gp.kernels import RBF
from mxfusion_amd.modules.gp_modules import GPRegression
from mxfusion_amd.inference import GradBasedInference, MAP, GradTransferInference, PILCOAlgorithm, BatchInferenceLoop
td = torch.float64 if DT == 'float64' else torch.float32
rng = np.random.RandomState(0)
X = rng.rand(N, 4); Y = np.stack([np.sin(X @ rng.randn(4)) for _ in range(3)], 1) + 0.05 * rng.randn(N, 3)
t = lambda a: torch.as_tensor(a, dtype=td).cuda()
m = Model(); m.N = Variable(); m.X = Variable(shape=(m.N, 4))
m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
m.kernel = RBF(input_dim=4, variance=1, lengthscale=1, ARD=True, dtype=DT)
m.Y = GPRegression.define_variable(X=m.X, kernel=m.kernel, noise_var=m.noise_var, shape=(m.N, 3), dtype=DT)
m.Y.factor.gp_log_pdf.jitter = 1e-6
infr = GradBasedInference(inference_algorithm=MAP(model=m, observed=[m.X, m.Y]), dtype=DT)
infr.run(X=t(X), Y=t(Y), max_iter=3, learning_rate=0.1)


class Policy(torch.nn.Module):
    def __init__(self):
        super().__init__(); self.l1 = torch.nn.Linear(3, 100); self.l2 = torch.nn.Linear(100, 1)
    def forward(self, x):
        return torch.tanh(self.l2(torch.relu(self.l1(x)))) * 2


def cost_fn(state, action):
    return (2. * (state[:, :, 0:1] - 1) ** 2).sum(-1) + (.001 * action ** 2).sum(-1) + (.1 * state[:, :, 2:3] ** 2).sum(-1)


policy = Policy().to(td).cuda()
s0 = t(rng.rand(S, 3))
alg = PILCOAlgorithm(model=m, observed=[m.X, m.Y], cost_function=cost_fn, policy=policy, n_time_steps=T, initial_state_generator=lambda n: s0, num_samples=S)
loop = BatchInferenceLoop(use_graph=bool(GRAPH))
ip = GradTransferInference(alg, infr_params=infr.params, train_params=list(policy.parameters()), grad_loop=loop, dtype=DT)
Xd, Yd = t(X), t(Y)
ip.initialize(X=Xd, Y=Yd)
ex = ip.create_executor()
from mxfusion_amd.inference.batch_loop import _Adam
opt = _Adam(ip.params, 1e-3)
for it in range(4):
    loss = loop.step(ex, [Xd, Yd], ip.params); opt.step()
torch.cuda.synchronize(); t0 = time.time()
K = 5
for it in range(K):
    loss = loop.step(ex, [Xd, Yd], ip.params); opt.step()
torch.cuda.synchronize(); dt = (time.time() - t0) / K
print('N=%d S=%d T=%d %s graph=%d: %.2f ms per policy-gradient step, %.1f us per rollout time step (fwd+bwd), loss %.6g' % (N, S, T, DT, GRAPH, dt * 1e3, dt / T * 1e6, float(loss.detach())))
