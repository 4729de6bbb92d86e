"""r05: where does the HOST time of a small eager step go?  cProfile of 200 steps of the notebook-sized SVGP MAP model (N = 1000, M = 20, B = 10)."""
import os, sys, cProfile, pstats
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from mxfusion_amd import Model, Variable
from mxfusion_amd.components.variables import PositiveTransformation
from mxfusion_amd.components.distributions.gp.kernels import RBF
from mxfusion_amd.modules.gp_modules import SVGPRegression
from mxfusion_amd.inference import GradBasedInference, MAP, BatchInferenceLoop
from mxfusion_amd.inference.batch_loop import _Adam
dtype = sys.argv[1] if len(sys.argv) > 1 else 'float32'
N, M, B, Q = 1000, 20, 10, 1
rng = np.random.RandomState(0)
X = rng.rand(N, Q) * 6 - 3; Y = np.sin(X[:, :1]) + 0.1 * rng.randn(N, 1)
td = torch.float32 if dtype == 'float32' else torch.float64
m = Model(); m.N = Variable(); m.X = Variable(shape=(m.N, Q))
m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
m.kernel = RBF(input_dim=Q, ARD=True, variance=1., lengthscale=np.ones(Q), dtype=dtype)
m.Y = SVGPRegression.define_variable(X=m.X, kernel=m.kernel, noise_var=m.noise_var, num_inducing=M, shape=(m.N, 1), dtype=dtype)
m.Y.factor.svgp_log_pdf.jitter = 1e-6
loop = BatchInferenceLoop()
infr = GradBasedInference(MAP(model=m, observed=[m.X, m.Y]), grad_loop=loop, dtype=dtype)
infr.initialize(X=(B, Q), Y=(B, 1))
Xd, Yd = torch.as_tensor(X[:B], dtype=td).cuda(), torch.as_tensor(Y[:B], dtype=td).cuda()
ex = infr.create_executor(); tr = _Adam(infr.params, 1e-2)
for _ in range(10):
    loop.step(ex, [Xd, Yd], infr.params); tr.step(batch_size=B)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    loop.step(ex, [Xd, Yd], infr.params); tr.step(batch_size=B)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(35)
