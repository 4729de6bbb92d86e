"""r05: is a SMALL step (the reference's svgp_regression notebook: N = 1000, M = 20, minibatches of 10; or one rank's share of a row-sharded
minibatch) paced by the host?  Enqueue time against wall time per step, eager, and the same step replayed as a hipGraph
(BatchInferenceLoop(use_graph=True) on a fixed batch).  usage: small_step.py [N M B Q]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
from mxfusion_amd import Model, Variable
from mxfusion_amd.components.variables import PositiveTransformation
from mxfusion_amd.components.distributions.gp.kernels import RBF
from mxfusion_amd.modules.gp_modules import SVGPRegression
from mxfusion_amd.inference import GradBasedInference, MAP, BatchInferenceLoop
from mxfusion_amd.inference.batch_loop import _Adam
N, M, B, Q = [int(a) for a in (sys.argv[1:5] + ['1000', '20', '10', '1'][len(sys.argv) - 1:])][:4]
for dtype in ('float32', 'float64'):
    for use_graph in (False, True):
        rng = np.random.RandomState(0)
        X = rng.rand(N, Q) * 6 - 3
        Y = np.sin(X[:, :1]) + 0.1 * rng.randn(N, 1)
        td = torch.float32 if dtype == 'float32' else torch.float64
        m = Model()
        m.N = Variable()
        m.X = Variable(shape=(m.N, Q))
        m.noise_var = Variable(shape=(1,), transformation=PositiveTransformation(), initial_value=0.01)
        m.kernel = RBF(input_dim=Q, ARD=True, variance=1., lengthscale=np.ones(Q), dtype=dtype)
        m.Y = SVGPRegression.define_variable(X=m.X, kernel=m.kernel, noise_var=m.noise_var, num_inducing=M, shape=(m.N, 1), dtype=dtype)
        m.Y.factor.svgp_log_pdf.jitter = 1e-6
        loop = BatchInferenceLoop(use_graph=use_graph)
        infr = GradBasedInference(MAP(model=m, observed=[m.X, m.Y]), grad_loop=loop, dtype=dtype)
        infr.initialize(X=(B, Q), Y=(B, 1))
        m.Y.factor.svgp_log_pdf.log_pdf_scaling = N / B
        Xd, Yd = torch.as_tensor(X[:B], dtype=td).cuda(), torch.as_tensor(Y[:B], dtype=td).cuda()
        ex = infr.create_executor()
        tr = _Adam(infr.params, 1e-2)
        for _ in range(6):
            loop.step(ex, [Xd, Yd], infr.params); tr.step(batch_size=B)
        torch.cuda.synchronize()
        K = 200
        t0 = time.perf_counter()
        for _ in range(K):
            loop.step(ex, [Xd, Yd], infr.params); tr.step(batch_size=B)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('N=%d M=%d B=%d Q=%d %s %s: host enqueue %.3f ms/step, wall %.3f ms/step' % (N, M, B, Q, dtype, 'hipGraph' if use_graph else 'eager   ',
                                                                                           (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3), flush=True)
