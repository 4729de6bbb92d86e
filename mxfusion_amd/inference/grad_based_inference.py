"""GradBasedInference (mxfusion/inference/grad_based_inference.py:22-140)."""
import torch

from ..common import config
from .batch_loop import BatchInferenceLoop
from .inference import Inference, discover_shape_constants
from .minibatch_loop import MinibatchInferenceLoop


class GradBasedInference(Inference):
    def __init__(self, inference_algorithm, grad_loop=None, constants=None, hybridize=False, dtype=None, context=None):
        if grad_loop is None:
            grad_loop = BatchInferenceLoop()
        super(GradBasedInference, self).__init__(inference_algorithm=inference_algorithm, constants=constants, hybridize=hybridize,
                                                 dtype=dtype, context=context)
        self._grad_loop = grad_loop

    def create_executor(self):
        rv_scaling = getattr(self._grad_loop, 'rv_scaling', None)          # (the minibatch loops; the row-sharded batch loop: scaling 1)
        gw = getattr(self._grad_loop, 'global_weight', None)
        gw = gw() if callable(gw) else gw                                  # 1 / world size for a row-sharded loop, else None
        return self._inference_algorithm.create_executor(data_def=self.observed_variable_UUIDs, params=self.params,
                                                         var_ties=self.params.var_ties, rv_scaling=rv_scaling, global_weight=gw)

    def run(self, optimizer='adam', learning_rate=1e-3, max_iter=2000, verbose=False, permutations=None, generator=None, **kwargs):
        """grad_based_inference.py:73-104.  `permutations` / `generator`: the minibatch loop's shuffle seam (one index sequence per epoch /
        a torch.Generator for torch.randperm) -- the reference's DataLoader shuffle is MXNet-RNG driven."""
        data = [self._to_device(kwargs[v]) for v in self.observed_variable_names]
        self.initialize(**kwargs)
        infr = self.create_executor()
        if isinstance(self._grad_loop, MinibatchInferenceLoop):
            def update_shape_constants(data_batch):
                shapes = {i: tuple(d.shape) for i, d in zip(self.observed_variable_UUIDs, data_batch)}
                self.params.update_constants(discover_shape_constants(shapes, self._graphs))
            out = self._grad_loop.run(infr_executor=infr, data=data, param_dict=self.params, ctx=self.mxnet_context,
                                      optimizer=optimizer, learning_rate=learning_rate, max_iter=max_iter, verbose=verbose,
                                      update_shape_constants=update_shape_constants, generator=generator, permutations=permutations)
        else:
            if hasattr(self._grad_loop, 'bind_data'):
                self._grad_loop.bind_data(self.observed_variable_UUIDs)
            local = self._grad_loop._local(data) if hasattr(self._grad_loop, '_local') else data
            restore = None
            if local is not data:          # row-sharded batch loop: the executor sees this rank's rows -- shape constants (N) follow them
                shapes = {i: tuple(d.shape) for i, d in zip(self.observed_variable_UUIDs, local)}
                full = discover_shape_constants({i: tuple(d.shape) for i, d in zip(self.observed_variable_UUIDs, data)}, self._graphs)
                restore = full
                self.params.update_constants(discover_shape_constants(shapes, self._graphs))
            try:
                out = self._grad_loop.run(infr_executor=infr, data=data, param_dict=self.params, ctx=self.mxnet_context,
                                          optimizer=optimizer, learning_rate=learning_rate, max_iter=max_iter, verbose=verbose)
            finally:
                if restore is not None:    # predictions / checkpoints seeded from infr.params see the FULL data's shape constants again
                    self.params.update_constants(restore)
        self._check_float32_validity()
        return out

    F32_COND_LIMIT = 1e3

    def _check_float32_validity(self):
        """The float32 forms of the SVGP bound are valid up to a condition number of Kuu + jitter I each (explicit inverse: 3e3, whitened:
        5e6; modules/gp_modules/_fused.py: Float32Guard moves every SVGP module between them and float64 by itself while it runs).  This
        records what happened for the caller: `last_kuu_condition` (largest condition number the run's calls published),
        `float32_fallback_active` (some module ended on float64), `float32_tiers` (the level of every module)."""
        if config.torch_dtype(self.dtype) != torch.float32 or not torch.cuda.is_available():
            return
        from ..modules.gp_modules.svgp_regression import SVGPRegression
        if not any(isinstance(f, SVGPRegression) for g in self._graphs for f in getattr(g, '_factors', [])):
            return
        ctx = self.mxnet_context
        if isinstance(ctx, str):
            ctx = torch.device(ctx)
        if isinstance(ctx, torch.device) and ctx.type != 'cuda':
            return
        from ..modules.gp_modules._fused import Float32Guard
        try:                                   # a diagnostic: it must never fail a finished run
            torch.cuda.synchronize()
            rep = Float32Guard.report(ctx)
        except Exception:                      # noqa: BLE001
            return
        cond = rep['kuu_cond_max']
        self.last_kuu_condition = cond
        self.float32_fallback_active = bool(rep['float32_fallback_active'])
        self.float32_tiers = rep['float32_tiers']
        if cond > self.F32_COND_LIMIT and not Float32Guard.enabled:
            import warnings
            warnings.warn('mxfusion_amd: cond_1(Kuu + jitter I) = %.2e after this float32 run: beyond ~%.0e the explicit-inverse float32 SVGP '
                          'bound loses accuracy (error ~ cond * 2^-24) and the automatic switch to the whitened / float64 forms is disabled; '
                          'enable Float32Guard or run the inference with dtype=\'float64\'.' % (cond, self.F32_COND_LIMIT))


class GradTransferInference(GradBasedInference):
    """grad_based_inference.py:106-140: gradient-based optimisation of EXTERNAL parameters (`train_params`, e.g. a policy network)
    through an algorithm whose model parameters are inherited from a finished inference and held fixed.

    The reference stores `train_params` but hands only its own (all-fixed) ParameterDict to the Trainer (grad_based_inference.py:130,
    100-104), so there nothing is ever updated; here `train_params` are what the loop optimises, which is what the PILCO example needs."""

    def __init__(self, inference_algorithm, infr_params, train_params, grad_loop=None, var_tie=None, constants=None, hybridize=False,
                 dtype=None, context=None):
        self._var_tie = var_tie if var_tie is not None else {}
        self._inherited_params = infr_params
        self.train_params = train_params
        if dtype is None:
            dtype = infr_params.dtype
        super(GradTransferInference, self).__init__(inference_algorithm=inference_algorithm, grad_loop=grad_loop, constants=constants,
                                                    hybridize=hybridize, dtype=dtype, context=context)

    def _initialize_params(self):
        self.params.initialize_params(self._graphs, self.observed_variable_UUIDs, carry=self._inherited_params.export_raw())
        self.params.fix_all()
        self.params.set_train_params(self.train_params)
