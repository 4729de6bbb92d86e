"""Distribution base (mxfusion/components/distributions/distribution.py): a Factor whose output is a
random variable; log_pdf / draw_samples fetch their inputs from the runtime `variables` dict."""
from ..factor import Factor
from ..variables.variable import Variable
from ...common import config


class Distribution(Factor):
    def __init__(self, inputs, outputs, input_names, output_names, rand_gen=None, dtype=None, ctx=None):
        super(Distribution, self).__init__(inputs, outputs, input_names, output_names)
        from .random_gen import TorchRandomGenerator
        self._rand_gen = TorchRandomGenerator if rand_gen is None else rand_gen
        self.dtype = dtype
        self.ctx = ctx
        self.log_pdf_scaling = 1

    @staticmethod
    def _as_variable(v):
        return v if isinstance(v, Variable) else Variable(value=v)

    def _generate_outputs(self, shape):
        self._outputs = [('random_variable', Variable(value=None, shape=shape))]
        self._output_names = ['random_variable']
        self._outputs[0][1].factor = self

    @property
    def random_variable(self):
        return self._outputs[0][1]

    def _fetch(self, variables):
        return {n: variables[v.uuid] for n, v in self.inputs}

    def log_pdf(self, F, variables, targets=None):
        kw = self._fetch(variables)
        kw['random_variable'] = variables[self.random_variable.uuid]
        return self.log_pdf_impl(F=F, **kw)

    def draw_samples(self, F, variables, num_samples=1, targets=None, always_return_tuple=False):
        kw = self._fetch(variables)
        from ...util.inference import realize_shape
        rv_shape = realize_shape(self.random_variable.shape, variables)     # symbolic dims (m.N) are bound constants
        out = self.draw_samples_impl(rv_shape=rv_shape, num_samples=num_samples, F=F, **kw)
        return (out,) if always_return_tuple else out

    def torch_dtype(self):
        return config.torch_dtype(self.dtype)
