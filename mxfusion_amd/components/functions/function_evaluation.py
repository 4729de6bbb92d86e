"""Deterministic functions inside a model (mxfusion/components/functions/*): only what the GP path needs --
an optional mean function m(X) evaluated before the module (gp_regression.py:63-65).  The callable receives
and returns arrays carrying the sample axis."""
from ..factor import Factor
from ..variables.variable import Variable


class FunctionEvaluation(Factor):
    def __init__(self, func, inputs, num_outputs=1):
        outs = [('output_%d' % i, Variable(shape=None)) for i in range(num_outputs)]
        super(FunctionEvaluation, self).__init__(inputs, outs, [n for n, _ in inputs], [n for n, _ in outs])
        self._func = func

    def eval(self, F, variables, always_return_tuple=False):
        out = self._func(*[variables[v.uuid] for _, v in self.inputs])
        if not isinstance(out, (tuple, list)):
            out = (out,)
        return tuple(out) if always_return_tuple or len(out) > 1 else out[0]


class MXFusionFunction(object):
    """m.mean_func = MXFusionFunction(callable); m.mean = m.mean_func(m.X)   (cf. MXFusionGluonFunction)."""

    def __init__(self, func, num_outputs=1, name='func'):
        self._func, self._num_outputs, self.name = func, num_outputs, name

    def __call__(self, *args):
        fe = FunctionEvaluation(self._func, [('arg_%d' % i, a) for i, a in enumerate(args)], self._num_outputs)
        outs = [v for _, v in fe.outputs]
        return outs[0] if len(outs) == 1 else tuple(outs)
