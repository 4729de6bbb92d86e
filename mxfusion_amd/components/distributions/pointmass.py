"""PointMass placeholder used by MAP (mxfusion/inference/map.py:56-59)."""
from .distribution import Distribution
from ..variables.variable import Variable


class PointMass(Distribution):
    def __init__(self, location, rand_gen=None, dtype=None, ctx=None):
        location = location if isinstance(location, Variable) else Variable(value=location)
        super(PointMass, self).__init__([('location', location)], None, ['location'], ['random_variable'], rand_gen, dtype, ctx)

    def log_pdf_impl(self, location, random_variable, F=None):
        return 0.

    def draw_samples_impl(self, location, rv_shape, num_samples=1, F=None):
        return location.expand((num_samples,) + tuple(location.shape[1:]))
