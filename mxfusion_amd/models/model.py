"""Model (mxfusion/models/model.py): the generative FactorGraph."""
from .factor_graph import FactorGraph


class Model(FactorGraph):
    def __init__(self, name='model', verbose=False):
        super(Model, self).__init__(name=name, verbose=verbose)
