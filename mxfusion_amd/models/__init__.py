from .factor_graph import FactorGraph  # noqa: F401
from .model import Model  # noqa: F401
from .posterior import Posterior  # noqa: F401
