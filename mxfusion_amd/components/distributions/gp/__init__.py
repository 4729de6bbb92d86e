from .gp import GaussianProcess  # noqa: F401
from .cond_gp import ConditionalGaussianProcess  # noqa: F401
