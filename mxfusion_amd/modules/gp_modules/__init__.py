from .gp_regression import GPRegression  # noqa: F401
from .svgp_regression import SVGPRegression  # noqa: F401
from .sparsegp_regression import SparseGPRegression  # noqa: F401
