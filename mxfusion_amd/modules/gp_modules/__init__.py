from .gp_regression import GPRegression  # noqa: F401
from .svgp_regression import SVGPRegression  # noqa: F401
