from .distribution import Distribution  # noqa: F401
from .normal import Normal  # noqa: F401
from .pointmass import PointMass  # noqa: F401
from .random_gen import RandomGenerator, TorchRandomGenerator, MockRandomGenerator  # noqa: F401
from .gp import GaussianProcess, ConditionalGaussianProcess  # noqa: F401
