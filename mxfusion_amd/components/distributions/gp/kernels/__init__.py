from .kernel import Kernel, NativeKernel, CombinationKernel, AddKernel, MultiplyKernel  # noqa: F401
from .stationary import StationaryKernel, RBF, Matern12, Matern32, Matern52  # noqa: F401
from .static import Linear, Bias, White  # noqa: F401
