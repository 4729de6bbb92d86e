"""mxfusion_amd -- MI355X-native Gaussian-process + SVI hot path behind MXFusion's Model / Posterior /
InferenceAlgorithm API (amzn/MXFusion v0.3.1).  Every array computation is a hand-written HIP kernel for gfx950
in libmxf_gp.so (C ABI: include/mxf_gp.h); there is no CPU fallback."""
__version__ = '0.1.0'

from .models import Model, Posterior  # noqa: F401,E402
from .components import Variable  # noqa: F401,E402
