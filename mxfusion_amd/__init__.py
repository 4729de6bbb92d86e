"""mxfusion_amd -- MI355X-native Gaussian-process + SVI hot path behind MXFusion's Model / Posterior /
InferenceAlgorithm API (amzn/MXFusion v0.3.1).  Every array computation is a hand-written HIP kernel for gfx950
in libmxf_gp.so (C ABI: include/mxf_gp.h); there is no CPU fallback."""
__version__ = '0.1.0'

import os as _os

# The training step runs on three HIP streams (csrc/common.h: main + two side streams).  With the runtime's default of 4 hardware queues two
# of them land on one queue once torch's and RCCL's own streams exist, and the two Cholesky chains serialise (+1.2 ms per step at 4 samples
# per GPU, measured).  The flag is read when the HIP runtime initialises, i.e. at the first device call, so import this package (or export
# the variable) before touching the GPU.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

from .models import Model, Posterior  # noqa: F401,E402
from .components import Variable  # noqa: F401,E402
