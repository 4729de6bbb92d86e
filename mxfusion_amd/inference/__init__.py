from .inference import Inference, TransferInference  # noqa: F401
from .grad_based_inference import GradBasedInference, GradTransferInference  # noqa: F401
from .inference_alg import InferenceAlgorithm, SamplingAlgorithm  # noqa: F401
from .variational import VariationalInference, StochasticVariationalInference  # noqa: F401
from .map import MAP  # noqa: F401
from .meanfield import create_Gaussian_meanfield  # noqa: F401
from .batch_loop import BatchInferenceLoop, DistributedBatchInferenceLoop  # noqa: F401
from .minibatch_loop import MinibatchInferenceLoop, DistributedMinibatchInferenceLoop  # noqa: F401
from .prediction import ModulePredictionAlgorithm  # noqa: F401
from .forward_sampling import ForwardSamplingAlgorithm, ForwardSampling, VariationalPosteriorForwardSampling  # noqa: F401
from .inference_parameters import InferenceParameters  # noqa: F401
from .expectation import ExpectationAlgorithm  # noqa: F401
from .pilco_alg import PILCOAlgorithm  # noqa: F401
