from .variable import Variable, VariableType  # noqa: F401
from .var_trans import PositiveTransformation, Softplus  # noqa: F401
from .runtime_variable import add_sample_dimension, add_sample_dimension_to_arrays, expectation, \
    array_has_samples, get_num_samples, as_samples, arrays_as_samples  # noqa: F401
