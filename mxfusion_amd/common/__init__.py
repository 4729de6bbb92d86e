from . import config  # noqa: F401
from .exceptions import ModelSpecificationError, InferenceError, SerializationError  # noqa: F401
