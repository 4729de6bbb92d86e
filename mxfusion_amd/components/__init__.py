from .variables import Variable, VariableType, PositiveTransformation  # noqa: F401
