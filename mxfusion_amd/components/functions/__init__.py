from .function_evaluation import FunctionEvaluation, MXFusionFunction  # noqa: F401
