"""mxfusion/common/exceptions.py:16-25."""


class ModelSpecificationError(Exception):
    pass


class InferenceError(Exception):
    pass


class SerializationError(Exception):
    pass
