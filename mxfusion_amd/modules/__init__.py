from .module import Module  # noqa: F401
