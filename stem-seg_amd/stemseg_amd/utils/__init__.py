from .global_registry import GlobalRegistry  # noqa: F401
