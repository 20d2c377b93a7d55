from . import batch_memory_manager  # noqa: F401
