from dp_timeautodiff_amd.validators import ModuleValidator  # noqa: F401

__all__ = ["ModuleValidator"]
