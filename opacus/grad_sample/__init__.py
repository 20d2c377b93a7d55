from dp_timeautodiff_amd.grad_sample import GradSampleModule  # noqa: F401
