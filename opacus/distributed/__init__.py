from dp_timeautodiff_amd.distributed import DifferentiallyPrivateDistributedDataParallel  # noqa: F401
