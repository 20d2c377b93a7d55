from dp_timeautodiff_amd.data_loader import DPDataLoader, switch_generator  # noqa: F401
