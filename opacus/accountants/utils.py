from dp_timeautodiff_amd.accountants.utils import get_noise_multiplier  # noqa: F401
