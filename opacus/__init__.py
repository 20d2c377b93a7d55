"""`opacus` alias: makes the reference's own import lines resolve to the MI355X-native implementation.

    from opacus import PrivacyEngine                                  # Model_Code/timeautoencoder.py:18
    from opacus.validators import ModuleValidator                     # :19
    from opacus.accountants.utils import get_noise_multiplier         # :20
    from opacus.utils.batch_memory_manager import BatchMemoryManager  # :21

Put the repository root on PYTHONPATH ahead of site-packages and `dp_job.sh` runs on the HIP kernels
with no source change.
"""
from dp_timeautodiff_amd.grad_sample import GradSampleModule
from dp_timeautodiff_amd.privacy_engine import PrivacyEngine

from . import accountants, optimizers, utils, validators  # noqa: F401

__version__ = "1.5.2+dpk"
__all__ = ["PrivacyEngine", "GradSampleModule"]
