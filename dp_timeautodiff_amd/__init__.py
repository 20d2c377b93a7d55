"""dp_timeautodiff_amd — MI355X-native DP-SGD step behind the Opacus-shaped API that
ammanyusuf/dp-timeautodiff calls (Model_Code/timeautoencoder.py:18-21, 653-759).

Compute runs only in the hand-written HIP kernels of libdpk_gfx950.so (include/dpk.h);
there is no CPU fallback.
"""
__version__ = "0.1.0"
