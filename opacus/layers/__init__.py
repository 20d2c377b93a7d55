from dp_timeautodiff_amd.layers import DPGRU, DPLSTM, DPMultiheadAttention, RNNLinear, SequenceBias  # noqa: F401
