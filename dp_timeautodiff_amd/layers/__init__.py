from .dp_rnn import DPGRU, DPLSTM, DPRNNBase, RNNLinear
from .dp_multihead_attention import DPMultiheadAttention, SequenceBias

__all__ = ["DPGRU", "DPLSTM", "DPRNNBase", "RNNLinear", "DPMultiheadAttention", "SequenceBias"]
from .fused_gru import FusedLinearCellGRU, fuse_recurrences, run_together
