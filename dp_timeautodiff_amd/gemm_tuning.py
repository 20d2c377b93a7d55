"""Library-GEMM selection for the model's own forward/backward GEMMs (the callers either side of the DP path).

The reference model's Linear layers and the two GEMMs around the fused GRU recurrence are ordinary fp32 GEMMs that
PyTorch hands to hipBLASLt / rocBLAS.  Their default heuristics choose poor kernels for this workload's skinny shapes
([B*T, 128] x [128, 384] ran 43 us against 30 us for the best solution).  PyTorch's TunableOp times the candidate
solutions the first time a shape is seen and keeps the winner; with the small budgets used here the whole model tunes
in a few seconds during the warm-up steps.  Tuning is switched off again before a step is captured into a hipGraph.
"""
from __future__ import annotations

import torch


def enable(max_ms_per_candidate: int = 30, max_iterations: int = 10, results_file: str | None = None) -> bool:
    """Turn GEMM tuning on for this process.  Returns False when the build has no TunableOp."""
    tun = getattr(torch.cuda, "tunable", None)
    if tun is None or not torch.cuda.is_available():
        return False
    tun.enable(True)
    tun.tuning_enable(True)
    tun.set_max_tuning_duration(int(max_ms_per_candidate))
    tun.set_max_tuning_iterations(int(max_iterations))
    if results_file:
        tun.set_filename(results_file)
    else:
        import os
        import tempfile
        # results are only a cache of this process's choices: keep them out of the working directory
        tun.set_filename(os.path.join(tempfile.gettempdir(), f"dpk_tunableop_{os.getpid()}.csv"))
    return True


def freeze() -> None:
    """Stop tuning new shapes (already chosen solutions stay in use).  Called before hipGraph capture: a tuning run
    launches and times kernels on its own, which a capturing stream does not allow."""
    tun = getattr(torch.cuda, "tunable", None)
    if tun is not None and torch.cuda.is_available() and tun.is_enabled():
        tun.tuning_enable(False)
