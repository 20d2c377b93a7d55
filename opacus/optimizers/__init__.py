from dp_timeautodiff_amd.optimizer import DistributedDPOptimizer, DPOptimizer  # noqa: F401

__all__ = ["DPOptimizer", "DistributedDPOptimizer"]
