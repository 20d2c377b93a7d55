from dp_timeautodiff_amd.batch_memory_manager import (BatchMemoryManager, BatchSplittingSampler,  # noqa: F401
                                                      wrap_data_loader)
