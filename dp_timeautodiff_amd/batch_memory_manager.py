"""BatchMemoryManager: split a Poisson (logical) batch into physical chunks of bounded size.

Behavioural restatement of `opacus.utils.batch_memory_manager` as used by the
reference at Model_Code/timeautoencoder.py:711-715 and :726-730 (max_physical_batch_size=128).  All but
the last chunk of a logical batch signal the optimizer to *skip* the step: it clips and accumulates the
chunk's per-sample gradients (K6 + K7a) and returns; the last chunk adds noise, scales and steps.  On
MI355X the same mechanism bounds the per-sample-gradient arena for BASELINE.json's long-horizon config.
"""
from __future__ import annotations

import math
from typing import List

import numpy as np
from torch.utils.data import BatchSampler, DataLoader, Sampler

from .data_loader import (DevicePoissonLoader, DistributedUniformWithReplacementSampler,
                          UniformWithReplacementSampler)


class BatchSplittingSampler(Sampler[List[int]]):
    def __init__(self, *, sampler: Sampler[List[int]], max_batch_size: int, optimizer):
        self.sampler = sampler
        self.max_batch_size = max_batch_size
        self.optimizer = optimizer

    def __iter__(self):
        for batch_idxs in self.sampler:
            if len(batch_idxs) == 0:
                self.optimizer.signal_skip_step(do_skip=False)
                yield []
                continue
            split_idxs = np.array_split(batch_idxs, math.ceil(len(batch_idxs) / self.max_batch_size))
            split_idxs = [s.tolist() for s in split_idxs]
            for x in split_idxs[:-1]:
                self.optimizer.signal_skip_step(do_skip=True)
                yield x
            self.optimizer.signal_skip_step(do_skip=False)
            yield split_idxs[-1]

    def __len__(self):
        if isinstance(self.sampler, BatchSampler):
            return math.ceil(len(self.sampler) * (self.sampler.batch_size / self.max_batch_size))
        elif isinstance(self.sampler, (UniformWithReplacementSampler, DistributedUniformWithReplacementSampler)):
            expected_batch_size = self.sampler.sample_rate * self.sampler.num_samples
            return math.ceil(len(self.sampler) * (expected_batch_size / self.max_batch_size))
        return len(self.sampler)


class _DeviceChunker:
    """BatchMemoryManager over a DevicePoissonLoader: each logical (Poisson) batch already sits in HBM and is cut
    into views of at most `max_batch_size` rows; all but the last chunk signal a skipped optimizer step."""

    def __init__(self, loader: DevicePoissonLoader, max_batch_size: int, optimizer):
        self.loader, self.max_batch_size, self.optimizer = loader, max_batch_size, optimizer

    def __len__(self):
        return len(self.loader)

    @property
    def dataset(self):
        return self.loader.dataset

    def __iter__(self):
        for batch in self.loader:
            n = batch[0].shape[0]
            bounds = list(range(0, n, self.max_batch_size)) or [0]
            for i, lo in enumerate(bounds):
                self.optimizer.signal_skip_step(do_skip=i < len(bounds) - 1)
                yield [t[lo:lo + self.max_batch_size] for t in batch]


def wrap_data_loader(*, data_loader: DataLoader, max_batch_size: int, optimizer):
    if isinstance(data_loader, DevicePoissonLoader):
        return _DeviceChunker(data_loader, max_batch_size, optimizer)
    return DataLoader(
        dataset=data_loader.dataset,
        batch_sampler=BatchSplittingSampler(sampler=data_loader.batch_sampler, max_batch_size=max_batch_size,
                                            optimizer=optimizer),
        num_workers=data_loader.num_workers, collate_fn=data_loader.collate_fn, pin_memory=data_loader.pin_memory,
        timeout=data_loader.timeout, worker_init_fn=data_loader.worker_init_fn,
        multiprocessing_context=data_loader.multiprocessing_context, generator=data_loader.generator,
        prefetch_factor=data_loader.prefetch_factor, persistent_workers=data_loader.persistent_workers)


class BatchMemoryManager(object):
    def __init__(self, *, data_loader: DataLoader, max_physical_batch_size: int, optimizer):
        self.data_loader = data_loader
        self.optimizer = optimizer
        self.max_physical_batch_size = max_physical_batch_size

    def __enter__(self):
        return wrap_data_loader(data_loader=self.data_loader, max_batch_size=self.max_physical_batch_size,
                                optimizer=self.optimizer)

    def __exit__(self, type, value, traceback):
        pass
