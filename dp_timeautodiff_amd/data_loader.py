"""DPDataLoader (Poisson sampling) and its samplers.

Behavioural restatement of Opacus' `opacus.data_loader.DPDataLoader` and
`opacus.utils.uniform_sampler`: the object
`PrivacyEngine.make_private_with_epsilon` hands back in place of the reference's shuffled
DataLoader (Model_Code/timeautoencoder.py:664-682).  Host-side only.
"""
from __future__ import annotations

import logging
from functools import partial
from typing import Any, List, Optional, Sequence, Tuple, Type, Union

import torch
from torch.utils.data import BatchSampler, DataLoader, Dataset, IterableDataset, Sampler
from torch.utils.data._utils.collate import default_collate
from torch.utils.data.dataloader import _collate_fn_t

logger = logging.getLogger(__name__)


class UniformWithReplacementSampler(Sampler[List[int]]):
    """Each sample is included in a batch independently with probability `sample_rate`."""

    def __init__(self, *, num_samples: int, sample_rate: float, generator=None, steps=None):
        self.num_samples = num_samples
        self.sample_rate = sample_rate
        self.generator = generator
        if self.num_samples <= 0:
            raise ValueError("num_samples should be a positive integer value, but got num_samples={}".format(
                self.num_samples))
        if steps is not None:
            self.steps = steps
        else:
            self.steps = int(1 / self.sample_rate)

    def __len__(self):
        return self.steps

    def __iter__(self):
        num_batches = self.steps
        while num_batches > 0:
            mask = torch.rand(self.num_samples, generator=self.generator) < self.sample_rate
            indices = mask.nonzero(as_tuple=False).reshape(-1).tolist()
            yield indices
            num_batches -= 1


class DistributedUniformWithReplacementSampler(Sampler[List[int]]):
    """Poisson sampling over this rank's share of a per-epoch seeded permutation (A.7)."""

    def __init__(self, *, total_size: int, sample_rate: float, shuffle: bool = True, shuffle_seed: int = 0,
                 steps: Optional[int] = None, generator=None):
        self.total_size = total_size
        self.sample_rate = sample_rate
        self.generator = generator
        self.num_replicas = torch.distributed.get_world_size()
        self.rank = torch.distributed.get_rank()
        self.epoch = 0
        self.shuffle = shuffle
        self.shuffle_seed = shuffle_seed
        if self.total_size <= 0:
            raise ValueError("total_size should be a positive integer value, but got total_size={}".format(
                self.total_size))
        self.num_samples = self.total_size // self.num_replicas
        if self.rank < self.total_size % self.num_replicas:
            self.num_samples += 1
        if steps is not None:
            self.num_batches = steps
        else:
            self.num_batches = int(1 / self.sample_rate)

    def __len__(self):
        return self.num_batches

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.shuffle_seed + self.epoch)
            indices = torch.randperm(self.total_size, generator=g)
        else:
            indices = torch.arange(self.total_size)
        indices = indices[self.rank:self.total_size:self.num_replicas]
        assert len(indices) == self.num_samples
        for _ in range(self.num_batches):
            mask = torch.rand(self.num_samples, generator=self.generator) < self.sample_rate
            selected_examples = mask.nonzero(as_tuple=False).reshape(-1)
            if len(selected_examples) > 0:
                yield indices[selected_examples].tolist()
            else:
                yield []

    def set_epoch(self, epoch: int):
        self.epoch = epoch


def shape_safe(x: Any) -> Tuple:
    return getattr(x, "shape", ())


def dtype_safe(x: Any) -> Union[torch.dtype, Type]:
    return getattr(x, "dtype", type(x))


def collate(batch: List[torch.Tensor], *, collate_fn: Optional[_collate_fn_t], sample_empty_shapes: Sequence[Tuple],
            dtypes: Sequence[Union[torch.dtype, Type]]):
    """Collate that tolerates the empty batches Poisson sampling produces (zero-length tensors)."""
    if len(batch) > 0:
        return collate_fn(batch)
    return [torch.zeros(shape, dtype=dtype) for shape, dtype in zip(sample_empty_shapes, dtypes)]


def wrap_collate_with_empty(*, collate_fn: Optional[_collate_fn_t], sample_empty_shapes, dtypes):
    return partial(collate, collate_fn=collate_fn, sample_empty_shapes=sample_empty_shapes, dtypes=dtypes)


class DPDataLoader(DataLoader):
    """DataLoader whose batches are Poisson samples with rate `sample_rate` (logical batch size is random)."""

    def __init__(self, dataset: Dataset, *, sample_rate: float, collate_fn: Optional[_collate_fn_t] = None,
                 drop_last: bool = False, generator=None, distributed: bool = False, **kwargs):
        self.sample_rate = sample_rate
        self.distributed = distributed
        if distributed:
            batch_sampler = DistributedUniformWithReplacementSampler(
                total_size=len(dataset), sample_rate=sample_rate, generator=generator)
        else:
            batch_sampler = UniformWithReplacementSampler(
                num_samples=len(dataset), sample_rate=sample_rate, generator=generator)
        sample_empty_shapes = [(0, *shape_safe(x)) for x in dataset[0]]
        dtypes = [dtype_safe(x) for x in dataset[0]]
        if collate_fn is None:
            collate_fn = default_collate
        if drop_last:
            logger.warning("Ignoring drop_last as it is not compatible with DPDataLoader.")
        super().__init__(
            dataset=dataset, batch_sampler=batch_sampler,
            collate_fn=wrap_collate_with_empty(collate_fn=collate_fn, sample_empty_shapes=sample_empty_shapes,
                                               dtypes=dtypes),
            generator=generator, **kwargs)

    @classmethod
    def from_data_loader(cls, data_loader: DataLoader, *, distributed: bool = False, generator=None):
        if isinstance(data_loader.dataset, IterableDataset):
            raise ValueError("Uniform sampling is not supported for IterableDataset")
        return cls(
            dataset=data_loader.dataset, sample_rate=1 / len(data_loader), num_workers=data_loader.num_workers,
            collate_fn=data_loader.collate_fn, pin_memory=data_loader.pin_memory, drop_last=data_loader.drop_last,
            timeout=data_loader.timeout, worker_init_fn=data_loader.worker_init_fn,
            multiprocessing_context=data_loader.multiprocessing_context,
            generator=generator if generator else data_loader.generator,
            prefetch_factor=data_loader.prefetch_factor, persistent_workers=data_loader.persistent_workers,
            distributed=distributed)


def _is_supported_batch_sampler(sampler: Sampler):
    return isinstance(sampler, (BatchSampler, UniformWithReplacementSampler,
                                DistributedUniformWithReplacementSampler))


def switch_generator(*, data_loader: DataLoader, generator):
    """Re-create `data_loader` with a different random generator (secure_mode / noise_generator plumbing)."""
    batch_sampler = data_loader.batch_sampler
    if batch_sampler is None or not _is_supported_batch_sampler(batch_sampler):
        raise ValueError("Non-batch processing is not supported: Opacus always assumes one of the input dimensions "
                         "to be batch dimension.")
    if isinstance(batch_sampler, BatchSampler):
        if not hasattr(batch_sampler.sampler, "generator"):
            raise ValueError("Target sampler doesn't have generator attribute: nothing to switch")
        batch_sampler.sampler.generator = generator
    else:
        batch_sampler.generator = generator
    return DataLoader(
        dataset=data_loader.dataset, batch_sampler=batch_sampler, num_workers=data_loader.num_workers,
        collate_fn=data_loader.collate_fn, pin_memory=data_loader.pin_memory, drop_last=data_loader.drop_last,
        timeout=data_loader.timeout, worker_init_fn=data_loader.worker_init_fn,
        multiprocessing_context=data_loader.multiprocessing_context, generator=generator,
        prefetch_factor=data_loader.prefetch_factor, persistent_workers=data_loader.persistent_workers)


class PoissonCapacityError(RuntimeError):
    """A Poisson draw selected more samples than a fixed-capacity batch holds.  The samples beyond the capacity would
    be dropped, i.e. the batch would no longer be a Poisson sample and the sampled-Gaussian accounting (every epsilon
    the accountant reports) would not cover the run: an error, never a warning."""


class DevicePoissonLoader:
    """Poisson-sampled batches drawn, compacted and gathered on the GPU (K11).

    Stands where `DPDataLoader.from_data_loader(...)` stands for a `TensorDataset` (what the reference builds at
    Model_Code/timeautoencoder.py:664-670): the dataset tensors live in HBM, each step `dpk_poisson_sample` selects
    every sample independently with probability `sample_rate` from a Philox stream and `index_select` gathers the
    rows — no per-sample Python collation, no host-to-device copy.

    Two ways to consume it:
      * `for batch in loader:` — drop-in for the reference loop: yields a list of tensors with exactly the n
        selected rows (n is read back, one host sync per step), `len(loader)` = int(1 / sample_rate) steps;
      * `loader.sample_static()` — shape-static: returns the SAME `[capacity, ...]` tensors every call plus a device
        int32 `n_valid`; rows >= n_valid are padding (copies of sample 0) that `DPOptimizer.set_valid_rows(n_valid)`
        masks out with clip factor 0.  This is the form a hipGraph-captured step replays.
    `capacity` defaults to expected + 8 sigma + 8 (overflow probability < 1e-14 per step).  A draw beyond it is never
    silently truncated: the iterator form grows the capacity and repeats the SAME draw (same Philox position), the
    shape-static form counts such draws on the device and `check_overflow()` — one host sync; `train.py` calls it once
    per epoch, before epsilon is reported — raises `PoissonCapacityError`.
    """

    def __init__(self, tensors, *, sample_rate: float, device=None, capacity: Optional[int] = None,
                 seed: Optional[int] = None, steps: Optional[int] = None, rank: int = 0, world_size: int = 1):
        from . import _native
        self._native = _native
        tensors = list(tensors)
        if world_size > 1:
            # data-parallel: rank r owns samples r, r+W, r+2W, ... and draws each of them with the same rate from
            # its own Philox key.  Independent Bernoulli draws over a fixed partition are exactly a Poisson sample
            # of the whole dataset (Opacus' DistributedUniformWithReplacementSampler reshuffles the partition every
            # epoch, which changes nothing in distribution), and no index ever crosses ranks.
            tensors = [t[rank::world_size] for t in tensors]
        self.rank, self.world_size = rank, world_size
        if not tensors or any(t.shape[0] != tensors[0].shape[0] for t in tensors):
            raise ValueError("DevicePoissonLoader needs tensors with a common first dimension")
        device = torch.device(device) if device is not None else tensors[0].device
        if device.type != "cuda":
            raise _native.DpkError("DevicePoissonLoader samples on a HIP device; there is no CPU path")
        self.tensors = [t.to(device).contiguous() for t in tensors]
        self.num_samples = self.tensors[0].shape[0]
        self.sample_rate = float(sample_rate)
        self.steps = steps if steps is not None else int(1 / self.sample_rate)
        exp = self.num_samples * self.sample_rate
        if capacity is None:
            capacity = int(exp + 8.0 * (exp * (1.0 - self.sample_rate)) ** 0.5 + 8)
        self.capacity = max(1, min(int(capacity), self.num_samples))
        import os
        self._graph_captured = False
        self._seed = int.from_bytes(os.urandom(8), "little") if seed is None else int(seed)  # the job's seed
        self.device = device
        self._blocks_per_draw = (self.num_samples + 3) // 4
        self._offset_dev = torch.zeros(1, dtype=torch.int64, device=device)  # advanced on device: graph-replayable
        self.counts = torch.zeros(2, dtype=torch.int32, device=device)
        self.n_valid = self.counts[0:1]
        self._overflow = torch.zeros(1, dtype=torch.int64, device=device)
        self._allocate(self.capacity)
        self.dataset = torch.utils.data.TensorDataset(*self.tensors)  # PrivacyEngine reads len(loader.dataset)

    def _allocate(self, capacity: int):
        self.capacity = max(1, min(int(capacity), self.num_samples))
        self.idx = torch.zeros(self.capacity, dtype=torch.int64, device=self.device)
        self._static = [torch.empty((self.capacity,) + tuple(t.shape[1:]), dtype=t.dtype, device=self.device)
                        for t in self.tensors]

    @classmethod
    def from_data_loader(cls, data_loader: DataLoader, *, device, seed: Optional[int] = None, capacity=None,
                         rank: int = 0, world_size: int = 1):
        ds = data_loader.dataset
        if not isinstance(ds, torch.utils.data.TensorDataset):
            raise ValueError("DevicePoissonLoader needs a TensorDataset (tensors that can live in HBM)")
        dl = cls(ds.tensors, sample_rate=1 / len(data_loader), device=device, seed=seed, capacity=capacity,
                 rank=rank, world_size=world_size)
        dl.dataset = ds  # PrivacyEngine sizes the expected batch from the WHOLE dataset, then divides by world size
        return dl

    def __len__(self):
        return self.steps

    @property
    def seed(self) -> int:
        return self._seed

    @seed.setter
    def seed(self, value: int):
        if self._graph_captured and int(value) != self._seed:
            # the sampler's Philox key is a by-value argument of the captured K11 launch, like the optimizer's noise key
            raise RuntimeError("the sampler seed cannot change once its draws have been captured into a hipGraph: set it "
                               "before GraphedDPStep.capture(), or capture again")
        self._seed = int(value)

    @property
    def key(self) -> int:
        """Philox key of this rank's sampler stream: its own purpose tag (independent of the DP-noise stream under
        the same seed, rng.py) and a distinct stream per rank."""
        from .rng import RANK_STRIDE, stream_key
        return stream_key((self.seed + RANK_STRIDE * self.rank) & (2**64 - 1), "sampler")

    def sample_static(self):
        """One Poisson draw into the static batch tensors (all device-side; capturable into a hipGraph)."""
        self._native.poisson_sample(self.key, 0, self.sample_rate, self.num_samples, self.idx, self.counts,
                                    offset_dev=self._offset_dev)
        self._offset_dev.add_(self._blocks_per_draw)
        self._overflow.add_((self.counts[1] > self.capacity).to(torch.int64))
        for dst, src in zip(self._static, self.tensors):
            torch.index_select(src, 0, self.idx, out=dst)
        return self._static, self.n_valid

    def overflow_steps(self) -> int:
        return int(self._overflow.item())

    def check_overflow(self):
        """Raise if any shape-static draw so far selected more samples than `capacity` (one host sync)."""
        n = self.overflow_steps()
        if n:
            raise PoissonCapacityError(
                f"{n} Poisson draw(s) selected more than capacity = {self.capacity} samples and were truncated: the "
                "steps taken are not Poisson-sampled DP-SGD steps and the accountant's epsilon does not cover them. "
                "Re-run with a larger `capacity` (tensors captured into a hipGraph cannot grow).")

    def __iter__(self):
        for _ in range(self.steps):
            static, _ = self.sample_static()
            n, selected = self.counts.tolist()  # the one host sync of the step
            if selected > self.capacity:
                # not a truncated batch: grow, and repeat the same draw from the same Philox position
                self._allocate(int(selected * 1.25) + 8)
                self._offset_dev.sub_(self._blocks_per_draw)
                self._overflow.sub_(1)
                static, _ = self.sample_static()
                n, selected = self.counts.tolist()
                assert selected <= self.capacity or self.capacity == self.num_samples
            yield [t[:n] for t in static]
