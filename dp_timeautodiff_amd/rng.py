"""Philox key derivation: one user seed, independent streams per purpose.

K7's DP noise and K11's Poisson sampler both run Philox4x32-10 keyed by a 64-bit word and both count blocks from
0.  Fed the same word (dp_job.py --seed S used to do that) the uniforms that decide which samples enter a batch
would be the very uniforms Box-Muller turns into the published noise: DP's analysis needs the two independent.
So no kernel is ever keyed with a user seed directly; each purpose gets its own key,

    key = splitmix64(seed XOR tag(purpose)),

and two purposes can share a seed without sharing a stream.  (The kernels' C ABI takes the KEY; see include/dpk.h.)
"""
from __future__ import annotations

_MASK = (1 << 64) - 1
_TAGS = {
    "noise": 0x6E6F6973655F6470,    # "noise_dp"
    "sampler": 0x706F6973736F6E5F,  # "poisson_"
}
RANK_STRIDE = 0x9E3779B97F4A7C15  # per-rank sampler seeds: seed + RANK_STRIDE * rank


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _MASK
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _MASK
    return x ^ (x >> 31)


def stream_key(seed: int, purpose: str) -> int:
    """64-bit Philox key of `purpose` ("noise" | "sampler") under the user's `seed`."""
    return splitmix64((int(seed) & _MASK) ^ _TAGS[purpose])
