"""Case sharding for batched inference and preprocessing: cases are independent, so each rank takes a
contiguous slice of the case list and there is no data-path collective."""
from __future__ import annotations

from typing import Tuple


def case_range(n_cases: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the cases rank `rank` owns; sizes differ by at most one."""
    base, rem = divmod(n_cases, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
