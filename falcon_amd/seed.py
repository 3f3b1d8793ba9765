"""Deterministic runs: one call seeds every host-side random generator the package can touch.

The reference seeds `random` and NumPy with 42 when `falcon.falcon` is imported (falcon/seed.py, used at
falcon.py:30); the device path itself draws no random numbers (k-means starts from fixed rows, DESIGN.md
section 3), so this only matters for host-side helpers and for torch code a caller runs next to it.
"""
from __future__ import annotations

import random
import sys
from typing import Optional

import numpy as np

DEFAULT_SEED = 42
_active: Optional[int] = None


def set_seeds(seed: Optional[int] = None) -> int:
    """Seed `random`, NumPy's legacy generator and -- if the caller already imported it -- torch.
    Returns the seed in force."""
    global _active
    value = DEFAULT_SEED if seed is None else int(seed)
    for seeder in (random.seed, np.random.seed):
        seeder(value)
    torch = sys.modules.get("torch")
    if torch is not None:
        torch.manual_seed(value)
    _active = value
    return value


def active_seed() -> Optional[int]:
    """The seed of the last `set_seeds` call (None before the first)."""
    return _active
