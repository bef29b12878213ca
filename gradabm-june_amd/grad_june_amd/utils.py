"""Small host helpers of the configuration surface (reference: grad_june/utils.py:17-94)."""
from __future__ import annotations

import datetime
import random
from pathlib import Path
from typing import Dict, List, Union

import numpy as np
import torch

PACKAGE_DIR = Path(__file__).resolve().parent


def read_path(path_str: Union[str, Path]) -> Path:
    """``@grad_june/...`` (reference spelling) and ``@grad_june_amd/...`` resolve inside this package."""
    path = Path(path_str)
    if path.parts and path.parts[0] in ("@grad_june", "@grad_june_amd"):
        return PACKAGE_DIR.joinpath(*path.parts[1:])
    return path


def read_date(date: Union[str, datetime.date, datetime.datetime]) -> datetime.datetime:
    if isinstance(date, str):
        return datetime.datetime.strptime(date, "%Y-%m-%d")
    if isinstance(date, datetime.datetime):
        return date
    if isinstance(date, datetime.date):
        return datetime.datetime(date.year, date.month, date.day)
    raise TypeError("date must be a string or a datetime.date object")


def parse_age_probabilities(age_dict: Dict[str, float], fill_value: float = 0) -> List[float]:
    """``{"lo-hi": p}`` -> 100 per-age values; age ``a`` falls in a bin when lo <= a < hi.

    Same lookup rule as the reference (grad_june/utils.py:47-72): the bin edges, ordered by
    their lower edge (ties keep file order), are laid out as one sorted list
    ``[lo0, hi0, lo1, hi1, ...]`` and ``searchsorted(edges, age + 1)`` picks the slot; odd slots
    are bins, even slots are the gaps between/around bins (``fill_value``).  Overlapping bins
    (the default care_visit table has ``75-85`` and ``75-100``) resolve exactly as there.
    """
    items = [(int(k.split("-")[0]), int(k.split("-")[1]), v) for k, v in age_dict.items()]
    order = np.argsort([lo for lo, _, _ in items])
    edges: List[int] = []
    slots: List[float] = [fill_value]
    for i in order:
        lo, hi, p = items[i]
        edges += [lo, hi]
        slots += [p, fill_value]
    return [slots[int(np.searchsorted(edges, age + 1))] for age in range(100)]


def parse_distribution(spec: dict, device):
    """``{"dist": "LogNormal", "loc": .., "scale": ..}`` -> torch.distributions object."""
    kwargs = dict(spec)
    cls = getattr(torch.distributions, kwargs.pop("dist"))
    return cls(**{k: torch.tensor(v, device=device, dtype=torch.float) for k, v in kwargs.items()})


def fix_seed(seed=None):
    if seed is None:
        seed = np.random.randint(0, 1000)
    print(f"Fixing seed to {seed}")
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed
