"""Per-agent infectiousness profile (reference grad_june/transmission.py:8-51).

``TransmissionSampler`` is the one-time draw of the four profile parameters (setup, plain torch).
``TransmissionUpdater.forward`` is row a1 of the hot path and runs as ``gj_transmission_update``.
"""
from __future__ import annotations

import math

import torch
import yaml

from .utils import parse_distribution
from .world import agent_buffers, engine_for, require_hip


class TransmissionSampler:
    def __init__(self, max_infectiousness, shape, rate, shift):
        self.max_infectiousness = max_infectiousness
        self.shape = shape
        self.rate = rate
        self.shift = shift

    def __call__(self, n):
        """[4, n]: max_infectiousness, shape, rate, shift - drawn in this order (RNG stream order)."""
        rows = [d.rsample((n,)) for d in (self.max_infectiousness, self.shape, self.rate, self.shift)]
        return torch.vstack(rows)

    @classmethod
    def from_file(cls, fpath=None):
        if fpath is None:
            from .defaults import default_parameters

            return cls.from_parameters(default_parameters())
        with open(fpath) as f:
            return cls.from_parameters(yaml.safe_load(f))

    @classmethod
    def from_parameters(cls, params):
        device = params["system"]["device"]
        return cls(**{k: parse_distribution(v, device=device) for k, v in params["transmission"].items()})


class TransmissionUpdater(torch.nn.Module):
    def forward(self, data, timer):
        """transmission[a] at ``timer.now`` - a NEW tensor, like the reference returns."""
        ag = data["agent"]
        device = require_hip(ag["is_infected"].device)
        engine = engine_for(data, [], device)      # the transmission kernel needs no edge set
        out = torch.empty(engine.plan.host.n_agents, dtype=torch.float32, device=device)
        bufs = agent_buffers(engine, data, need_params=True, need_stage=False)
        bufs.tensors["transmission"] = out
        bufs.c.transmission = out.data_ptr()
        p = engine.params(now=timer.now, delta_time=timer.duration, day_type=0, active=[], betas={},
                          has_quarantine=False, q_threshold=math.inf)
        engine.transmission_update(bufs, p)
        return out
