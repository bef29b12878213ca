"""Infection sampling and state update (reference grad_june/infection.py:3-63).

``IsInfectedSampler.forward`` (a8) and ``infect_people`` (a9) run as ``gj_sample_infect``; the
two seeding helpers are one-time setup and stay plain torch/numpy like the reference's.
"""
from __future__ import annotations

import itertools

import torch

from . import _native as N
from .world import require_hip

_philox_step = itertools.count(1 << 40)   # sampler calls outside GradJune.forward get their own streams


def _launch_sample(p, exp_noise, new_inf, now=0.0, state=(None, None, None), seed=None, step=None, agent_offset=0):
    lib = N.load()
    if seed is None:
        seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
    if step is None:
        step = next(_philox_step)
    s, i, t = state
    N.check(lib.gj_sample_infect(p.numel(), N.ptr(p), N.ptr(exp_noise), int(seed), int(step), int(agent_offset), float(now),
                                 N.ptr(new_inf), N.ptr(s), N.ptr(i), N.ptr(t), N.current_stream()),
            "gj_sample_infect")


class IsInfectedSampler(torch.nn.Module):
    def forward(self, not_infected_probs, exp_noise=None):
        """Hard Gumbel-softmax (tau 0.1) over {not infected, infected}; returns 1.0 where infected.

        ``exp_noise`` ([2, A] Exponential(1) draws, row 0 = "not infected") reproduces the reference
        bit-for-bit decisions given its noise; without it the kernel draws Philox4x32-10 noise keyed
        by (torch.initial_seed(), call counter, agent)."""
        p = not_infected_probs
        require_hip(p.device)
        if torch.is_grad_enabled() and p.requires_grad:
            return self._forward_straight_through(p, exp_noise)
        p = p.detach().to(torch.float32).contiguous()
        if exp_noise is not None:
            exp_noise = exp_noise.to(device=p.device, dtype=torch.float32).contiguous()
            if exp_noise.numel() != 2 * p.numel():
                raise ValueError("exp_noise must be [2, A]")
        out = torch.empty_like(p)
        _launch_sample(p, exp_noise, out)
        return out

    def _forward_straight_through(self, p, exp_noise):
        """Stand-alone call in grad mode: the kernel takes the hard decision, the soft sample that carries the
        gradient (F.gumbel_softmax(..., tau=0.1, hard=True), infection.py:13-18) is formed with device tensor ops
        from the same noise.  (Inside GradJune.forward the whole step is one autograd node instead.)"""
        n = p.numel()
        if exp_noise is None:
            exp_noise = torch.empty(2, n, dtype=torch.float32, device=p.device).exponential_()
        exp_noise = exp_noise.to(device=p.device, dtype=torch.float32).contiguous().view(2, n)
        hard = torch.empty(n, dtype=torch.float32, device=p.device)
        _launch_sample(p.detach().to(torch.float32).contiguous(), exp_noise, hard)
        logits = torch.vstack((p, 1.0 - p)).log()
        y_soft = torch.softmax((logits - exp_noise.log()) / 0.1, dim=0)
        ret0 = (1.0 - hard) - y_soft[0].detach() + y_soft[0]
        return 1.0 - ret0


def infect_people(data, timer, new_infected):
    """susceptibility/is_infected/infection_time update for a given 0/1 vector (model.py:103-110)."""
    ag = data["agent"]
    nw = new_infected
    ag.susceptibility = torch.clamp(ag.susceptibility - nw, min=0.0)
    ag.is_infected = ag.is_infected + nw
    ag.infection_time = ag.infection_time + nw * (timer.now - ag.infection_time)


def infect_fraction_of_people(data, timer, symptoms_updater, fraction, device, exp_noise=None, agent_offset=0):
    """Seed: every agent infected independently with probability ``fraction`` (a8+a9 fused launch).
    ``agent_offset``: global id of local agent 0 when ``data`` is one rank's part of a partitioned world."""
    device = require_hip(device)
    ag = data["agent"]
    n = ag.id.shape[0]
    probs = torch.full((n,), 1.0 - float(fraction), dtype=torch.float32, device=device)
    for k in ("susceptibility", "is_infected", "infection_time"):
        ag[k] = ag[k].detach().to(device=device, dtype=torch.float32).contiguous()
    new_inf = torch.empty(n, dtype=torch.float32, device=device)
    if exp_noise is not None:
        exp_noise = exp_noise.to(device=device, dtype=torch.float32).contiguous()
    _launch_sample(probs, exp_noise, new_inf, now=timer.now,
                   state=(ag.susceptibility, ag.is_infected, ag.infection_time), agent_offset=agent_offset)
    return new_inf


def infect_people_at_indices(data, indices, device="cuda:0"):
    ag = data["agent"]
    idx = torch.as_tensor(list(indices), dtype=torch.long, device=ag["susceptibility"].device)
    for key, value in (("susceptibility", 0.0), ("is_infected", 1.0), ("infection_time", 0.0)):
        t = ag[key].clone()
        t[idx] = value
        ag[key] = t.to(device)
    for key, value in (("next_stage", 2), ("current_stage", 1)):
        t = ag["symptoms"][key].clone()
        t[idx] = value
        ag["symptoms"][key] = t.to(device)
    return data
