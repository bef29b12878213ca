"""The reference's model / runner API on one rank of a multi-GPU job (SURVEY section 8e at the API level).

``DistributedRunner.from_parameters(params)`` is ``Runner.from_parameters`` for a process of a
``torch.distributed`` job (one process per GPU, backend ``nccl`` = RCCL): every rank reads the same world
file, keeps the agents of its contiguous range and compiles its part of the contact graph
(``distributed.build_rank_world``); ``runner()`` then runs the same time loop, every timestep as the
multi-rank launch sequence of ``distributed.DistributedHotPath.run_step`` (halo all-to-all + partial-sum
all-reduce), with timer, policies and ``log_beta`` evaluated per step exactly as ``GradJune`` does.  Per-agent
work (symptoms, state updates) is local; the per-step result series are summed over the ranks once, after the
loop.  Sampling noise is Philox keyed by the GLOBAL agent id and sums are fixed-point, so the series equal the
single-GPU run's for the same seed (tests/test_gpu_distributed_virtual.py).  Differentiable like the single-GPU
``Runner`` (example_scripts/run_model.py:9-11): with a ``log_beta`` that requires grad the step is the autograd node
``autograd.DistributedHotPathStep`` and the result series are summed over the ranks on the graph, so a loss on
``results`` back-propagated on EVERY rank leaves the whole world's gradient in every rank's ``log_beta.grad``.

    torch.manual_seed(seed)                    # the same seed on every rank
    runner = DistributedRunner.from_parameters(params)          # params["system"]["device"] = this rank's GPU
    results, is_infected_local = runner()      # results identical on every rank; is_infected: this rank's agents
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from .distributed import DistributedHotPath, with_twins, world_from_data
from .engine import AgentBuffers
from .graph import HeteroData
from .model import GradJune
from .plan import SPLIT_SUFFIX
from .runner import Runner
from .timer import Timer
from .world import require_hip


class DistributedGradJune(GradJune):
    """``GradJune`` whose hot path runs one rank's part of the world (``partition`` first)."""

    def partition(self, data: HeteroData, group=None, rank: Optional[int] = None, world_size: Optional[int] = None,
                  collectives: bool = True) -> HeteroData:
        """Split the full world ``data`` (as ``Runner.get_data`` returns it, the same on every rank): compile this
        rank's part and return the rank-local ``HeteroData`` (agents [a0, a1) only; no edge lists - the compiled
        plan holds them)."""
        import torch.distributed as dist

        device = require_hip(self.device)
        if rank is None:
            rank, world_size = dist.get_rank(group), dist.get_world_size(group)
        nets = self.infection_networks.networks
        specs = [n.spec() for n in nets.values()]
        world = world_from_data(data, model=self)
        has_q = bool(self.policies.quarantine_policies)
        self._hp = DistributedHotPath(world, specs, dict.fromkeys(world["networks"], 1.0), device, rank, world_size,
                                      group=group, collectives=collectives,
                                      quarantine_threshold=0.0 if has_q else None)
        hp = self._hp
        a0, a1 = hp.a0, hp.a0 + hp.rw.n_local
        self.agent_range = (a0, a1)
        self.symptoms_updater.agent_offset = a0
        if collectives and world_size > 1:       # one Philox key for the whole job: rank 0's
            seed = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64,
                                device=device if dist.get_backend(group) != "gloo" else "cpu")
            dist.broadcast(seed, src=0, group=group)
            self.rng_seed = int(seed.item())
            self.symptoms_updater.rng_seed = self.rng_seed
        local = HeteroData()
        ag, src = local["agent"], data["agent"]

        def cut(v):
            if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == world["n_agents"]:
                return v[a0:a1].clone().to(device)
            if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == world["n_agents"]:
                return v[a0:a1].copy()
            if isinstance(v, dict):
                return {k: cut(x) for k, x in v.items()}
            return v

        for name in list(src.keys()):
            ag[name] = cut(src[name])
        ag.transmission = hp.state["transmission"][: hp.rw.n_local]
        return local

    def hot_path(self, data, timer, exp_noise=None, want_probs=False):
        """Rows a1-a9 across the ranks.  Returns (new_infected, not_infected_probs or None) of the local agents."""
        hp = getattr(self, "_hp", None)
        if hp is None:
            raise RuntimeError("call partition(data) first")
        nets = self.infection_networks
        active = nets.active_networks(timer, self.policies)
        differentiable = torch.is_grad_enabled() and (
            any(isinstance(n.log_beta, torch.Tensor) and n.log_beta.requires_grad for n in active)
            or any(data["agent"][k].requires_grad for k in ("susceptibility", "is_infected", "infection_time")))
        self.policies.apply(timer=timer, data=data)
        engine, dev, n = hp.engine, hp.device, hp.rw.n_local
        for net in active:
            if net.name not in engine.plan.networks:
                raise KeyError(f"network '{net.name}': edge set 'attends_{net.edge_set}' is not in the world")
        qp = self.policies.quarantine_policies
        has_q = bool(qp)
        betas = {net.name: net.beta_value(self.policies, timer) for net in active}
        betas.update({k + SPLIT_SUFFIX: v for k, v in list(betas.items())})
        if self.rng_seed is None:
            self.rng_seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        step = self.n_steps
        self.n_steps += 1
        day_type = 0 if timer.day_type == "weekday" else 1

        now, duration, seed = timer.now, timer.duration, self.rng_seed      # (the backward pass calls params_of later)
        q_threshold = qp.threshold if has_q else math.inf

        def params_of(sets):
            # a network on a set the partition split runs on both halves (its twin: same beta)
            nets_by_name = {net.name: net for net in active}
            names = with_twins([net.name for net in active], lambda n: nets_by_name[n].edge_set,
                               lambda n: n + SPLIT_SUFFIX if n + SPLIT_SUFFIX in engine.plan.networks else None)
            if sets is not None:
                names = [n for n in names if engine.plan.networks[n].edge_set in sets]
            return engine.params(now=now, delta_time=duration, day_type=day_type, active=names, betas=betas,
                                 has_quarantine=has_q, q_threshold=q_threshold, seed=seed, step=step,
                                 agent_offset=hp.a0)

        ag = data["agent"]
        if differentiable:
            return self._hot_path_differentiable(data, hp, params_of, active, betas, has_q, exp_noise, want_probs)

        def f32(t):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            return t

        for k in ("susceptibility", "is_infected", "infection_time"):
            ag[k] = f32(ag[k])                                    # updated in place by the launch
        ip = ag["infection_parameters"]
        stage = f32(ag["symptoms"]["current_stage"]) if has_q else None
        bufs = AgentBuffers(engine.plan, max_infectiousness=f32(ip["max_infectiousness"]), shape=f32(ip["shape"]),
                            rate=f32(ip["rate"]), shift=f32(ip["shift"]), infection_time=ag["infection_time"],
                            is_infected=ag["is_infected"], susceptibility=ag["susceptibility"],
                            transmission=hp.state["transmission"], q_transmission=hp.state["q_transmission"],
                            current_stage=stage)
        new_infected = torch.empty(n, dtype=torch.float32, device=dev)
        probs = torch.empty(n, dtype=torch.float32, device=dev) if want_probs else None
        if exp_noise is not None:                                 # [2, A] for the whole world or [2, n_local]
            exp_noise = exp_noise.to(device=dev, dtype=torch.float32).reshape(2, -1)
            if exp_noise.shape[1] != n:
                exp_noise = exp_noise[:, hp.a0:hp.a0 + n]
            exp_noise = exp_noise.contiguous()
        hp.run_step(bufs, engine.io(not_infected_probs=probs, new_infected=new_infected, exp_noise=exp_noise), params_of)
        ag.transmission = hp.state["transmission"][:n]
        return new_infected, probs


    def _hot_path_differentiable(self, data, hp, params_of, active, betas, has_q, exp_noise, want_probs):
        """Row f3 across the ranks: the step as the autograd node ``autograd.DistributedHotPathStep``."""
        from .autograd import DistributedHotPathStep

        if want_probs:
            raise NotImplementedError("want_probs is not available in differentiable mode")
        dev, n = hp.device, hp.rw.n_local
        ag = data["agent"]
        f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        ip = ag["infection_parameters"]
        fixed = {k: f(ip[k]) for k in ("max_infectiousness", "shape", "rate", "shift")}
        stage = f(ag["symptoms"]["current_stage"]).clone() if has_q else None
        if exp_noise is not None:
            exp_noise = exp_noise.to(device=dev, dtype=torch.float32).reshape(2, -1)
            if exp_noise.shape[1] != n:
                exp_noise = exp_noise[:, hp.a0:hp.a0 + n]
            exp_noise = exp_noise.contiguous()
        env = {"hp": hp, "params_of": params_of, "fixed": fixed, "stage": stage, "exp_noise": exp_noise,
               "nets": list(active), "betas": dict(betas)}
        state = [ag[k] if ag[k].dtype == torch.float32 else ag[k].to(torch.float32) for k in
                 ("susceptibility", "is_infected", "infection_time")]
        susc, inf, time, new_infected = DistributedHotPathStep.apply(env, *state, *[n_.log_beta for n_ in active])
        ag.susceptibility, ag.is_infected, ag.infection_time = susc, inf, time
        ag.transmission = hp.state["transmission"][:n]
        return new_infected, None


class DistributedRunner(Runner):
    """``Runner`` for one rank; see the module docstring."""

    @classmethod
    def from_parameters(cls, params, group=None, rank: Optional[int] = None, world_size: Optional[int] = None,
                        collectives: bool = True):
        model = DistributedGradJune.from_parameters(params)
        full = Runner.get_data(params)                # the whole world, identical on every rank (same torch seed)
        n_total = len(full["agent"]["id"])
        local = model.partition(full, group=group, rank=rank, world_size=world_size, collectives=collectives)
        del full
        torch.cuda.empty_cache()
        runner = cls(model=model, data=local, timer=Timer.from_parameters(params),
                     log_fraction_initial_cases=params["infection_seed"]["log_fraction_initial_cases"],
                     save_path=params["save_path"], parameters=params,
                     age_bins=params.get("age_bins_to_save", (0, 18, 65, 100)))
        runner.agent_offset = model.agent_range[0]
        runner.n_agents_total = n_total
        runner.group, runner.collectives = group, collectives
        return runner

    def _reduce_differentiable(self, series: torch.Tensor) -> torch.Tensor:
        from .autograd import AllReduceSum

        if not self.collectives:
            return series
        return AllReduceSum.apply(self.model._hp, series)

    def _finalize_series(self, n_rows: int) -> None:
        import torch.distributed as dist

        if not self.collectives or not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        if dist.get_backend(self.group) == "gloo":                # tests: staged through the host
            host = self._series[:n_rows].cpu()
            dist.all_reduce(host, group=self.group)
            self._series[:n_rows].copy_(host)
        else:
            dist.all_reduce(self._series[:n_rows], group=self.group)
