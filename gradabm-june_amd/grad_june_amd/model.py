"""One simulated timestep (reference grad_june/model.py:18-144).

``GradJune.forward(data, timer) -> data`` keeps the reference's contract; rows a1-a9 in the middle
of it (model.py:125-138) run as the fused ``gj_step`` launch sequence on the HIP device, updating
``data["agent"].{transmission, susceptibility, is_infected, infection_time}`` in place.
"""
from __future__ import annotations

import math

import torch
import yaml

from .infection import IsInfectedSampler
from .infection_networks import InfectionNetworks
from .policies import Policies
from .symptoms import SymptomsUpdater
from .transmission import TransmissionUpdater
from .world import agent_buffers, engine_for, require_hip


class GradJune(torch.nn.Module):
    def __init__(self, symptoms_updater=None, policies=None, infection_networks=None, device="cuda:0"):
        super().__init__()
        self.symptoms_updater = symptoms_updater if symptoms_updater is not None else SymptomsUpdater.from_file()
        self.policies = policies if policies is not None else Policies.from_file()
        self.infection_networks = (infection_networks if infection_networks is not None
                                   else InfectionNetworks.from_file())
        self.transmission_updater = TransmissionUpdater()
        self.is_infected_sampler = IsInfectedSampler()
        self.device = device
        self.rng_seed = None        # Philox key; defaults to torch.initial_seed() at first use
        self.n_steps = 0            # Philox stream id: one stream per forward() call
        self.step_stats = None      # see forward()

    @classmethod
    def from_file(cls, fpath=None):
        if fpath is None:
            from .defaults import default_parameters

            return cls.from_parameters(default_parameters())
        with open(fpath) as f:
            return cls.from_parameters(yaml.safe_load(f))

    @classmethod
    def from_parameters(cls, params):
        return cls(
            symptoms_updater=SymptomsUpdater.from_parameters(params),
            policies=Policies.from_parameters(params),
            infection_networks=InfectionNetworks.from_parameters(params),
            device=params["system"]["device"],
        )

    def infect_people(self, data, timer, new_infected):
        from .infection import infect_people

        infect_people(data, timer, new_infected)

    def hot_path(self, data, timer, exp_noise=None, want_probs=False):
        """Rows a1-a9.  Returns (new_infected, not_infected_probs or None)."""
        device = require_hip(self.device)
        nets = self.infection_networks
        active = nets.active_networks(timer, self.policies)
        differentiable = torch.is_grad_enabled() and (
            any(isinstance(n.log_beta, torch.Tensor) and n.log_beta.requires_grad for n in active)
            or any(data["agent"][k].requires_grad for k in ("susceptibility", "is_infected", "infection_time")))
        self.policies.apply(timer=timer, data=data)
        engine = engine_for(data, [n.spec() for n in nets.networks.values()], device)
        for n in active:
            if n.name not in engine.plan.networks:
                raise KeyError(f"network '{n.name}': edge set 'attends_{n.edge_set}' is not in the world")
        qp = self.policies.quarantine_policies
        has_q = bool(qp)
        if self.rng_seed is None:
            self.rng_seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        params = engine.params(
            now=timer.now, delta_time=timer.duration, day_type=0 if timer.day_type == "weekday" else 1,
            active=[n.name for n in active], betas={n.name: n.beta_value(self.policies, timer) for n in active},
            has_quarantine=has_q, q_threshold=qp.threshold if has_q else math.inf,
            seed=self.rng_seed, step=self.n_steps)
        self.n_steps += 1
        if differentiable:
            return self._hot_path_differentiable(data, engine, params, active, has_q, exp_noise, want_probs)
        bufs = agent_buffers(engine, data, need_params=True, need_stage=has_q)
        n = engine.plan.host.n_agents
        new_infected = torch.empty(n, dtype=torch.float32, device=device)
        probs = torch.empty(n, dtype=torch.float32, device=device) if want_probs else None
        if exp_noise is not None:
            exp_noise = exp_noise.to(device=device, dtype=torch.float32).contiguous()
        engine.step(bufs, params, engine.io(not_infected_probs=probs, new_infected=new_infected, exp_noise=exp_noise))
        return new_infected, probs

    def _hot_path_differentiable(self, data, engine, params, active, has_q, exp_noise, want_probs):
        """Row f3: the step as an autograd node (grad_june_amd.autograd.HotPathStep)."""
        from .autograd import HotPathStep

        if want_probs:
            raise NotImplementedError("want_probs is not available in differentiable mode")
        dev = engine.plan.device
        ag = data["agent"]
        n = engine.plan.host.n_agents
        f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        ip = ag["infection_parameters"]
        fixed = {k: f(ip[k]) for k in ("max_infectiousness", "shape", "rate", "shift")}
        stage = f(ag["symptoms"]["current_stage"]).clone() if has_q else None
        if exp_noise is not None:
            exp_noise = exp_noise.to(device=dev, dtype=torch.float32).contiguous()
        env = {"engine": engine, "params": params, "fixed": fixed, "stage": stage, "exp_noise": exp_noise,
               "nets": list(active), "betas": {n_.name: float(params.nets[i].beta) for i, n_ in enumerate(active)}}
        state = [ag[k] if ag[k].dtype == torch.float32 else ag[k].to(torch.float32) for k in
                 ("susceptibility", "is_infected", "infection_time")]
        susc, inf, time, new_infected = HotPathStep.apply(env, *state, *[n_.log_beta for n_ in active])
        ag.susceptibility, ag.is_infected, ag.infection_time = susc, inf, time
        return new_infected, None

    def forward(self, data, timer, exp_noise=None):
        new_infected, _ = self.hot_path(data, timer, exp_noise=exp_noise)
        # in grad mode new_infected stays on the graph: the symptoms update is then an autograd node too
        # (autograd.SymptomsStep), which is what makes the deaths series differentiable (runner.py:198-215)
        # step_stats: set by the Runner for the duration of its time loop - the per-step result reductions
        # (runner.py:167-171) then ride on the symptoms pass instead of re-reading the arrays it just wrote
        self.symptoms_updater(data=data, timer=timer, new_infected=new_infected,
                              stats=getattr(self, "step_stats", None))
        return data
