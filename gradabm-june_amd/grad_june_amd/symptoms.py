"""Disease-stage progression (reference grad_june/symptoms.py:10-257) - row f1 of SURVEY section 8.

Runs right after the hot path each step.  ``SymptomsUpdater.forward`` is one fused per-agent HIP
kernel (``gj_symptoms_update``): the reference's ~60 elementwise ops, ``torch.bernoulli``, up to ten
``rsample((A,))`` draws and two device->host syncs per stage (``if n_symp > 0``) become a single
pass over the three symptom arrays.  Randomness is Philox4x32-10 keyed by (seed, step, agent) or,
for parity with the reference, injected (``progresses`` = the bernoulli outcome, ``dwell`` = the
stage-time sample each agent consumes).  Stage-time distributions other than LogNormal / Normal
are not supported by the kernel and raise.
"""
from __future__ import annotations

import ctypes as C

import torch
import yaml

from . import _native as N
from .utils import parse_age_probabilities, parse_distribution


class SymptomsSampler:
    def __init__(self, stages, stage_transition_probabilities, stage_transition_times, recovery_times, device):
        self.stages = stages
        self.device = device
        self.stages_ids = torch.arange(0, len(stages))
        n = len(stages)
        probs = torch.zeros((n, 100), device=device)
        for i, stage in enumerate(stages):
            if stage in stage_transition_probabilities:
                probs[i] = torch.tensor(parse_age_probabilities(stage_transition_probabilities[stage]),
                                        dtype=torch.float32, device=device)
        self.stage_transition_probabilities = probs
        self.stage_transition_times = {i: (parse_distribution(stage_transition_times[s], device)
                                           if s in stage_transition_times else None) for i, s in enumerate(stages)}
        self.recovery_times = {i: (parse_distribution(recovery_times[s], device) if s in recovery_times else None)
                               for i, s in enumerate(stages)}

    @classmethod
    def from_file(cls, fpath=None):
        if fpath is None:
            from .defaults import default_parameters

            return cls.from_parameters(default_parameters())
        with open(fpath) as f:
            return cls.from_parameters(yaml.safe_load(f))

    @classmethod
    def from_parameters(cls, params):
        return cls(**params["symptoms"], device=params["system"]["device"])

    def _get_need_to_transition(self, current_stage, time_to_next_stage, time):
        return (time >= time_to_next_stage) * (current_stage < len(self.stages) - 1)

    def _get_prob_next_symptoms_stage(self, ages, stages):
        return self.stage_transition_probabilities[stages, ages]

    def kernel_params(self) -> "N.SymptomsParams":
        """Tables of the state machine as the kernel takes them."""
        p = N.SymptomsParams()
        n = len(self.stages)
        if n > N.GJ_MAX_STAGES:
            raise NotImplementedError(f"at most {N.GJ_MAX_STAGES} symptom stages")
        p.n_stages = n
        for table, kind, loc, scale in ((self.stage_transition_times, p.next_kind, p.next_loc, p.next_scale),
                                        (self.recovery_times, p.rec_kind, p.rec_loc, p.rec_scale)):
            for i, dist in table.items():
                if dist is None:
                    kind[i] = 0
                elif isinstance(dist, torch.distributions.LogNormal):
                    kind[i], loc[i], scale[i] = 1, float(dist.loc), float(dist.scale)
                elif isinstance(dist, torch.distributions.Normal):
                    kind[i], loc[i], scale[i] = 2, float(dist.loc), float(dist.scale)
                else:
                    raise NotImplementedError(f"stage-time distribution {type(dist).__name__}: only LogNormal/Normal")
        return p

    def sample_next_stage(self, ages, current_stage, next_stage, time_to_next_stage, time, sex=None):
        """The reference's method (symptoms.py:82-128) on the HIP kernel: agents that are due move to their next stage
        and draw what follows (``gj_symptoms_update`` with nobody newly infected).  Returns new tensors."""
        from .world import require_hip

        device = require_hip(current_stage.device)
        n = ages.shape[0]
        cur, nxt, due = (t.detach().to(device=device, dtype=torch.float32).contiguous().clone()
                         for t in (current_stage, next_stage, time_to_next_stage))
        sexv = torch.zeros(n, dtype=torch.long, device=device) if sex is None else sex.to(device).long()
        cls = (sexv * 100 + ages.to(device).long()).to(torch.uint8).contiguous()
        table = self.stage_transition_probabilities.to(device=device, dtype=torch.float32).contiguous()
        p = self.kernel_params()
        p.progress = table.data_ptr()
        p.time = float(time)
        self._direct_calls = getattr(self, "_direct_calls", 0) + 1
        p.seed, p.step, p.agent_offset = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, (1 << 62) + self._direct_calls, 0
        zeros = torch.zeros(n, dtype=torch.float32, device=device)
        N.check(N.load().gj_symptoms_update(n, N.ptr(cls), N.ptr(zeros), N.ptr(cur), N.ptr(nxt), N.ptr(due), C.byref(p),
                                            None, None, N.current_stream()), "gj_symptoms_update")
        return cur, nxt, due


class SymptomsUpdater(torch.nn.Module):
    def __init__(self, symptoms_sampler):
        super().__init__()
        if not isinstance(symptoms_sampler, SymptomsSampler):
            raise TypeError("symptoms_sampler must be an instance of SymptomsSampler.")
        self.symptoms_sampler = symptoms_sampler

    @classmethod
    def from_file(cls, fpath=None):
        return cls(SymptomsSampler.from_file(fpath))

    @classmethod
    def from_parameters(cls, params):
        return cls(SymptomsSampler.from_parameters(params))

    @property
    def stages_ids(self):
        return self.symptoms_sampler.stages_ids

    def forward(self, data, timer, new_infected, progresses=None, dwell=None, stats=None):
        """HIP path (fused kernel).  ``progresses`` / ``dwell``: optional injected randomness [A].
        ``stats``: {"cls", "edges", "n_bins", "dead", "out"} - the Runner's per-step reductions (runner.py:167-171) of
        the UPDATED state are taken in the same pass (``gj_symptoms_step_stats``) into the zeroed fp64 row ``out``;
        ``stats["done"]`` is set when that happened (not in grad mode, where the update is an autograd node)."""
        from .world import require_hip

        ag = data["agent"]
        try:
            symptoms = ag.symptoms
        except (KeyError, AttributeError):
            raise KeyError("data must contain the 'agent' key.")
        for key in ("current_stage", "next_stage", "time_to_next_stage"):
            if key not in symptoms:
                raise KeyError("symptoms must contain the 'current_stage', 'next_stage', and "
                               "'time_to_next_stage' keys.")
        device = require_hip(new_infected.device)
        n = new_infected.numel()
        for key in ("current_stage", "next_stage", "time_to_next_stage"):
            t = symptoms[key]
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != device:
                symptoms[key] = t.detach().to(device=device, dtype=torch.float32).contiguous()
        sampler = self.symptoms_sampler
        cls = getattr(self, "_cls", None)
        if cls is None or cls.numel() != n or cls.device != device:
            sex = ag["sex"] if "sex" in ag else torch.zeros_like(ag.age)
            cls = self._cls = (sex.to(device).long() * 100 + ag.age.to(device).long()).to(torch.uint8).contiguous()
        table = getattr(self, "_table", None)
        if table is None or table.device != device:
            table = self._table = sampler.stage_transition_probabilities.to(device=device, dtype=torch.float32).contiguous()
        p = getattr(self, "_params", None)
        if p is None:      # built once: reading the distributions' parameters synchronises with the device
            p = self._params = sampler.kernel_params()
        p.progress = table.data_ptr()
        p.time = float(timer.now)
        if getattr(self, "rng_seed", None) is None:
            self.rng_seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        self.n_calls = getattr(self, "n_calls", 0) + 1
        # agent_offset: global id of local agent 0 (a rank of a partitioned run sets it; the Philox key is the global id)
        p.seed, p.step, p.agent_offset = self.rng_seed, self.n_calls, int(getattr(self, "agent_offset", 0))
        if (progresses is None) != (dwell is None):
            raise ValueError("inject both progresses and dwell, or neither")
        if progresses is not None:
            progresses = progresses.to(device=device, dtype=torch.float32).contiguous()
            dwell = dwell.to(device=device, dtype=torch.float32).contiguous()
        if torch.is_grad_enabled() and (new_infected.requires_grad or any(
                symptoms[k].requires_grad for k in ("current_stage", "next_stage", "time_to_next_stage"))):
            # row f3: the update as an autograd node, so that a loss on the stages (deaths) reaches log_beta
            from .autograd import SymptomsStep

            env = {"cls": cls, "params": p, "progresses": progresses, "dwell": dwell, "table": table}
            (symptoms["current_stage"], symptoms["next_stage"], symptoms["time_to_next_stage"]) = SymptomsStep.apply(
                env, new_infected, symptoms["current_stage"], symptoms["next_stage"], symptoms["time_to_next_stage"])
            self.used_kernel = True
            return symptoms
        nw = new_infected.detach().to(torch.float32).contiguous()
        if stats is not None:
            inf = ag.is_infected
            if inf.dtype != torch.float32 or not inf.is_contiguous() or inf.device != device:
                inf = inf.detach().to(device=device, dtype=torch.float32).contiguous()
            N.check(N.load().gj_symptoms_step_stats(
                n, N.ptr(cls), N.ptr(nw), N.ptr(symptoms["current_stage"]), N.ptr(symptoms["next_stage"]),
                N.ptr(symptoms["time_to_next_stage"]), C.byref(p), N.ptr(progresses), N.ptr(dwell), N.ptr(inf),
                int(stats["n_bins"]), stats["edges"], int(stats["dead"]), N.ptr(stats["out"]), N.current_stream()),
                "gj_symptoms_step_stats")
            stats["done"] = True
            self.used_kernel = True
            return symptoms
        N.check(N.load().gj_symptoms_update(n, N.ptr(cls), N.ptr(nw), N.ptr(symptoms["current_stage"]),
                                            N.ptr(symptoms["next_stage"]), N.ptr(symptoms["time_to_next_stage"]),
                                            C.byref(p), N.ptr(progresses), N.ptr(dwell), N.current_stream()),
                "gj_symptoms_update")
        self.used_kernel = True
        return symptoms
