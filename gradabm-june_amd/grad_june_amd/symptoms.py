"""Disease-stage progression (reference grad_june/symptoms.py:10-257).

NOT part of the accelerated path in this release ("next"): it runs
right after the hot path each step and is kept as device-side torch ops so that ``GradJune`` /
``Runner`` are complete.  Semantics follow the reference's state machine; unlike the reference it
never synchronises with the host (the reference's ``if n_symp > 0`` checks force a device->host
sync per stage), so the random-number stream differs - parity for this module is statistical.
"""
from __future__ import annotations

import torch
import yaml

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

    def sample_next_stage(self, ages, current_stage, next_stage, time_to_next_stage, time):
        n = ages.shape[0]
        moving = self._get_need_to_transition(current_stage, time_to_next_stage, time)
        current_stage = current_stage - (current_stage - next_stage) * moving
        stage_idx = current_stage.long()
        progresses = torch.bernoulli(self._get_prob_next_symptoms_stage(ages, stage_idx)).to(torch.bool)
        for i in range(2, len(self.stages) - 1):            # skip recovered, susceptible and dead
            here = (stage_idx == i) & moving.to(torch.bool)
            onward = (here & progresses).to(current_stage.dtype)
            recover = (here & ~progresses).to(current_stage.dtype)
            next_stage = next_stage + onward
            time_to_next_stage = time_to_next_stage + self.stage_transition_times[i].rsample((n,)) * onward
            next_stage = next_stage - next_stage * recover
            time_to_next_stage = time_to_next_stage + self.recovery_times[i].rsample((n,)) * recover
        return current_stage, next_stage, time_to_next_stage


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

    def forward(self, data, timer, new_infected):
        try:
            symptoms = data["agent"].symptoms
        except (KeyError, AttributeError):
            raise KeyError("data must contain the 'agent' key.")
        for key in ("current_stage", "next_stage", "time_to_next_stage"):
            if key not in symptoms:
                raise KeyError("symptoms must contain the 'current_stage', 'next_stage', and "
                               "'time_to_next_stage' keys.")
        time = timer.now
        # newly infected agents: next stage = exposed (2), due now
        nxt = symptoms["next_stage"] + new_infected * (2.0 - symptoms["next_stage"])
        due = symptoms["time_to_next_stage"] + new_infected * (time - symptoms["time_to_next_stage"])
        cur, nxt, due = self.symptoms_sampler.sample_next_stage(
            ages=data["agent"].age, current_stage=symptoms["current_stage"], next_stage=nxt,
            time_to_next_stage=due, time=time)
        symptoms["current_stage"], symptoms["next_stage"], symptoms["time_to_next_stage"] = cur, nxt, due
        return symptoms
