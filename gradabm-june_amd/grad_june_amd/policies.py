"""Date-windowed interventions: host-side logic whose OUTPUTS are hot-path inputs.

Mirror of the reference's policy classes (grad_june/policies/*.py): same class names, constructor
arguments, YAML schema and ``apply`` signatures.  What reaches the kernels per step:
  * SocialDistancing  -> a scalar factor on a network's beta      (interaction_policies.py:25-31)
  * Quarantine        -> ``q[a] = current_stage[a] < threshold``   (quarantine_policies.py:13-33);
                         the product over active policies is ``stage < min(thresholds)``, so the
                         kernels take one threshold (+inf when no policy is active)
  * CloseVenue        -> networks dropped from the step's list     (close_venue_policies.py:11-22)
"""
from __future__ import annotations

import math
import sys
from typing import List, Optional

import torch
import yaml

from .utils import read_date


class Policy(torch.nn.Module):
    spec: Optional[str] = None

    def __init__(self, start_date, end_date, device="cpu"):
        super().__init__()
        self.start_date = read_date(start_date)
        self.end_date = read_date(end_date)
        self.device = device

    def is_active(self, date) -> bool:
        return self.start_date <= date < self.end_date

    def apply(self, *args, **kwargs):
        raise NotImplementedError


class PolicyCollection(torch.nn.Module):
    """Policies of one kind.  Always truthy, like the reference's (an ``nn.Module``)."""

    def __init__(self, policies):
        super().__init__()
        self.policies = torch.nn.ModuleList(policies)

    def __getitem__(self, idx):
        return self.policies[idx]


# ---- interaction -------------------------------------------------------------------------------
class InteractionPolicy(Policy):
    spec = "interaction"


class SocialDistancing(InteractionPolicy):
    def __init__(self, start_date, end_date, beta_factors, device="cpu"):
        super().__init__(start_date, end_date, device)
        # factors live on the host: beta is a launch scalar, never a device tensor
        self.beta_factors = {k: torch.tensor(float(v)) for k, v in beta_factors.items()}

    def apply(self, beta, name, timer):
        if not self.is_active(timer.date):
            return beta
        factor = self.beta_factors.get(name)
        if factor is None:
            factor = self.beta_factors.get("all", torch.tensor(1.0))
        return beta * factor


class InteractionPolicies(PolicyCollection):
    def apply(self, beta, name, timer):
        for policy in self.policies:
            beta = policy.apply(beta=beta, name=name, timer=timer)
        return beta


# ---- quarantine --------------------------------------------------------------------------------
class Quarantine(Policy):
    spec = "quarantine"

    def __init__(self, start_date, end_date, stage_threshold, device="cpu"):
        super().__init__(start_date, end_date, device)
        self.stage_threshold = stage_threshold

    def apply(self, symptom_stages, timer):
        if self.is_active(timer.date):
            return (symptom_stages < self.stage_threshold).to(torch.float)
        return torch.ones(symptom_stages.shape, device=symptom_stages.device)


class QuarantinePolicies(PolicyCollection):
    def __init__(self, policies):
        super().__init__(policies)
        self._stages = None
        self.threshold = math.inf

    def apply(self, symptom_stages, timer):
        """Records the step's threshold; the per-agent mask itself is formed inside the kernels
        (and lazily by :attr:`quarantine_mask` for callers that want the tensor)."""
        self._stages = symptom_stages
        active = [float(p.stage_threshold) for p in self.policies if p.is_active(timer.date)]
        self.threshold = min(active) if active else math.inf

    @property
    def quarantine_mask(self):
        if self._stages is None:
            return 1.0
        if math.isinf(self.threshold):
            return torch.ones(self._stages.shape, device=self._stages.device)
        return (self._stages < self.threshold).to(torch.float)


# ---- close venue -------------------------------------------------------------------------------
class CloseVenue(Policy):
    spec = "close_venue"

    def __init__(self, start_date, end_date, names, device="cpu"):
        super().__init__(start_date, end_date, device)
        self.edge_type_to_close = {str(n) for n in names}

    def apply(self, edge_types, timer):
        if not self.is_active(timer.date):
            return edge_types
        return [e for e in edge_types if e not in self.edge_type_to_close]


class CloseVenuePolicies(PolicyCollection):
    def apply(self, edge_types, timer):
        for policy in self.policies:
            edge_types = policy.apply(edge_types=edge_types, timer=timer)
        return edge_types


# ---- container ---------------------------------------------------------------------------------
class Policies(torch.nn.Module):
    def __init__(self, interaction_policies=None, quarantine_policies=None, close_venue_policies=None):
        super().__init__()
        self.interaction_policies = interaction_policies
        self.quarantine_policies = quarantine_policies
        self.close_venue_policies = close_venue_policies

    @classmethod
    def from_policy_list(cls, policies):
        policies = list(policies or [])

        def of(kind):
            return [p for p in policies if p.spec == kind]

        return cls(
            interaction_policies=InteractionPolicies(of("interaction")),
            quarantine_policies=QuarantinePolicies(of("quarantine")),
            close_venue_policies=CloseVenuePolicies(of("close_venue")),
        )

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
        found: List[Policy] = []
        for group in (params.get("policies") or {}).values():
            for name, config in group.items():
                found += cls._parse_policy_config(config, name=name, device=device)
        return cls.from_policy_list(found)

    @staticmethod
    def _parse_policy_config(config, name, device):
        """``social_distancing`` -> class ``SocialDistancing``; either one window or numbered windows."""
        cls_name = "".join(part.capitalize() for part in name.split("_"))
        policy_class = getattr(sys.modules[__name__], cls_name)
        if "start_date" in config:
            return [policy_class(**config, device=device)]
        out = []
        for window in config.values():
            if "start_date" not in window or "end_date" not in window:
                raise ValueError("policy config file not valid.")
            out.append(policy_class(**window, device=device))
        return out

    def apply(self, data, timer):
        if self.quarantine_policies:
            self.quarantine_policies.apply(
                timer=timer, symptom_stages=data["agent"]["symptoms"]["current_stage"]
            )
