"""Infection networks: the operator API of the hot path, backed by the HIP kernels.

Drop-in mirror of the reference's ``grad_june.infection_networks`` package (base.py:11-167,
leisure_network.py:7-120): the same eleven ``<Venue>Network`` class names (looked up by name from
the YAML ``networks:`` section, base.py:99-110), ``InfectionNetworks(device, **networks)`` with a
``.networks`` ModuleDict whose ``log_beta`` a user may swap for an ``nn.Parameter``
(example_scripts/run_model.py:6-8), and ``forward(data, timer, policies) -> not_infected_probs[A]``.

What differs is only WHERE the arithmetic runs: a network no longer gathers/scatters with PyG on
the host; it contributes a descriptor (edge set, mask recipe, beta, leisure table) and the two
passes over all active networks run as ``gj_venue_reduce`` + ``gj_agent_gather``.
"""
from __future__ import annotations

import math
import re
import sys
from typing import List

import numpy as np
import torch
import yaml

from . import _native as N
from .plan import NetworkSpec
from .utils import parse_age_probabilities
from .world import agent_buffers, engine_for, require_hip


class InfectionNetwork(torch.nn.Module):
    mask_kind = N.MASK_Q

    def __init__(self, log_beta, device="cuda:0"):
        super().__init__()
        self.device = device
        if type(log_beta) != torch.nn.Parameter:
            log_beta = torch.tensor(float(log_beta))
        self.log_beta = log_beta
        self.name = self._get_name()

    @classmethod
    def _get_name(cls) -> str:
        """``CareVisitNetwork`` -> ``care_visit`` (reference base.py:26-28)."""
        words = re.findall("[A-Z][^A-Z]*", cls.__name__)
        return "_".join(words[:-1]).lower()

    @classmethod
    def from_parameters(cls, params):
        return cls(device=params["system"]["device"], **params["networks"][cls._get_name()])

    # descriptor consumed by the graph compiler / kernels -----------------------------------------
    @property
    def edge_set(self) -> str:
        return self.name

    def spec(self) -> NetworkSpec:
        return NetworkSpec(self.name, self.edge_set, self.mask_kind, None)

    def beta_value(self, policies, timer) -> float:
        """10**log_beta times the active SocialDistancing factors, in fp32 like the reference
        (base.py:36-42, interaction_policies.py:25-31); a host scalar handed to the launch."""
        beta = 10.0 ** self.log_beta.detach().to("cpu", torch.float32)
        ip = getattr(policies, "interaction_policies", None)
        if ip:
            beta = ip.apply(beta=beta, name=self.name, timer=timer)
        return float(beta)

    # introspection helpers with the reference's names --------------------------------------------
    def _get_edge_index(self, data):
        return data["attends_" + self.edge_set].edge_index

    def _get_reverse_edge_index(self, data):
        return data["rev_attends_" + self.edge_set].edge_index

    def _get_people_per_group(self, data):
        return data[self.edge_set]["people"]

    def _get_beta(self, policies, timer, data):
        n = len(data[self.edge_set]["id"])
        return self.beta_value(policies, timer) * torch.ones(n, device=self.device)

    def _quarantine(self, policies):
        qp = getattr(policies, "quarantine_policies", None)
        return qp.quarantine_mask if qp else 1.0

    def _get_transmissions(self, data, policies, timer):
        return self._quarantine(policies) * data["agent"].transmission

    def _get_susceptibilities(self, data, policies, timer):
        return self._quarantine(policies) * data["agent"].susceptibility

    def forward(self, data, timer, policies):
        """This network's ``trans_susc`` term alone (reference InfectionNetwork.forward, base.py:61-84)."""
        return _run_networks([self], data, timer, policies, self.device, want="trans_susc")


class HouseholdNetwork(InfectionNetwork):
    mask_kind = N.MASK_RAW   # household contacts ignore quarantine (reference base.py:144-149)

    def _get_transmissions(self, data, policies, timer):
        return data["agent"].transmission

    def _get_susceptibilities(self, data, policies, timer):
        return data["agent"].susceptibility


class CareHomeNetwork(InfectionNetwork):
    pass


class SchoolNetwork(InfectionNetwork):
    pass


class CompanyNetwork(InfectionNetwork):
    pass


class UniversityNetwork(InfectionNetwork):
    pass


class LeisureNetwork(InfectionNetwork):
    """Attendance-probability-weighted network on the shared ``attends_leisure`` edge set
    (reference leisure_network.py:7-85)."""

    mask_kind = N.MASK_QL

    def __init__(self, log_beta, leisure_probabilities, device="cuda:0"):
        super().__init__(log_beta=log_beta, device=device)
        self.leisure_probabilities = self._parse_leisure_probabilities(leisure_probabilities)
        self.weekday_probabilities = None
        self.weekend_probabilities = None

    @classmethod
    def from_parameters(cls, params):
        name = cls._get_name()
        return cls(device=params["system"]["device"], leisure_probabilities=params["leisure"][name],
                   **params["networks"][name])

    def _parse_leisure_probabilities(self, probs) -> torch.Tensor:
        table = np.zeros((2, 2, 100), dtype=np.float32)
        for i, day in enumerate(("weekday", "weekend")):
            for j, sex in enumerate(("male", "female")):
                table[i, j] = np.asarray(parse_age_probabilities(probs[day][sex]), dtype=np.float32)
        return torch.from_numpy(table)   # host copy; the device copy lives in the plan

    @property
    def edge_set(self) -> str:
        return "leisure"

    def spec(self) -> NetworkSpec:
        return NetworkSpec(self.name, "leisure", self.mask_kind, self.leisure_probabilities.numpy())

    def initialize_leisure_probabilities(self, data):
        sex, age = data["agent"].sex.cpu(), data["agent"].age.cpu()
        dev = data["agent"].age.device
        self.weekday_probabilities = self.leisure_probabilities[0, sex, age].to(dev)
        self.weekend_probabilities = self.leisure_probabilities[1, sex, age].to(dev)

    def _leisure_mask(self, data, timer):
        if self.weekday_probabilities is None:
            self.initialize_leisure_probabilities(data)
        return self.weekday_probabilities if timer.day_type == "weekday" else self.weekend_probabilities

    def _get_transmissions(self, data, policies, timer):
        return self._quarantine(policies) * self._leisure_mask(data, timer) * data["agent"].transmission

    def _get_susceptibilities(self, data, policies, timer):
        return self._quarantine(policies) * self._leisure_mask(data, timer) * data["agent"].susceptibility


class PubNetwork(LeisureNetwork):
    pass


class CinemaNetwork(LeisureNetwork):
    pass


class GroceryNetwork(LeisureNetwork):
    pass


class GymNetwork(LeisureNetwork):
    pass


class VisitNetwork(LeisureNetwork):
    pass


class CareVisitNetwork(LeisureNetwork):
    mask_kind = N.MASK_QL_AGE75   # only the over-75s are exposed (reference leisure_network.py:107-120)

    def _get_susceptibilities(self, data, policies, timer):
        return super()._get_susceptibilities(data, policies, timer) * (data["agent"].age > 75)


def _step_inputs(active_networks, all_networks, data, timer, policies, device):
    """Everything a launch needs: engine (plan cached on ``data``), params, buffers."""
    engine = engine_for(data, [n.spec() for n in all_networks], device)
    qp = getattr(policies, "quarantine_policies", None)
    has_q = bool(qp)
    betas = {n.name: n.beta_value(policies, timer) for n in active_networks}
    for n in active_networks:
        if n.name not in engine.plan.networks:
            raise KeyError(f"network '{n.name}': edge set 'attends_{n.edge_set}' is not in the world")
    params = engine.params(
        now=timer.now, delta_time=timer.duration, day_type=0 if timer.day_type == "weekday" else 1,
        active=[n.name for n in active_networks], betas=betas, has_quarantine=has_q,
        q_threshold=qp.threshold if has_q else math.inf)
    return engine, params, has_q


def _run_networks(active_networks, data, timer, policies, device, want="probs", all_networks=None):
    device = require_hip(device)
    if hasattr(policies, "apply"):
        policies.apply(timer=timer, data=data)
    engine, params, has_q = _step_inputs(active_networks, all_networks or active_networks, data, timer, policies, device)
    ag = data["agent"]
    if torch.is_grad_enabled() and (
            any(isinstance(n.log_beta, torch.Tensor) and n.log_beta.requires_grad for n in active_networks)
            or any(isinstance(ag.get(k), torch.Tensor) and ag[k].requires_grad for k in ("transmission", "susceptibility"))):
        # row f3: the stand-alone forward as an autograd node (gradients w.r.t. log_beta, transmission, susceptibility)
        from .autograd import NetworksForward

        stage = None
        if has_q:
            stage = ag["symptoms"]["current_stage"].detach().to(device=device, dtype=torch.float32).contiguous()
        env = {"engine": engine, "params": params, "want": want, "nets": list(active_networks), "stage": stage,
               "betas": {n.name: float(params.nets[i].beta) for i, n in enumerate(active_networks)}}
        f = lambda t: t if t.dtype == torch.float32 else t.to(torch.float32)
        return NetworksForward.apply(env, f(ag["transmission"]).to(device), f(ag["susceptibility"]).to(device),
                                     *[n.log_beta for n in active_networks])
    bufs = agent_buffers(engine, data, need_params=False, need_stage=has_q, need_infection_state=False)
    n = engine.plan.host.n_agents
    out = torch.empty(n, dtype=torch.float32, device=device)
    # q*transmission is normally produced by the transmission kernel; callers of this entry set
    # data["agent"].transmission themselves (the reference's InfectionNetworks contract)
    engine.quarantine_transmission(bufs, params)
    engine.venue_reduce(bufs, params)
    io = engine.io(not_infected_probs=out) if want == "probs" else engine.io(trans_susc=out)
    engine.agent_gather(bufs, params, io, sample=False)
    return out


class InfectionNetworks(torch.nn.Module):
    def __init__(self, device="cuda:0", **kwargs):
        super().__init__()
        self.networks = torch.nn.ModuleDict(kwargs)
        self.device = device

    def __getitem__(self, item):
        return self.networks[item]

    @classmethod
    def from_parameters(cls, params):
        this = sys.modules[__name__]
        nets = {}
        for key in params["networks"]:
            cls_name = "".join(w.title() for w in key.split("_")) + "Network"
            nets[key] = getattr(this, cls_name).from_parameters(params)
        return cls(device=params["system"]["device"], **nets)

    @classmethod
    def from_file(cls, fpath=None):
        if fpath is None:
            from .defaults import default_parameters

            return cls.from_parameters(default_parameters())
        with open(fpath) as f:
            return cls.from_parameters(yaml.safe_load(f))

    def active_networks(self, timer, policies) -> List[InfectionNetwork]:
        """The step's networks in accumulation order (timer hierarchy, minus closed venues)."""
        order = timer.get_activity_order()
        cv = getattr(policies, "close_venue_policies", None)
        if cv:
            order = cv.apply(edge_types=order, timer=timer)
        return [self.networks[a] for a in order]

    def forward(self, data, timer, policies):
        """Per-agent probability of NOT being infected in this step (reference base.py:118-141)."""
        return _run_networks(self.active_networks(timer, policies), data, timer, policies, self.device,
                             want="probs", all_networks=list(self.networks.values()))
