"""Per-timestep driver of the HIP kernels: owns workspaces, fills ``gj_step_params``, launches.

This is the seam between the reference-shaped Python API (``InfectionNetworks``, ``GradJune``) and
the C ABI.  It keeps no model logic: which networks are active, their beta and the quarantine
threshold are decided by the host mirror of the reference's timer / policies and handed in.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Sequence

import torch

from . import _native as N
from .plan import DevicePlan


class AgentBuffers:
    """Device pointers of the per-agent state for one call (tensors stay owned by the caller)."""

    _F32 = ("max_infectiousness", "shape", "rate", "shift", "infection_time", "is_infected",
            "susceptibility", "transmission")

    def __init__(self, plan: DevicePlan, *, max_infectiousness=None, shape=None, rate=None, shift=None,
                 infection_time=None, is_infected=None, susceptibility, transmission, q_transmission=None,
                 current_stage=None):
        n, n_ext = plan.host.n_agents, plan.host.n_ext_agents
        self.tensors = dict(max_infectiousness=max_infectiousness, shape=shape, rate=rate, shift=shift,
                            infection_time=infection_time, is_infected=is_infected,
                            susceptibility=susceptibility, transmission=transmission,
                            q_transmission=q_transmission, current_stage=current_stage)
        for k, t in self.tensors.items():
            if t is None:
                continue
            want = n_ext if k in ("transmission", "q_transmission") else n
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != plan.device or t.numel() != want:
                raise ValueError(f"{k}: need contiguous float32[{want}] on {plan.device}, got "
                                 f"{t.dtype}[{tuple(t.shape)}] on {t.device}")
        c = N.AgentState()
        for k, t in self.tensors.items():
            setattr(c, k, N.ptr(t))
        self.c = c


class StepClock:
    """``gj_clock`` in device memory: the two scalars that change from step to step (``now``, the Philox ``step``).
    A step captured in a hipGraph with ``params.clock`` set is replayed for the following timesteps: the graph's first
    node advances the clock (``advance``)."""

    def __init__(self, device):
        self.lib = N.load()
        # {float now; float pad; uint64 step; double now0; uint64 step0}
        self.buf = torch.zeros(4, dtype=torch.int64, device=device)

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr()

    def set(self, now: float, step: int) -> None:
        import numpy as np

        host = np.zeros(4, dtype=np.int64)
        host.view(np.float32)[0] = now
        host[1] = step
        host.view(np.float64)[2] = now          # the origin later advances are measured from
        host[3] = step
        self.buf.copy_(torch.from_numpy(host))

    def read(self):
        host = self.buf.cpu().numpy()
        return float(host.view("float32")[0]), int(host[1])

    def advance(self, delta_now: float) -> None:
        """step += 1; now = float32(now0 + (step - step0) * delta_now), evaluated in double on the device."""
        N.check(self.lib.gj_clock_advance(self.ptr, float(delta_now), N.current_stream()), "gj_clock_advance")


class InfectionEngine:
    def __init__(self, plan: DevicePlan):
        self.lib = N.load()
        self.plan = plan
        self._q_trans: Optional[torch.Tensor] = None

    # -- parameter marshalling --------------------------------------------------------------
    def q_transmission_buffer(self) -> torch.Tensor:
        if self._q_trans is None:
            self._q_trans = torch.zeros(self.plan.host.n_ext_agents, dtype=torch.float32, device=self.plan.device)
        return self._q_trans

    def params(self, *, now: float, delta_time: float, day_type: int, active: Sequence[str],
               betas: Dict[str, float], has_quarantine: bool = False, q_threshold: float = math.inf,
               seed: int = 0, step: int = 0, agent_offset: int = 0) -> N.StepParams:
        """``active``: network names already in accumulation order (activity hierarchy)."""
        if len(active) > N.GJ_MAX_NETS:
            raise ValueError(f"at most {N.GJ_MAX_NETS} active networks")
        p = N.StepParams()
        p.now, p.delta_time, p.day_type = float(now), float(delta_time), int(day_type)
        p.has_quarantine = 1 if has_quarantine else 0
        p.q_threshold = float(q_threshold)
        p.seed, p.step, p.agent_offset = int(seed), int(step), int(agent_offset)
        p.n_nets = len(active)
        for i, name in enumerate(active):
            spec = self.plan.networks[name]
            p.nets[i].beta = float(betas[name])
            p.nets[i].set = self.plan.host.set_index[spec.edge_set]
            p.nets[i].mask_kind = spec.mask_kind
            p.nets[i].table = self.plan.table_index.get(name, -1)
        return p

    @staticmethod
    def io(not_infected_probs=None, new_infected=None, exp_noise=None, trans_susc=None, agent_sums=None) -> N.StepIO:
        io = N.StepIO()
        io.not_infected_probs = N.ptr(not_infected_probs)
        io.new_infected = N.ptr(new_infected)
        io.exp_noise = N.ptr(exp_noise)
        io.trans_susc = N.ptr(trans_susc)
        io.agent_sums = N.ptr(agent_sums)
        io._keep = (not_infected_probs, new_infected, exp_noise, trans_susc, agent_sums)
        return io

    def _prep(self, bufs: AgentBuffers, p: N.StepParams):
        if p.has_quarantine and bufs.tensors["q_transmission"] is None:
            bufs.tensors["q_transmission"] = self.q_transmission_buffer()
            bufs.c.q_transmission = bufs.tensors["q_transmission"].data_ptr()

    # -- launches (all asynchronous on torch's current stream) -------------------------------
    def transmission_update(self, bufs: AgentBuffers, p: N.StepParams):
        self._prep(bufs, p)
        N.check(self.lib.gj_transmission_update(C.byref(self.plan.c), C.byref(bufs.c), C.byref(p),
                                                N.current_stream()), "gj_transmission_update")

    def quarantine_transmission(self, bufs: AgentBuffers, p: N.StepParams):
        self._prep(bufs, p)
        N.check(self.lib.gj_quarantine_transmission(C.byref(self.plan.c), C.byref(bufs.c), C.byref(p),
                                                    N.current_stream()), "gj_quarantine_transmission")

    def venue_reduce(self, bufs: AgentBuffers, p: N.StepParams):
        self._prep(bufs, p)
        N.check(self.lib.gj_venue_reduce(C.byref(self.plan.c), C.byref(bufs.c), C.byref(p),
                                         N.current_stream()), "gj_venue_reduce")

    def agent_gather(self, bufs: AgentBuffers, p: N.StepParams, io: N.StepIO, sample: bool):
        self._prep(bufs, p)
        N.check(self.lib.gj_agent_gather(C.byref(self.plan.c), C.byref(bufs.c), C.byref(p), C.byref(io),
                                         1 if sample else 0, N.current_stream()), "gj_agent_gather")

    def step(self, bufs: AgentBuffers, p: N.StepParams, io: N.StepIO):
        self._prep(bufs, p)
        N.check(self.lib.gj_step(C.byref(self.plan.c), C.byref(bufs.c), C.byref(p), C.byref(io),
                                 N.current_stream()), "gj_step")

    def step_phase(self, bufs: AgentBuffers, p: N.StepParams, io: N.StepIO, phase: int):
        self._prep(bufs, p)
        N.check(self.lib.gj_step_phase(C.byref(self.plan.c), C.byref(bufs.c), C.byref(p), C.byref(io), int(phase),
                                       N.current_stream()), "gj_step_phase")

    def sample_infect(self, not_infected_probs, *, now: float, susceptibility, is_infected, infection_time,
                      exp_noise=None, new_infected=None, seed: int = 0, step: int = 0, agent_offset: int = 0):
        n = not_infected_probs.numel()
        N.check(self.lib.gj_sample_infect(n, N.ptr(not_infected_probs), N.ptr(exp_noise), int(seed), int(step),
                                          int(agent_offset), float(now), N.ptr(new_infected),
                                          N.ptr(susceptibility), N.ptr(is_infected), N.ptr(infection_time),
                                          N.current_stream()), "gj_sample_infect")


class HipTimer:
    """hipEvent pair on the stream the kernels are launched on (bench.py roofline timing)."""

    def __init__(self):
        self.lib = N.load()
        self.a, self.b = C.c_void_p(), C.c_void_p()
        N.check(self.lib.gj_event_create(C.byref(self.a)), "gj_event_create")
        N.check(self.lib.gj_event_create(C.byref(self.b)), "gj_event_create")

    def start(self):
        N.check(self.lib.gj_event_record(self.a, N.current_stream()), "gj_event_record")

    def stop(self):
        N.check(self.lib.gj_event_record(self.b, N.current_stream()), "gj_event_record")

    def elapsed_ms(self) -> float:
        ms = C.c_float()
        N.check(self.lib.gj_event_elapsed_ms(self.a, self.b, C.byref(ms)), "gj_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            self.lib.gj_event_destroy(self.a)
            self.lib.gj_event_destroy(self.b)
        except Exception:
            pass
