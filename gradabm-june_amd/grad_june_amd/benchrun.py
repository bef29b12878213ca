"""Hot-path stepping on one GPU for bench.py and the large-size tests: plan upload, state upload,
a step loop with HIP events between the launches (on the stream the kernels run on)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence

import numpy as np
import torch

from . import _native as N
from .engine import AgentBuffers, InfectionEngine
from .plan import DevicePlan, NetworkSpec, compile_plan


class EventLog:
    """hipEvent marks recorded during the timed region, resolved after it."""

    def __init__(self):
        self.lib = N.load()
        self.marks: List[tuple] = []

    def mark(self, label: str):
        e = C.c_void_p()
        N.check(self.lib.gj_event_create(C.byref(e)), "gj_event_create")
        N.check(self.lib.gj_event_record(e, N.current_stream()), "gj_event_record")
        self.marks.append((label, e))

    def spans(self) -> Dict[str, List[float]]:
        """label of mark i = the kernel launched between mark i-1 and mark i."""
        out: Dict[str, List[float]] = {}
        ms = C.c_float()
        for (_, a), (label, b) in zip(self.marks[:-1], self.marks[1:]):
            if label == "begin":
                continue
            N.check(self.lib.gj_event_elapsed_ms(a, b, C.byref(ms)), "gj_event_elapsed_ms")
            out.setdefault(label, []).append(float(ms.value))
        return out

    def clear(self):
        for _, e in self.marks:
            self.lib.gj_event_destroy(e)
        self.marks = []


class SingleGpuHotPath:
    PHASES = {"csr": ((0, "transmission"), (1, "venue_reduce"), (3, "agent_gather")),
              "tiled": ((0, "transmission"), (1, "tile_scatter"), (2, "tile_venues"), (3, "tile_agents"))}

    def __init__(self, world: dict, specs: Sequence[NetworkSpec], betas: Dict[str, float], device,
                 seed: int = 0, quarantine_threshold=None, exp_noise=None, layout: str = "tiled", progress=None,
                 split_epilogue: bool = False, device_compile: bool = False, **plan_kw):
        self.device = torch.device(device)
        self.layout = layout
        if device_compile:           # build the tiled arrays on the GPU (the compile kernels, tiling_native), not with numpy on the host
            plan_kw["device"] = self.device
        host = compile_plan(world["n_agents"], world["edge_sets"], age=world["age"], sex=world["sex"],
                            layout=layout, progress=progress, **plan_kw)
        self.engine = InfectionEngine(DevicePlan(host, specs, self.device, split_epilogue=split_epilogue))
        self.networks = list(world["networks"])
        self.betas = betas
        self.seed = seed
        self.q_thr = quarantine_threshold
        st = world["state"]
        self.state = {k: torch.from_numpy(np.ascontiguousarray(v)).to(self.device) for k, v in st.items()}
        A = world["n_agents"]
        self.state["transmission"] = torch.zeros(A, dtype=torch.float32, device=self.device)
        self.new_infected = torch.empty(A, dtype=torch.float32, device=self.device)
        self.probs = torch.empty(A, dtype=torch.float32, device=self.device)
        s = self.state
        self.bufs = AgentBuffers(self.engine.plan, max_infectiousness=s["max_infectiousness"], shape=s["shape"],
                                 rate=s["rate"], shift=s["shift"], infection_time=s["infection_time"],
                                 is_infected=s["is_infected"], susceptibility=s["susceptibility"],
                                 transmission=s["transmission"], current_stage=s["current_stage"])
        import os

        self.stamps = torch.zeros(A, dtype=torch.float32, device=self.device) if os.environ.get("GJ_DIAG_STAMPS") else None
        self.io = self.engine.io(not_infected_probs=self.probs, new_infected=self.new_infected, exp_noise=exp_noise,
                                 trans_susc=self.stamps)
        self.t = 0
        self.log = EventLog()

    def load_state(self, state: Dict[str, np.ndarray]) -> None:
        """Overwrite the per-agent arrays in place (same world, another epidemic state) and restart the clock."""
        for k, v in state.items():
            self.state[k].copy_(torch.from_numpy(np.ascontiguousarray(v)))
        self.state["transmission"].zero_()
        self.t = 0

    def params(self):
        has_q = self.q_thr is not None
        return self.engine.params(now=1.0 + self.t, delta_time=1.0, day_type=0, active=self.networks,
                                  betas=self.betas, has_quarantine=has_q,
                                  q_threshold=self.q_thr if has_q else float("inf"), seed=self.seed, step=self.t)

    # ---- one captured step replayed: a single host call per timestep ------------------------------------
    def capture(self, delta_now: float = 1.0):
        """Capture the step (clock advance + the launches of gj_step) in a hipGraph; ``step()`` then replays it.
        The step's scalars that change - ``now`` and the Philox stream id - live in device memory (StepClock)."""
        from .engine import StepClock

        self.clock = StepClock(self.device)
        self._delta_now = float(delta_now)
        p = self._graph_params = self.params()
        p.clock = self.clock.ptr
        self.clock.set(p.now - delta_now, self.t - 1)              # the first replay advances to (now, t)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.clock.advance(delta_now)
            self.engine.step(self.bufs, p, self.io)
        return self.graph

    def step(self, timed: bool = False):
        if getattr(self, "graph", None) is not None and not timed:
            self.graph.replay()
            self.t += 1
            return
        if getattr(self, "graph", None) is not None:      # an eager step of a captured runner: keep the device clock in step
            self.clock.advance(self._delta_now)
        p = self.params()
        e = self.engine
        if not timed:
            e.step(self.bufs, p, self.io)
        else:
            self.log.mark("begin")
            for phase, label in self.PHASES[self.layout]:
                e.step_phase(self.bufs, p, self.io, phase)
                self.log.mark(label)
        self.t += 1

    # ---- "full step" of SURVEY section 8d: hot path + symptoms (f1) + result reductions (f2) ----------
    def enable_full_step(self, n_rows: int = 4096):
        from .defaults import default_parameters
        from .symptoms import SymptomsSampler

        dev = self.device
        A = self.engine.plan.host.n_agents
        self._sym = SymptomsSampler.from_parameters(default_parameters(str(dev)))
        self._sym_p = self._sym.kernel_params()
        self._sym_table = self._sym.stage_transition_probabilities.to(dev).contiguous()
        self._sym_p.progress = self._sym_table.data_ptr()
        st = self.state
        st["next_stage"] = torch.where(st["is_infected"] > 0, st["current_stage"] + 1, st["current_stage"]).contiguous()
        st["time_to_next_stage"] = torch.zeros(A, dtype=torch.float32, device=dev)
        self._series = torch.zeros(n_rows, 5, dtype=torch.float64, device=dev)
        self._n_rows = n_rows
        self._edges = (C.c_int32 * 4)(0, 18, 65, 100)
        self._row = 0

    def full_step(self):
        lib = N.load()
        self.step()
        st = self.state
        p = self._sym_p
        p.time, p.seed, p.step, p.agent_offset = float(self.t), self.seed, self.t, 0
        row = self._series[self._row % self._n_rows]
        if self._row >= self._n_rows:        # the ring wrapped: the kernel ACCUMULATES into its row (atomics), so zero it first
            row.zero_()
        N.check(lib.gj_symptoms_step_stats(self.new_infected.numel(), N.ptr(self.engine.plan.agent_class),
                                           N.ptr(self.new_infected), N.ptr(st["current_stage"]), N.ptr(st["next_stage"]),
                                           N.ptr(st["time_to_next_stage"]), C.byref(p), None, None,
                                           N.ptr(st["is_infected"]), 3, self._edges, 7,
                                           N.ptr(row), N.current_stream()),
                "gj_symptoms_step_stats")
        self._row += 1

    def reset_timers(self):
        self.log.clear()

    def kernel_ms(self) -> Dict[str, float]:
        return {k: float(np.mean(v)) for k, v in self.log.spans().items()}

    def time_stateless(self, steps: int = 10, repeats: int = 3) -> float:
        """ms per pass of transmission + both sparse passes + probabilities (no decision, no state update):
        what ``tune_geometry`` compares.  Leaves the epidemic state untouched.  The fastest of ``repeats`` loops of
        ``steps`` passes: a loop lasts about a millisecond of wall clock, and one host hiccup inside it (round 2 and 3
        each saw a candidate "measure" 6-7 ms per step that runs at 0.11 - profiles/r03_geometry_cliff_c2_candidate.txt)
        would otherwise decide the race."""
        import time

        p = self.params()
        phases = (0, 1, 2, 4) if self.layout == "tiled" else (0, 1, 4)
        for _ in range(2):
            for ph in phases:
                self.engine.step_phase(self.bufs, p, self.io, ph)
        best = float("inf")
        for _ in range(max(1, repeats)):
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(steps):
                for ph in phases:
                    self.engine.step_phase(self.bufs, p, self.io, ph)
            torch.cuda.synchronize(self.device)
            best = min(best, 1e3 * (time.perf_counter() - t0) / steps)
        return best


#: tile geometries tried by tune_geometry: {} = the size-based defaults of tiling.py
GEOMETRY_CANDIDATES = ({}, {"eb_target": 131072, "sv_max": 16384}, {"eb_target": 65536, "sv_max": 16384},
                       {"eb_target": 131072, "sv_max": 16384, "slice_agents": 2048},
                       {"eb_target": 32768, "sv_max": 16384, "slice_agents": 2048},
                       {"eb_target": 131072, "sv_max": 16384, "direct": False},
                       # (round 4) more, smaller venue blocks: a world of this size has ~1 block per CU otherwise
                       {"eb_target": 32768, "sv_max": 16384}, {"eb_target": 32768, "sv_max": 8192},
                       {"eb_target": 16384, "sv_max": 4096})


def tune_geometry(world: dict, specs, betas, device, candidates=GEOMETRY_CANDIDATES, progress=None, **kw):
    """Compile the world under each candidate geometry, time the stateless part of the step on the device
    and return ``(best SingleGpuHotPath, {label: ms})``.  The size-based defaults are right for the worlds
    they were measured on (1.5 memberships per agent and edge set); worlds with much denser membership
    (BASELINE's C2: 5 per agent and set) prefer larger tiles - measuring is cheaper than modelling, and
    compiling a world of a few 10^7 edges takes seconds."""
    best, best_ms, seen = None, float("inf"), {}
    for cand in candidates:
        cand = dict(cand)
        sa = cand.pop("slice_agents", None)
        if sa is not None:        # twice the default number of slices: phase A likes it, phase D does not - measure
            cand["slices"] = (-(-world["n_agents"] // sa), sa)
        r = SingleGpuHotPath(world, specs, betas, device, progress=None, **{**kw, **cand})
        ms = r.time_stateless()
        label = ",".join(f"{k}={v}" for k, v in cand.items()) or "default"
        label = label.replace(" ", "")
        seen[label] = ms
        if progress:
            progress(f"geometry {label}: {ms:.3f} ms")
        if ms < best_ms:
            best, best_ms = r, ms
        else:
            del r
            torch.cuda.empty_cache()
    return best, seen
