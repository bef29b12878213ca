"""Experiment driver (reference grad_june/runner.py:15-242): builds model + world + timer from the
YAML schema, seeds infections, runs the time loop and collects the result series.

Same public surface as the reference's ``Runner``.  Per-step state lives on the HIP device; the
hot path of each step is ``GradJune.hot_path`` (HIP kernels).  The per-step result reductions
(cases, cases by age, deaths) are small device reductions written into preallocated series
 instead of the reference's growing hstack/vstack chains.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
import yaml

from .graph import HeteroData, ToUndirected, load_world
from .infection import infect_fraction_of_people
from .model import GradJune
from .timer import Timer
from .transmission import TransmissionSampler
from .utils import read_path
from .world import require_hip

EDGE_SETS = ("household", "company", "school", "university", "care_home", "leisure")


def world_from_npz(path) -> HeteroData:
    """Neutral-array world (tests/golden/make_golden.py layout) -> the reference's graph format."""
    with np.load(path, allow_pickle=False) as z:
        arrays = {k: z[k] for k in z.files}
    data = HeteroData()
    ag = data["agent"]
    n = int(arrays["n_agents"])
    ag.id = torch.from_numpy(arrays.get("agent/id", np.arange(n)))
    ag.age = torch.from_numpy(arrays["age"])
    ag.sex = torch.from_numpy(arrays["sex"])
    for key in ("ethnicity", "area"):
        if "agent/" + key in arrays:
            ag[key] = arrays["agent/" + key]
    for s in EDGE_SETS:
        if f"es/{s}/agent" not in arrays:
            continue
        data[s].id = torch.from_numpy(arrays.get(f"venue_id/{s}", np.arange(len(arrays[f"es/{s}/people"]))))
        data[s].people = torch.from_numpy(arrays[f"es/{s}/people"])
        data["agent", "attends_" + s, s].edge_index = torch.from_numpy(
            np.vstack((arrays[f"es/{s}/agent"], arrays[f"es/{s}/venue"])))
    return ToUndirected()(data)


class Runner(torch.nn.Module):
    def __init__(self, model, data, timer, log_fraction_initial_cases, save_path, parameters,
                 age_bins=(0, 18, 65, 100)):
        super().__init__()
        self.model = model
        self.data = data
        self.data_backup = self.backup_infection_data(data)
        self.timer = timer
        self.log_fraction_initial_cases = log_fraction_initial_cases
        self.device = model.device
        self.age_bins = torch.tensor(age_bins, device=self.device)
        eth = data["agent"].get("ethnicity", None)
        self.ethnicities = np.sort(np.unique(eth)) if eth is not None else np.zeros(0)
        self.n_agents = data["agent"].id.shape[0]
        self.population_by_age = self.get_people_by_age()
        self.save_path = Path(save_path)
        self.input_parameters = parameters
        self.restore_initial_data()

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
            model=GradJune.from_parameters(params),
            data=cls.get_data(params),
            timer=Timer.from_parameters(params),
            log_fraction_initial_cases=params["infection_seed"]["log_fraction_initial_cases"],
            save_path=params["save_path"],
            parameters=params,
            age_bins=params.get("age_bins_to_save", (0, 18, 65, 100)),
        )

    @staticmethod
    def get_data(params):
        device = require_hip(params["system"]["device"])
        path = read_path(params["data_path"])
        data = world_from_npz(path) if str(path).endswith(".npz") else load_world(path)
        # optional (not a reference key): ``system.locality_order: household`` renumbers the agents venue-major
        # at load time (graph.locality_order); the per-agent result of ``runner()`` is reported in the
        # file's original order
        by = params["system"].get("locality_order")
        if by:
            from .graph import locality_order

            data, original = locality_order(data, by=by)
            data["agent"].original_index = original
        data = data.to(device)
        n = len(data["agent"]["id"])
        values = TransmissionSampler.from_parameters(params)(n)
        ag = data["agent"]
        ag.infection_parameters = {"max_infectiousness": values[0].contiguous(), "shape": values[1].contiguous(),
                                   "rate": values[2].contiguous(), "shift": values[3].contiguous()}
        ag.transmission = torch.zeros(n, device=device)
        ag.susceptibility = torch.ones(n, device=device)
        ag.is_infected = torch.zeros(n, device=device)
        ag.infection_time = torch.zeros(n, device=device)
        ag.symptoms = {"current_stage": torch.ones(n, dtype=torch.long, device=device),
                       "next_stage": torch.ones(n, dtype=torch.long, device=device),
                       "time_to_next_stage": torch.zeros(n, device=device)}
        return data

    # in-memory checkpoint of the mutable state ---------------------------------------------------
    _STATE = ("susceptibility", "is_infected", "infection_time", "transmission")
    _SYMPTOMS = ("current_stage", "next_stage", "time_to_next_stage")

    def backup_infection_data(self, data):
        ag = data["agent"]
        ret = {k: ag[k].detach().clone() for k in self._STATE}
        ret["symptoms"] = {k: ag["symptoms"][k].detach().clone() for k in self._SYMPTOMS}
        return ret

    def restore_initial_data(self):
        ag = self.data["agent"]
        for k in self._STATE:
            ag[k] = self.data_backup[k].detach().clone()
        for k in self._SYMPTOMS:
            ag.symptoms[k] = self.data_backup["symptoms"][k].detach().clone()
        self.data["results"] = {"deaths_per_timestep": None}

    def set_initial_cases(self):
        new_infected = infect_fraction_of_people(
            data=self.data, timer=self.timer, symptoms_updater=self.model.symptoms_updater,
            device=self.device, fraction=10.0 ** self.log_fraction_initial_cases,
            agent_offset=getattr(self, "agent_offset", 0))
        self.model.symptoms_updater(data=self.data, timer=self.timer, new_infected=new_infected)

    # per-step result reductions: one fused pass (gj_step_stats) into a preallocated series ------------
    def _stats_args(self, data):
        """(agent classes, age-bin edges as a C array, dead stage) of gj_step_stats / gj_symptoms_step_stats."""
        import ctypes as C

        ag = data["agent"]
        if getattr(self, "_cls", None) is None:
            dev = require_hip(self.device)
            sex = ag["sex"] if "sex" in ag else torch.zeros_like(ag.age)
            self._cls = (sex.long() * 100 + ag.age.long()).to(device=dev, dtype=torch.uint8).contiguous()
            self._edges = (C.c_int32 * (len(self.age_bins)))(*[int(b) for b in self.age_bins.cpu()])
        return self._cls, self._edges, int(self.model.symptoms_updater.stages_ids[-1])

    def _record(self, data, row):
        from . import _native as N

        ag = data["agent"]
        n = self.n_agents
        cls, edges, dead = self._stats_args(data)
        stage = ag.symptoms["current_stage"]
        if stage.dtype != torch.float32:
            stage = stage.to(torch.float32)
        inf = ag.is_infected
        if inf.dtype != torch.float32 or not inf.is_contiguous():
            inf = inf.to(torch.float32).contiguous()
        N.check(N.load().gj_step_stats(n, N.ptr(cls), N.ptr(inf), N.ptr(stage.contiguous()),
                                       len(self.age_bins) - 1, edges, dead, N.ptr(self._series[row]),
                                       N.current_stream()), "gj_step_stats")

    # time loop --------------------------------------------------------------------------------------
    def forward(self):
        timer, model, data = self.timer, self.model, self.data
        timer.reset()
        self.restore_initial_data()
        self.set_initial_cases()
        n_bins = len(self.age_bins) - 1
        n_rows = 1
        probe = Timer.from_parameters(self.input_parameters) if isinstance(self.input_parameters, dict) else None
        if probe is not None:
            while probe.date < probe.final_date:
                next(probe)
                n_rows += 1
        else:
            n_rows = 4096
        self._series = torch.zeros(n_rows, 2 + n_bins, dtype=torch.float64, device=require_hip(self.device))
        # differentiable run (a log_beta is an nn.Parameter, grad mode on): the case series must stay on
        # the autograd graph, so they are formed with tensor ops instead of the fused reduction kernel
        differentiable = torch.is_grad_enabled() and any(
            isinstance(n.log_beta, torch.Tensor) and n.log_beta.requires_grad
            for n in model.infection_networks.networks.values())
        diff_rows = []

        def record(row, done=False):
            if not done:
                self._record(data, row)
            if differentiable:
                ag = data["agent"]
                stage = ag.symptoms["current_stage"]
                dead = float(self.model.symptoms_updater.stages_ids[-1])
                deaths = ((stage == dead) * stage / dead).sum()          # store_differentiable_deaths' form
                diff_rows.append(torch.cat((ag.is_infected.sum().reshape(1), self.get_cases_by_age(data),
                                            deaths.reshape(1))))

        record(0)
        dates = [timer.date]
        row = 0
        while timer.date < timer.final_date:
            next(timer)
            row += 1
            sink = None
            if not differentiable:      # rows f1 + f2 in one pass: the model's symptoms update fills this step's row
                cls, edges, dead = self._stats_args(data)
                sink = {"cls": cls, "edges": edges, "n_bins": n_bins, "dead": dead, "out": self._series[row]}
            model.step_stats = sink
            try:
                data = model(data, timer)
            finally:
                model.step_stats = None
            record(row, done=bool(sink and sink.get("done")))
            dates.append(timer.date)
        self._finalize_series(row + 1)
        series = self._series[: row + 1].to(torch.float32)
        if differentiable:
            series = self._reduce_differentiable(torch.stack(diff_rows).to(torch.float32))
        cases_per_timestep = series[:, 0]
        data["results"]["deaths_per_timestep"] = series[:, 1 + n_bins]
        results = {
            "dates": dates,
            "cases_per_timestep": cases_per_timestep,
            "daily_cases_per_timestep": torch.diff(cases_per_timestep,
                                                   prepend=torch.tensor([0.0], device=self.device)),
            "deaths_per_timestep": data.results["deaths_per_timestep"],
        }
        for i, key in enumerate(self.age_bins[1:]):
            results[f"cases_by_age_{int(key):02d}"] = series[:, 1 + i]
        is_infected = data["agent"].is_infected
        if "original_index" in data["agent"]:
            out = torch.empty_like(is_infected)
            out[data["agent"].original_index.to(out.device)] = is_infected
            is_infected = out
        return results, is_infected

    def _finalize_series(self, n_rows: int) -> None:
        """Hook: a partitioned run sums the ranks' per-step reductions here (distributed_api.DistributedRunner)."""

    def _reduce_differentiable(self, series: torch.Tensor) -> torch.Tensor:
        """Hook: the same for the series kept on the autograd graph in a differentiable run."""
        return series

    def save_results(self, results, is_infected):
        import pandas as pd

        self.save_path.mkdir(exist_ok=True, parents=True)
        df = pd.DataFrame(index=results["dates"])
        df.index.name = "date"
        for key, series in results.items():
            if key != "dates":
                df[key] = series.detach().cpu().numpy()
        df.to_csv(self.save_path / "results.csv")
        pd.DataFrame({"is_infected": is_infected.detach().cpu().numpy()}).to_csv(
            self.save_path / "results_is_infected.csv")

    def store_differentiable_deaths(self, data):
        stage = data["agent"].symptoms["current_stage"]
        dead = int(self.model.symptoms_updater.stages_ids[-1])
        deaths = ((stage == dead) * stage / dead).sum()
        prev = data["results"]["deaths_per_timestep"]
        data["results"]["deaths_per_timestep"] = deaths if prev is None else torch.hstack((prev, deaths))

    def _age_masks(self, ages):
        lo, hi = self.age_bins[:-1], self.age_bins[1:]
        return (ages[None, :] > lo[:, None]) & (ages[None, :] < hi[:, None])     # [bins, A], open intervals

    def get_cases_by_age(self, data):
        return (self._age_masks(data["agent"].age) * data["agent"].is_infected[None, :]).sum(1)

    def get_people_by_age(self):
        counts = self._age_masks(self.data["agent"].age).sum(1)
        return {int(self.age_bins[i + 1].item()): counts[i] for i in range(len(counts))}

    def get_cases_by_ethnicity(self, data):
        ret = torch.zeros(len(self.ethnicities), device=self.device)
        for i, ethnicity in enumerate(self.ethnicities):
            mask = torch.tensor(self.data["agent"].ethnicity == ethnicity, device=self.device)
            ret[i] = (mask * data["agent"].is_infected).sum()
        return ret
