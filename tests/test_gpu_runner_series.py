"""GPU, row f2: the fused per-step reductions (gj_step_stats) against the REFERENCE Runner's own result series.

tests/golden/june769_series.npz holds the results dict of the reference's unmodified Runner.forward() (runner.py:151-183)
over a 90-day run of the shipped 769-agent world - cases_per_timestep, daily_cases_per_timestep, deaths_per_timestep,
cases_by_age_18/65/100 - next to the per-agent state after every step; june769.npz / june769_hot.npz hold the reference's
get_cases_by_age / store_differentiable_deaths rows for the recorded 15-step trajectories.  Every recorded state goes
through gj_step_stats and every row must come out exactly - including the reference's open age intervals
(lo < age < hi: ages 0, 18 and 65 are in no bin, runner.py:217-224)."""
import ctypes as C

import numpy as np
import pytest
import torch

import gj_testlib as L
from grad_june_amd import _native as N

pytestmark = pytest.mark.gpu


def step_stats(device, cls, is_infected, current_stage, edges, dead_stage):
    n_bins = len(edges) - 1
    out = torch.zeros(2 + n_bins, dtype=torch.float64, device=device)
    inf = torch.from_numpy(np.ascontiguousarray(is_infected, dtype=np.float32)).to(device)
    stage = torch.from_numpy(np.ascontiguousarray(current_stage, dtype=np.float32)).to(device)
    e = (C.c_int32 * len(edges))(*[int(x) for x in edges])
    N.check(N.load().gj_step_stats(inf.numel(), N.ptr(cls), N.ptr(inf), N.ptr(stage), n_bins, e, int(dead_stage), N.ptr(out),
                                   N.current_stream()), "gj_step_stats")
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_runner_result_series_of_the_reference(device):
    z = L.load_npz("june769_series.npz")
    age, sex = z["age"], z["sex"]
    cls = torch.from_numpy((sex * 100 + age).astype(np.uint8)).to(device)
    edges, dead = z["age_bins"], int(z["dead_stage"])
    assert edges.tolist() == [0, 18, 65, 100] and dead == 7
    T = int(z["results/n_dates"])
    assert z["post/is_infected"].shape == (T - 1, 769)
    cases = [float(z["results/cases_per_timestep"][0])]
    for t in range(1, T):
        got = step_stats(device, cls, z["post/is_infected"][t - 1], z["post/current_stage"][t - 1], edges, dead)
        assert got[0] == z["results/cases_per_timestep"][t], t
        for b, key in enumerate((18, 65, 100)):
            assert got[1 + b] == z[f"results/cases_by_age_{key:02d}"][t], (t, key)
        assert got[4] == z["results/deaths_per_timestep"][t], t
        cases.append(got[0])
    daily = np.diff(np.array(cases), prepend=0.0)
    assert np.array_equal(daily.astype(np.float32), z["results/daily_cases_per_timestep"])
    # the run exercises what it should: deaths accumulate, and agents sit exactly on the open bin edges
    assert z["results/deaths_per_timestep"][-1] >= 10 and z["results/deaths_per_timestep"][0] == 0
    on_edge = np.isin(age, (0, 18, 65))
    assert on_edge.sum() > 5 and (z["final/is_infected"][on_edge] > 0).any()
    last = step_stats(device, cls, z["final/is_infected"], z["post/current_stage"][-1], edges, dead)
    assert last[1:4].sum() == (z["final/is_infected"] * ~on_edge).sum() < last[0]


@pytest.mark.parametrize("name", ["june769.npz", "june769_hot.npz"])
def test_recorded_trajectories_by_age_and_deaths(device, name):
    z = L.load_npz(name)
    world = L.world_from(z)
    cls = torch.from_numpy((world["sex"].numpy() * 100 + world["age"].numpy()).astype(np.uint8)).to(device)
    edges, dead = z["series/age_bins"], int(z["series/dead_stage"])
    rows = [(z["seed/is_infected"], z["seed/current_stage"])]
    for i in range(int(z["n_steps"])):
        rows.append((z[f"step{i}/post/is_infected"], z[f"step{i}/sym_post/current_stage"]))
    assert len(rows) == len(z["series/cases_by_age"]) == len(z["cases_per_timestep"])
    for t, (inf, stage) in enumerate(rows):
        got = step_stats(device, cls, inf, stage, edges, dead)
        assert got[0] == z["cases_per_timestep"][t], t
        assert np.array_equal(got[1:4].astype(np.float32), z["series/cases_by_age"][t]), t
        assert got[4] == z["series/deaths_per_timestep"][t], t
    assert np.array_equal(np.diff(z["cases_per_timestep"], prepend=np.float32(0)), z["series/daily_cases_per_timestep"])
