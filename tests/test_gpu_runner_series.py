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


# ---- rows f1 + f2 in one pass (gj_symptoms_step_stats) ---------------------------------------------------------------
def _updater(device):
    from grad_june_amd.defaults import default_parameters
    from grad_june_amd.symptoms import SymptomsUpdater

    return SymptomsUpdater.from_parameters(default_parameters(str(device)))


class _T:
    def __init__(self, now):
        self.now = now


@pytest.mark.parametrize("name", ["june769.npz", "june769_hot.npz"])
def test_fused_symptoms_and_series_reproduce_the_reference(device, name):
    """The reference's recorded trajectories through the fused pass: with its randomness injected the three symptom
    arrays after every step AND the step's row of the Runner's series (cases, cases by age bin, deaths) come out
    exactly - from one kernel that reads the arrays once."""
    import grad_june_amd as G

    z = L.load_npz(name)
    world = L.world_from(z)
    upd = _updater(device)
    d = G.HeteroData()
    d["agent"].age = world["age"].to(device)
    d["agent"].sex = world["sex"].to(device)
    cls = (world["sex"] * 100 + world["age"]).to(torch.uint8).to(device)
    edges_np, dead = z["series/age_bins"], int(z["series/dead_stage"])
    edges = (C.c_int32 * len(edges_np))(*[int(x) for x in edges_np])
    for i in range(int(z["n_steps"])):
        rec = L.step_record(z, f"step{i}/")
        d["agent"].symptoms = {k[8:]: torch.from_numpy(v).to(device) for k, v in rec.items() if k.startswith("sym_pre/")}
        d["agent"].is_infected = torch.from_numpy(z[f"step{i}/post/is_infected"]).to(device)
        out = torch.zeros(5, dtype=torch.float64, device=device)
        sink = {"cls": cls, "edges": edges, "n_bins": 3, "dead": dead, "out": out}
        upd(d, _T(float(rec["now"])), torch.from_numpy(rec["new_infected"]).to(device),
            progresses=torch.from_numpy(rec["sym/progresses"]), dwell=torch.from_numpy(rec["sym/dwell"]), stats=sink)
        assert sink.get("done")
        for k in ("current_stage", "next_stage", "time_to_next_stage"):
            assert np.array_equal(d["agent"].symptoms[k].cpu().numpy(), rec["sym_post/" + k]), (i, k)
        got = out.cpu().numpy()
        assert got[0] == z["cases_per_timestep"][i + 1], i
        assert np.array_equal(got[1:4].astype(np.float32), z["series/cases_by_age"][i + 1]), i
        assert got[4] == z["series/deaths_per_timestep"][i + 1], i


@pytest.mark.parametrize("n", [1_000_003, 4096, 5])
def test_fused_pass_equals_the_two_kernels(device, n):
    """Philox mode, odd sizes (vector body + scalar tail), unaligned views: gj_symptoms_step_stats == gj_symptoms_update
    followed by gj_step_stats, bit for bit, over several steps of a population in every stage."""
    upd = _updater(device)
    sp = upd.symptoms_sampler
    p = sp.kernel_params()
    table = sp.stage_transition_probabilities.to(device=device, dtype=torch.float32).contiguous()
    p.progress = table.data_ptr()
    g = torch.Generator(device="cpu").manual_seed(n)
    cls = (torch.randint(0, 2, (n,), generator=g) * 100 + torch.randint(0, 100, (n,), generator=g)).to(torch.uint8)
    cur = torch.randint(0, 8, (n,), generator=g).float()
    nxt = torch.clamp(cur + torch.randint(0, 2, (n,), generator=g).float(), max=7.0)
    ttn = torch.rand(n, generator=g) * 6.0
    inf = (cur >= 2).float() * (1.0 + (torch.rand(n, generator=g) < 0.01).float())     # the additive is_infected: 2.0 occurs
    edges = (C.c_int32 * 4)(0, 18, 65, 100)
    lib = N.load()

    def run(fused, offset):
        # `offset` floats into a larger buffer: 16-byte alignment present (0) or absent (1)
        def dev(t):
            buf = torch.empty(t.numel() + 4, dtype=t.dtype, device=device)
            view = buf[offset:offset + t.numel()]
            view.copy_(t)
            return view

        c, x, t_, i_, k_ = dev(cur), dev(nxt), dev(ttn), dev(inf), dev(cls)
        rows = []
        for step in range(4):
            new = torch.zeros(n)
            new[(torch.arange(n) * 7 + step) % 11 == 0] = 1.0
            new = dev(new * (cur < 2).float())
            p.time, p.seed, p.step, p.agent_offset = 2.0 + step, 99, step, 12345
            out = torch.zeros(5, dtype=torch.float64, device=device)
            if fused:
                N.check(lib.gj_symptoms_step_stats(n, N.ptr(k_), N.ptr(new), N.ptr(c), N.ptr(x), N.ptr(t_), C.byref(p), None,
                                                   None, N.ptr(i_), 3, edges, 7, N.ptr(out), N.current_stream()), "fused")
            else:
                N.check(lib.gj_symptoms_update(n, N.ptr(k_), N.ptr(new), N.ptr(c), N.ptr(x), N.ptr(t_), C.byref(p), None, None,
                                               N.current_stream()), "symptoms")
                N.check(lib.gj_step_stats(n, N.ptr(k_), N.ptr(i_), N.ptr(c), 3, edges, 7, N.ptr(out), N.current_stream()),
                        "stats")
            torch.cuda.synchronize()
            rows.append(out.cpu().numpy())
        return [v.cpu().numpy() for v in (c, x, t_)], np.stack(rows)

    ref_arrays, ref_rows = run(False, 0)
    assert (ref_arrays[0] != cur.numpy()).mean() > 0.05            # the population does move
    for offset in (0, 1):
        arrays, rows = run(True, offset)
        for a, b in zip(arrays, ref_arrays):
            assert np.array_equal(a, b)
        assert np.array_equal(rows, ref_rows)
    assert ref_rows[:, 0].min() > 0 and ref_rows[-1, 4] > 0
