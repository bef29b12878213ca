"""CPU: the oracle must reproduce every stage the reference recorded (bit for bit: same ATen ops)."""
import numpy as np
import pytest
import torch

import gj_oracle as O
import gj_testlib as L


def check_step(npz, prefix, world, tables):
    rec = L.step_record(npz, prefix)
    sc = L.step_scalars(rec)
    out = O.hot_path_step(world, L.pre_state(rec), leisure_tables=tables,
                          exp_noise=torch.from_numpy(rec["exp_noise"]), return_intermediates=True, **sc)
    assert np.array_equal(out["transmission"].numpy(), rec["transmission"], equal_nan=True)
    for n in sc["active"]:
        assert np.array_equal(out["cum_" + n].numpy(), rec["cum/" + n]), n
        assert np.array_equal(out["ts_" + n].numpy(), rec["ts/" + n]), n
    assert np.array_equal(out["not_infected_probs"].numpy(), rec["not_infected_probs"])
    assert np.array_equal(out["new_infected"].numpy(), rec["new_infected"])
    for k in ("susceptibility", "is_infected", "infection_time"):
        assert np.array_equal(out[k].numpy(), rec["post/" + k]), k
    if "qmask" in rec:
        assert np.array_equal(out["qmask"].numpy(), rec["qmask"])


def test_c100_variants():
    npz = L.load_npz("c100.npz")
    world = L.world_from(npz)
    for v in str(npz["variants"]).split(","):
        check_step(npz, v + "/", world, None)


@pytest.mark.parametrize("name", ["june769.npz", "june769_hot.npz", "synth10k.npz"])
def test_trajectories(name):
    npz = L.load_npz(name)
    world = L.world_from(npz)
    tables = L.tables_from(npz)
    for i in range(int(npz["n_steps"])):
        check_step(npz, f"step{i}/", world, tables)


def test_chained_counts_june769():
    """Oracle carried forward over the 15 steps (recorded symptom stage injected) reproduces the
    reference's cases_per_timestep."""
    for name in ("june769.npz", "june769_hot.npz"):
        npz = L.load_npz(name)
        world = L.world_from(npz)
        tables = L.tables_from(npz)
        st = L.pre_state(L.step_record(npz, "step0/"))
        cases = [float(st["is_infected"].sum())]
        for i in range(int(npz["n_steps"])):
            rec = L.step_record(npz, f"step{i}/")
            st["current_stage"] = torch.from_numpy(rec["pre/current_stage"])
            out = O.hot_path_step(world, st, leisure_tables=tables, exp_noise=torch.from_numpy(rec["exp_noise"]),
                                  **L.step_scalars(rec))
            for k in ("susceptibility", "is_infected", "infection_time"):
                st[k] = out[k]
            cases.append(float(st["is_infected"].sum()))
        assert np.array_equal(np.array(cases, dtype=np.float32), npz["cases_per_timestep"])


def test_sampler_matches_torch_gumbel_softmax():
    """sample_infected(noise) == F.gumbel_softmax on the same generator stream."""
    torch.manual_seed(5)
    p = torch.rand(4096).clamp(1e-6, 1 - 1e-6)
    state = torch.get_rng_state()
    a = O.sample_infected_torch(p)
    torch.set_rng_state(state)
    noise = O.draw_exp_noise(p.numel())
    b = O.sample_infected(p, noise)
    assert torch.equal(a, b)
    assert set(np.unique(b.numpy()).tolist()) <= {0.0, 1.0}


def test_fp64_mode_is_close_to_fp32():
    npz = L.load_npz("synth10k.npz")
    world = L.world_from(npz)
    tables = L.tables_from(npz)
    rec = L.step_record(npz, "step0/")
    sc = L.step_scalars(rec)
    o64 = O.hot_path_step(world, L.pre_state(rec), leisure_tables=tables, dtype=torch.float64,
                          exp_noise=torch.from_numpy(rec["exp_noise"]), **sc)
    assert np.abs(o64["not_infected_probs"].numpy() - rec["not_infected_probs"]).max() < 2e-6
