"""GPU: the multi-GPU step on ONE device.  R "virtual ranks" each compile their part of the world
(halo agents, partial-sum sets) and run the real kernels; the two collectives are stood in by
device-side copies / sums between the ranks' buffers.  The result must equal the single-rank run:
probabilities to 1e-6, decisions and infection counts exactly (Philox is keyed by global agent id)."""
import numpy as np
import pytest
import torch

import bench as B
from grad_june_amd.benchrun import SingleGpuHotPath
from grad_june_amd.distributed import DistributedHotPath, partition_bounds
from grad_june_amd.synthetic import make_world

pytestmark = pytest.mark.gpu


class VirtualRank(DistributedHotPath):
    """DistributedHotPath whose collectives are performed by the test across in-process ranks."""

    def step_until_exchange(self):
        self.p = self.params()
        self.engine.step_phase(self.bufs, self.p, self.io, 0)

    def step_after_halo(self):
        self.engine.step_phase(self.bufs, self.p, self.io, 1)
        self.engine.step_phase(self.bufs, self.p, self.io, 5)

    def step_after_reduce(self):
        self.engine.step_phase(self.bufs, self.p, self.io, 6)
        self.engine.step_phase(self.bufs, self.p, self.io, 3)
        self.t += 1


@pytest.mark.parametrize("R", [2, 4])
def test_virtual_ranks_match_single_rank(device, R):
    world = make_world("c3", n_agents=40_000, seed=3, infected_fraction=0.05)
    specs, betas = B.network_specs(world), B.betas_of(world)
    single = SingleGpuHotPath(world, specs, betas, device, seed=7, layout="tiled")
    ranks = [VirtualRank(world, specs, betas, device, r, R, seed=7, collectives=False) for r in range(R)]
    assert {m for rk in ranks for m in rk.rw.modes.values()} == {"halo", "partial"}
    b = partition_bounds(world["n_agents"], R)
    for step in range(3):
        single.step()
        for rk in ranks:
            rk.step_until_exchange()
        # halo all-to-all stand-in: every rank's halo slots <- the owners' fresh transmissions
        glob = torch.cat([rk.state["transmission"][: rk.rw.n_local] for rk in ranks])
        for rk in ranks:
            idx = torch.from_numpy(rk.rw.halo_global).to(device)
            rk.state["transmission"][rk.rw.n_local_pad:rk.rw.n_local_pad + rk.rw.n_halo] = glob[idx]
        for rk in ranks:
            rk.step_after_halo()
        # all-reduce stand-in over the flat partial-sum buffers
        total = torch.stack([rk.flat_cum for rk in ranks]).sum(0)
        for rk in ranks:
            rk.flat_cum.copy_(total)
        for rk in ranks:
            rk.step_after_reduce()
        torch.cuda.synchronize()
        for k in ("is_infected", "susceptibility", "infection_time"):
            got = torch.cat([rk.state[k] for rk in ranks]).cpu().numpy()
            ref = single.state[k].cpu().numpy()
            assert np.array_equal(got, ref), f"step {step}: {k}"
        got_new = torch.cat([rk.new_infected for rk in ranks]).cpu().numpy()
        assert np.array_equal(got_new > 0.5, single.new_infected.cpu().numpy() > 0.5)
    assert single.state["is_infected"].sum().item() > 0.05 * world["n_agents"]
