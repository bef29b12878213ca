"""Experiment (GPU): host cost of one step's launch sequence - a tiny world, so the GPU is never the limit.

    python tools/host_overhead.py
"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradabm-june_amd"))
sys.path.insert(0, ROOT)

import torch

import bench as B
import __graft_entry__ as entry
from grad_june_amd.benchrun import SingleGpuHotPath
from grad_june_amd.distributed import DistributedHotPath, choose_modes
from grad_june_amd.synthetic import make_world


def main():
    entry.build()
    world = make_world("c3", n_agents=20000)
    specs, betas = B.network_specs(world), B.betas_of(world)
    dev = torch.device("cuda:0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    import torch.distributed as dist

    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    modes = {k: ("partial" if k != "household" else "local") for k in choose_modes(world, 1)}
    runners = {"single": SingleGpuHotPath(world, specs, betas, dev, seed=1, layout="tiled"),
               "distributed(1 rank), sequential form": DistributedHotPath(world, specs, betas, dev, 0, 1, seed=1, modes=modes),
               "distributed(1 rank), production form, one all-reduce": DistributedHotPath(
                   world, specs, betas, dev, 0, 1, seed=1, modes=modes, production_at_one_rank=True),
               "distributed(1 rank), production form, two all-reduces": DistributedHotPath(
                   world, specs, betas, dev, 0, 1, seed=1, modes=modes, production_at_one_rank=True, min_group_floats=1)}
    captured = DistributedHotPath(world, specs, betas, dev, 0, 1, seed=1, modes=modes, production_at_one_rank=True)
    captured.capture()
    runners["distributed(1 rank), production form CAPTURED in a hipGraph (kernels + collectives), one replay per step"] = captured
    single_graph = SingleGpuHotPath(world, specs, betas, dev, seed=1, layout="tiled")
    single_graph.capture()
    runners["single, captured"] = single_graph
    for name, r in runners.items():
        for _ in range(20):
            r.step()
        torch.cuda.synchronize()
        n = 2000
        t0 = time.perf_counter()
        for _ in range(n):
            r.step()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"{name}: enqueue {1e6 * t_enq / n:.1f} us/step, with final sync {1e6 * t_all / n:.1f} us/step", flush=True)
        if os.environ.get("GJ_PROFILE"):
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(500):
                r.step()
            pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
