#!/usr/bin/env python3
"""What ONE rank of an N-GPU run does, measured on one GPU: the rank's share of the benchmark world is built exactly
as `bench.py --gpus N` builds it (streamed set by set, household-major order, exchange classes venue by venue -
`distributed.classify_venues`; `--exchange-rule set` = the per-set rule of rounds 1-3, `distributed.mode_of`),
compiled, and stepped WITHOUT the collectives (halo slots and remote partial sums stay zero: the kernels' work does
not depend on the values).  Prints one JSON line: the rank's kernel time per step, its halo and partial-sum volumes
(what the two collectives would move) and the host memory the set-up needed.

    python tools/rank_share.py --preset c5 --agents 100000000 --of 8 [--rank 0] [--steps 20]
    python tools/rank_share.py --geography clustered --of 8          # a world with a geography (synthetic.GEOGRAPHY)
"""
from __future__ import annotations

import argparse
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gradabm-june_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="c3")
    ap.add_argument("--agents", type=int, default=None)
    ap.add_argument("--of", type=int, default=8, help="ranks of the run")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--eb-target", type=int, default=None)
    ap.add_argument("--sv-max", type=int, default=None)
    ap.add_argument("--slice-agents", type=int, default=None, help="agents per slice (default: sized for the owned agents)")
    ap.add_argument("--geography", default="random", choices=["random", "clustered"])
    ap.add_argument("--graph", type=int, default=1, help="also replay the step from a hipGraph (0: eager launches only)")
    ap.add_argument("--generator", default="numpy", choices=["numpy", "torch"],
                    help="torch: the world is drawn on the device and the share cut out there (synthetic.iter_world_torch)")
    ap.add_argument("--exchange-rule", default="venue", choices=["venue", "set"])
    ap.add_argument("--halo-order", default="venue", choices=["venue", "id"],
                    help="order of a peer's halo agents in the extended index range: by the venue that needs them, or by id")
    a = ap.parse_args()
    import torch

    import __graft_entry__ as entry
    import bench as B
    from grad_june_amd import distributed as D
    from grad_june_amd.distributed import DistributedHotPath, stream_rank_share
    from grad_june_amd.synthetic import DEFAULT_AGENTS, iter_world, iter_world_torch

    D.EXCHANGE_RULE = a.exchange_rule
    D.HALO_ORDER = a.halo_order
    entry.build()
    dev = torch.device("cuda:0")
    t0 = time.time()

    def progress(msg):
        print(f"[rank_share {time.time() - t0:6.1f}s] {msg}", file=sys.stderr, flush=True)

    if a.generator == "torch":
        n = a.agents or DEFAULT_AGENTS[a.preset]
        pieces = iter_world_torch(a.preset, n, a.seed, dev, infected_fraction=0.01, geography=a.geography, progress=progress)
    else:
        pieces = iter_world(a.preset, n_agents=a.agents, seed=a.seed, infected_fraction=0.01, progress=progress,
                            geography=a.geography)
    rw, share = stream_rank_share(pieces, a.rank, a.of, reorder="household", progress=progress, slice_agents=a.slice_agents)
    del pieces
    torch.cuda.empty_cache()
    t_part = time.time() - t0
    world = {"n_agents": share["n_agents"], "networks": share["networks"], "state": share["state"],
             "edge_sets": {k: {"n_edges": e, "n_venues": v} for k, (e, v) in share["sizes"].items()}}
    specs, betas = B.network_specs(world), B.betas_of(world)
    hp = DistributedHotPath(world, specs, betas, dev, a.rank, a.of, seed=a.seed, collectives=False, progress=progress,
                            rank_world=rw, total_edges=share["total_edges"], device_compile=True,
                            plan_kw={k: v for k, v in (("eb_target", a.eb_target), ("sv_max", a.sv_max)) if v})
    t_setup = time.time() - t0
    stamps = None
    if os.environ.get("GJ_DIAG_STAMPS"):       # a library built with -DGJ_DIAG_STAMPS (tools/ab.py --build-only stamps:GJ_DIAG_STAMPS=1):
        stamps = torch.zeros(rw.n_local, dtype=torch.float32, device=dev)      # phase D writes cycle stamps per workgroup
        hp.io = hp.engine.io(new_infected=hp.new_infected, trans_susc=stamps)
    inf0 = float(hp.state["is_infected"].clamp(max=1).mean())
    for _ in range(a.warmup):
        hp.step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(a.steps):
        hp.step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t1) / a.steps
    hp.reset_timers()
    for _ in range(max(4, a.steps // 4)):
        hp.step(timed=True)
    torch.cuda.synchronize()
    graph_ms = None
    if a.graph:                                 # the same launches replayed from a hipGraph (DistributedHotPath.capture):
        hp.capture()                            # what the production loop does; takes the host's launch cost out of the figure
        for _ in range(a.warmup):
            hp.step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.steps):
            hp.step()
        e1.record()
        torch.cuda.synchronize()
        graph_ms = e0.elapsed_time(e1) / a.steps
    out = {
        "what": f"rank {a.rank} of {a.of}: kernels of one step on its share, collectives not executed",
        "geometry": {"slice_agents": int(rw.slice_agents), "n_slices": int(rw.n_slices), "eb_target": a.eb_target, "sv_max": a.sv_max},
        "preset": a.preset, "geography": a.geography, "generator": a.generator, "exchange_rule": a.exchange_rule, "halo_order": a.halo_order, "n_agents_world": share["n_agents"], "n_owned": int(rw.n_local), "n_halo": int(rw.n_halo),
        "modes": rw.modes, "venue_classes": share["classes"],
        "local_set_edges": {k: int(len(v["agent"])) for k, v in rw.edge_sets.items()},
        "halo_all_to_all_bytes_in_per_step": 4 * int(rw.n_halo),
        "partial_sum_all_reduce_bytes_per_step": 4 * int(hp.flat_cum.numel()) if hp.flat_cum is not None else 0,
        "kernel_ms_per_step": ms, "graph_replay_ms_per_step": graph_ms, "kernel_ms": hp.kernel_ms(),
        "infected_fraction": {"start": inf0, "end": float(hp.state["is_infected"].clamp(max=1).mean())},
        "setup_s": {"stream_and_partition": t_part, "total": t_setup},
        "host_peak_rss_mb": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0,
        "hbm_peak_gb": {"allocated": torch.cuda.max_memory_allocated() / 1e9, "reserved": torch.cuda.max_memory_reserved() / 1e9},
    }
    if stamps is not None:
        import numpy as np

        sa = int(rw.slice_agents)
        st = stamps[: (rw.n_local // sa) * sa].view(-1, sa)[:, :16].cpu().numpy()
        out["phase_d_stamps_mean_cycles"] = [float(x) for x in np.round(st.mean(0))]
        out["phase_d_stamps_legend"] = ("slot 1 after the tiled gather, 2 sums in registers, 8+2t / 9+2t direct item t begins / "
                                        "ends, 3 after the direct sets, 4 ts in LDS, 6 after the epilogue loop, 5 end")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
