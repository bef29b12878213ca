#!/usr/bin/env python3
"""Benchmark of the per-timestep infection hot path (rows a1-a9 of SURVEY.md section 8) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--preset c3] [--agents A]

A "step" is one pass of the hot path (transmission update -> pass 1 -> pass 2 + epilogue +
Gumbel decision + state update) over the synthetic contact world, with every input resident in
HBM when the timed region starts.  Default workload = BASELINE.json configs[2]: 10 M agents, 8
infection networks on 6 edge sets, 120 M network-edges (grad_june_amd/synthetic.py, seed 1234),
in-kernel Philox noise.  Prints ONE JSON line (rank 0).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); the SAME world is
partitioned across ranks (strong scaling) - see grad_june_amd/distributed.py.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gradabm-june_amd"))

import numpy as np
import torch

from grad_june_amd.synthetic import DEFAULT_AGENTS  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preset", default="c3", choices=["c2", "c3", "c5", "june"],
                    help="c2 / c3 / c5: BASELINE.json configs[1] / [2] / [4] (c3 = the headline workload); june: a world with "
                         "the membership structure of the reference's own graphs and its default eleven networks "
                         "(synthetic.JUNE_WORLD; numpy generator, always on a map)")
    ap.add_argument("--agents", type=int, default=None)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--geography", default="random", choices=["random", "clustered"],
                    help="random: agents dealt to venue slots uniformly at random (SURVEY 8d's specification; every "
                         "committed headline figure); clustered: the same presets on a world with a geography - households "
                         "of neighbours, venues in the own / a neighbouring super area, a stated leak (synthetic.GEOGRAPHY)")
    ap.add_argument("--infected", type=float, default=0.01)
    ap.add_argument("--layout", default="tiled", choices=["tiled", "csr"],
                    help="tiled: LDS propagation-blocked kernels (default); csr: deterministic CSR kernels")
    ap.add_argument("--sv-max", type=int, default=None)
    ap.add_argument("--eb-target", type=int, default=None)
    ap.add_argument("--slice-agents", type=int, default=None, help="tiled: agents per slice (multiple of 64)")
    ap.add_argument("--device-compile", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--host-compile", action="store_true",
                    help="compile the contact graph with numpy on the host instead of by the library's compile kernels on "
                         "the GPU (the same arrays, ~13 s instead of ~0.5 s for the default workload)")
    ap.add_argument("--tune", default="auto", choices=["auto", "on", "off"],
                    help="measure candidate tile geometries at set-up and keep the fastest "
                         "(auto: single-GPU worlds of at most 4e7 set-edges, where compiling takes seconds)")
    ap.add_argument("--reorder", default="auto", choices=["auto", "none", "household"],
                    help="graph-compile-time agent renumbering for locality (results map back through original_id); "
                         "auto = household-major: the members of a household become neighbours, which halves the halo of "
                         "a partitioned run and makes the household tiles of a single GPU diagonal (-1.5 %% per step)")
    ap.add_argument("--parts", type=int, default=0,
                    help="one GPU: step the world as this many agent partitions in turn (0 = one partition: the fastest at "
                         "every size measured, see auto_parts)")
    ap.add_argument("--quarantine", type=float, default=None,
                    help="second configuration of SURVEY 8d: an active quarantine policy with this stage threshold")
    ap.add_argument("--edge-mult", type=float, default=1.0, help="experiments: memberships per agent x this")
    ap.add_argument("--direct", default="auto", choices=["auto", "off"],
                    help="pass 2 of sets with few venues straight from an LDS table of venue values (auto) or, like the "
                         "other sets, through the per-edge workspace (off)")
    ap.add_argument("--generator", default="auto", choices=["auto", "numpy", "torch"],
                    help="numpy: the seeded generator of synthetic.make_world (what every committed headline figure uses); "
                         "torch: the same distributions drawn on the device in seconds (synthetic.iter_world_torch - another "
                         "random world; N > 1: every rank draws the same seeded stream on its own GPU and cuts its share out "
                         "there, nothing touches host memory); auto: numpy up to 4e7 agents, torch above (1e8 agents: numpy "
                         "needs most of an hour and 18 GB per rank)")
    ap.add_argument("--runs", default="auto", choices=["auto", "off"],
                    help="run form of the edge set that orders the agents (households under --reorder household): one "
                         "edge per agent read from the per-agent arrays instead of the tiled index arrays (auto) or "
                         "every edge in the tiled arrays (off)")
    ap.add_argument("--presum", default="off", choices=["on", "off"],
                    help="experiment: pass 1 of the sets in the direct form from their ELL rows and per-workgroup LDS "
                         "tables (on) instead of through phases A + B like the other sets (off, the default: faster)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"], nargs="?", const="on",
                    help="N > 1: capture the production step (kernels + RCCL collectives) in a hipGraph after the warm-up "
                         "and replay it per timed step (13.5 us of host time per step instead of ~100).  auto (default): "
                         "capture, replay ONE step against an eager step from the same state and keep the graph only if "
                         "every rank's state agrees bit for bit - otherwise the eager step is timed and the JSON says why; then "
                         "time 20 replays against 20 eager steps from one state and keep the faster form (the JSON "
                         "carries both figures).  on: the validated graph without that trial")
    ap.add_argument("--prewarm-ms", type=float, default=100.0,
                    help="one GPU: milliseconds of stateless passes at the end of the set-up (0: none), see main()")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step timed region is run this many times back to back; `value` is the FIRST region, the "
                         "others give min / median / max (one 11 ms region per run is a noisy round-over-round figure)")
    ap.add_argument("--backward", action="store_true",
                    help="row f3: time forward and backward of a T-step differentiable run (the eight log_beta as "
                         "nn.Parameter, loss = cases of the last step) instead of the forward-only hot path")
    ap.add_argument("--multi-net-block-div", type=int, default=None,
                    help="experiment: edges per venue block of the sets that carry several networks = eb_target / this")
    ap.add_argument("--wide-min-share", type=float, default=None,
                    help="experiment: share of a set's chunks spanning more than two tiles above which it gets the 8-word "
                         "descriptors (tiling.WIDE_MIN_SHARE, default 0.01); below it such chunks take a row of explicit slots")
    ap.add_argument("--tile-pad", type=int, default=1,
                    help="experiment (with --host-compile): pad every tile to a multiple of this many edges in both orders")
    ap.add_argument("--backward-recompute", action="store_true",
                    help="--backward: the memory-lean form (a step keeps its pre-state only; the backward recomputes the "
                         "forward's two sparse passes)")
    ap.add_argument("--backward-steps", type=int, default=4, help="--backward: timesteps on the autograd graph")
    ap.add_argument("--no-events", action="store_true",
                    help="diagnostic: time the K steps as plain gj_step calls, without HIP events between the launches")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: diagnostic runs with several ranks sharing one GPU (collectives staged through the host)")
    ap.add_argument("--force-distributed", action="store_true",
                    help="diagnostic: take the torch.distributed code path even with one rank")
    ap.add_argument("--world-cache", default=None,
                    help="directory in which generated worlds are kept (.npz) and reloaded from: kernel experiments "
                         "that run bench.py many times on one box skip the ~30 s of generation")
    ap.add_argument("--high-prevalence", type=float, default=0.3,
                    help="second, short timed region on the same world re-seeded at this infected fraction (0 = skip): "
                         "k_transmission skips uninfected agents' parameter lines and phase D rewrites state only where it "
                         "changes, so the headline state (1 %% infected, SURVEY 8d) is the cheap end")
    ap.add_argument("--work-order", default=None, choices=["heavy", "light", "mixed", "set", "stagger", "stagger40", "stagger64", "stagger100", "stagger128"],
                    help="experiments: order of the venue launch's (set, block) work list (default: heaviest first)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only-headline", action="store_true",
                    help="profiling runs: only the warm-up and the timed region(s) of the headline configuration (no "
                         "quarantine / high-prevalence / full-step regions, no CPU baseline), so that a profile's last "
                         "dispatches are the timed steps")
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="one GPU: after the timed regions, this many seconds of steps in blocks of 25 with the state reset "
                         "before every block - ms/step min / median / max and the last quarter's mean (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=150.0,
                    help="wall-clock bound of the CPU baseline (SURVEY 8d protocol: 3 warm-up + 10 timed steps, no_grad "
                         "and autograd on; a leg that would exceed the bound stops early and says so in `sample`)")
    return ap.parse_args()


DEFAULT_LOG_BETA = {"household": -0.4, "company": -0.3, "school": -0.3, "pub": -1.2, "gym": -1.2, "grocery": -1.2,
                    "visit": -1.2, "cinema": -1.2, "university": -0.5, "care_visit": -0.4, "care_home": -0.4}


def network_specs(world):
    from grad_june_amd import _native as N
    from grad_june_amd.defaults import default_parameters
    from grad_june_amd.infection_networks import LeisureNetwork
    from grad_june_amd.plan import NetworkSpec
    from grad_june_amd.synthetic import edge_set_of

    leisure = default_parameters()["leisure"]
    specs = []
    for name in world["networks"]:
        es = edge_set_of(name)
        if name == "household":
            specs.append(NetworkSpec(name, es, N.MASK_RAW, None))
        elif es == "leisure":
            tab = LeisureNetwork(0.0, leisure[name], device="cpu").leisure_probabilities.numpy()
            specs.append(NetworkSpec(name, es, N.MASK_QL_AGE75 if name == "care_visit" else N.MASK_QL, tab))
        else:
            specs.append(NetworkSpec(name, es, N.MASK_Q, None))
    return specs


def betas_of(world):
    return {n: float(torch.tensor(10.0) ** torch.tensor(DEFAULT_LOG_BETA[n])) for n in world["networks"]}


def kernel_bytes(world, networks):
    """Split of B_step over the three launches."""
    from grad_june_amd.synthetic import edge_set_of

    A = world["n_agents"]
    E = {k: (len(v["agent"]) if "agent" in v else v["n_edges"]) for k, v in world["edge_sets"].items()}
    V = {k: (len(v["people"]) if "people" in v else v["n_venues"]) for k, v in world["edge_sets"].items()}
    sets = {edge_set_of(n) for n in networks}
    e_sets = sum(E[s] for s in sets)
    e_nets = sum(E[edge_set_of(n)] for n in networks)
    v_nets = sum(V[edge_set_of(n)] for n in networks)
    N = len(networks)
    p1 = 4 * e_sets + 4 * e_nets + 12 * v_nets                  # pass 1
    p2 = 4 * e_sets + 4 * e_nets + 8 * N * A + 32 * A           # pass 2 + probs + sample + infect
    return {
        "transmission": 32 * A,                                   # 6 reads + 1 write + quarantine mask
        "venue_reduce": p1, "agent_gather": p2,                   # CSR layout: one launch per pass
        # tiled layout: pass 1 = scatter (A) + first half of the venue launch (B); pass 2 = second half (C)
        # + agents (D).  The venue launch is priced at half of each pass's edge terms + the venue terms.
        "tile_scatter": (4 * e_sets + 4 * e_nets) // 2,
        "tile_venues": (4 * e_sets + 4 * e_nets) // 2 + 12 * v_nets + (4 * e_sets + 4 * e_nets) // 2,
        "tile_agents": (4 * e_sets + 4 * e_nets) // 2 + 8 * N * A + 32 * A,
        # multi-GPU: the venue launch runs as its two halves around the partial-sum all-reduce
        "tile_venues_B": (4 * e_sets + 4 * e_nets) // 2 + 12 * v_nets,
        "tile_venues_C": (4 * e_sets + 4 * e_nets) // 2,
    }


def cgroup_cpu_quota():
    """CPUs the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when no quota is set / visible."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(world, networks, betas, tables, budget_s):
    """The CPU oracle (= the reference's ATen op sequence) timed on this host.  Protocol of SURVEY 8d: 3 warm-up
    + 10 timed steps of the same workload, once under no_grad and once with autograd on (the reference's default:
    the state is chained functionally, model.py:103-110, and every step stays on the graph of the log_beta
    parameters).  ``value`` is the no_grad figure - the faster of the two."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gj_oracle as O

    # the GPU box gives one GPU's job a 16-core CPU share; os.cpu_count() reports the whole host
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # every core this process may USE: the affinity mask bounded by the container's CPU quota
    # (cgroup cpu.max; the GPU box shows all 256 host cores to a job that is given 16 - with 256 threads the oracle
    # runs five times slower than with 16).  GJ_CPU_THREADS overrides.
    quota = cgroup_cpu_quota()
    share = avail if quota is None else max(1, min(avail, int(round(quota))))
    if quota is None and avail > 64 and torch.cuda.is_available():
        share = 16 * max(1, torch.cuda.device_count())         # no quota visible: the pool's documented share per GPU
    cores = max(1, min(avail, int(os.environ.get("GJ_CPU_THREADS", str(share)))))
    torch.set_num_threads(cores)
    w = {"n_agents": world["n_agents"], "age": torch.from_numpy(world["age"]), "sex": torch.from_numpy(world["sex"]),
         "edge_sets": {k: {kk: torch.from_numpy(vv) for kk, vv in v.items()} for k, v in world["edge_sets"].items()}}
    tabs = {k: torch.from_numpy(v) for k, v in tables.items()}
    t_begin = time.perf_counter()

    def leg(autograd: bool, deadline: float):
        st = {k: torch.from_numpy(v.copy()) for k, v in world["state"].items()}
        if autograd:
            # what example_scripts/run_model.py:5-8 does: every log_beta an nn.Parameter, beta = 10 ** log_beta
            logb = {n: torch.nn.Parameter(torch.tensor(DEFAULT_LOG_BETA[n])) for n in networks}
            b = {n: 10.0 ** logb[n] for n in networks}
        else:
            b = betas
        kw = dict(delta_time=1.0, day_type=0, active=networks, betas=b, leisure_tables=tabs,
                  quarantine_thresholds=None)
        times, warm = [], 0
        with torch.enable_grad() if autograd else torch.no_grad():
            for i in range(13):
                t0 = time.perf_counter()
                out = O.hot_path_step(w, st, now=1.0 + i, **kw)
                dt = time.perf_counter() - t0
                for k in ("susceptibility", "is_infected", "infection_time"):
                    st[k] = out[k]            # chained: with autograd on the graph grows step by step, as in the reference
                if i < 3:
                    warm += 1
                else:
                    times.append(dt)
                if time.perf_counter() + dt > deadline and len(times) >= 2:
                    break
        return times, warm

    half = t_begin + 0.45 * budget_s
    t_ng, w_ng = leg(False, half)
    t_ag, w_ag = leg(True, t_begin + budget_s)
    sps = 1.0 / float(np.mean(t_ng))
    sps_ag = 1.0 / float(np.mean(t_ag))
    n_edges = sum(len(world["edge_sets"][_es(n_)]["agent"]) for n_ in networks)
    return {"value": sps, "unit": "steps/s", "cores": cores, "cores_visible": avail, "cores_cgroup_quota": quota,
            "cores_host": os.cpu_count(), "kind": "port",
            "sample": f"{len(t_ng)} timed full steps of the same workload after {w_ng} warm-up steps, torch CPU ops, "
                      f"no_grad (wall-clock bound {budget_s:g} s for both legs)",
            "edges_per_s": sps * n_edges,
            "ms_per_step": {"mean": 1e3 * float(np.mean(t_ng)), "min": 1e3 * float(np.min(t_ng)),
                            "max": 1e3 * float(np.max(t_ng))},
            "autograd_on": {"value": sps_ag, "unit": "steps/s", "edges_per_s": sps_ag * n_edges,
                            "sample": f"{len(t_ag)} timed steps after {w_ag} warm-up steps, log_beta as nn.Parameter, "
                                      "state chained on the autograd graph (forward only; the reference's default mode)",
                            "ms_per_step": {"mean": 1e3 * float(np.mean(t_ag)), "min": 1e3 * float(np.min(t_ag)),
                                            "max": 1e3 * float(np.max(t_ag))}}}


def _es(n):
    from grad_june_amd.synthetic import edge_set_of

    return edge_set_of(n)


STEP_KERNEL_SOURCES = ("gj_device.h", "gj_tiled.h", "gradjune_hip.hip")


def csrc_hash() -> str:
    """sha256 over the sources of the step's kernels (what a PMC profile is valid for; the graph-compile kernels of
    gj_compile.hip and the declarations in the header do not change the step's traffic)."""
    import hashlib

    h = hashlib.sha256()
    for name in STEP_KERNEL_SOURCES:
        with open(os.path.join(ROOT, "gradabm-june_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def cached_world(args, progress, make_world):
    def gen():
        if args.generator == "torch":
            from grad_june_amd.synthetic import make_world_torch

            n = args.agents or DEFAULT_AGENTS[args.preset]
            w = make_world_torch(args.preset, n, args.seed, torch.device("cuda", torch.cuda.current_device()),
                                 infected_fraction=args.infected, geography=args.geography)
            for es in w["edge_sets"].values():      # the partitioner and the locality order work on host arrays
                es["agent"], es["venue"] = es["agent"].cpu().numpy(), es["venue"].cpu().numpy()
            torch.cuda.empty_cache()
            progress("world drawn on the device")
            return w
        return make_world(args.preset, n_agents=args.agents, seed=args.seed, infected_fraction=args.infected,
                          edge_mult=args.edge_mult, progress=progress, geography=args.geography)

    if not args.world_cache:
        return gen()
    os.makedirs(args.world_cache, exist_ok=True)
    path = os.path.join(args.world_cache,
                        f"{args.preset}_{args.agents}_{args.seed}_{args.infected}_{args.edge_mult}_{args.generator}"
                        + ("" if args.geography == "random" else "_" + args.geography) + ".npz")
    if not os.path.exists(path):
        w = gen()
        flat = {"n_agents": w["n_agents"], "age": w["age"], "sex": w["sex"], "networks": ",".join(w["networks"])}
        for k, es in w["edge_sets"].items():
            for kk, v in es.items():
                flat[f"es/{k}/{kk}"] = v
        for k, v in w["state"].items():
            flat[f"state/{k}"] = v
        np.savez(path + ".tmp.npz", **flat)
        os.replace(path + ".tmp.npz", path)
        return w
    with np.load(path) as z:
        w = {"preset": args.preset, "n_agents": int(z["n_agents"]), "age": z["age"], "sex": z["sex"],
             "networks": str(z["networks"]).split(","), "edge_sets": {}, "state": {}}
        for k in z.files:
            if k.startswith("es/"):
                _, name, kk = k.split("/")
                w["edge_sets"].setdefault(name, {})[kk] = z[k]
            elif k.startswith("state/"):
                w["state"][k[6:]] = z[k]
    order = [s for s in ("household", "care_home", "company", "school", "university", "leisure") if s in w["edge_sets"]]
    w["edge_sets"] = {s: w["edge_sets"][s] for s in order}
    progress(f"world loaded from {path}")
    return w


def backward_bench(args, world, specs, networks, dev, progress):
    """Row f3 measured (example_scripts/run_model.py:5-11: every log_beta an nn.Parameter, ``cases.backward()``):
    T chained differentiable steps (autograd.HotPathStep: the fused forward step, which keeps its per-agent and
    per-venue sums, + a hand-written backward that runs the two sparse passes transposed) and the backward through all
    of them.  ``--backward-recompute``: the memory-lean form (the backward recomputes the forward's two passes first)."""
    from types import SimpleNamespace

    from grad_june_amd import autograd as AG
    from grad_june_amd.autograd import HotPathStep
    from grad_june_amd.benchrun import SingleGpuHotPath
    from grad_june_amd.synthetic import algorithmic_bytes, network_edges

    betas = betas_of(world)
    keep = not args.backward_recompute
    AG.KEEP_FORWARD_SUMS = keep
    r = SingleGpuHotPath(world, specs, betas, dev, seed=args.seed, device_compile=not args.host_compile,
                         progress=progress)
    logb = {n: torch.nn.Parameter(torch.tensor(DEFAULT_LOG_BETA[n], device=dev)) for n in networks}
    nets = [SimpleNamespace(name=n, log_beta=logb[n]) for n in networks]
    fixed = {k: r.state[k] for k in ("max_infectiousness", "shape", "rate", "shift")}
    T = args.backward_steps
    state0 = [r.state[k].clone() for k in ("susceptibility", "is_infected", "infection_time")]

    def once():
        for v in logb.values():
            v.grad = None
        s, i, t = state0
        torch.cuda.synchronize()
        t_a = time.perf_counter()
        for k in range(T):
            params = r.engine.params(now=1.0 + k, delta_time=1.0, day_type=0, active=networks, betas=betas,
                                     seed=args.seed, step=k)
            env = {"engine": r.engine, "params": params, "fixed": fixed, "stage": None, "exp_noise": None,
                   "nets": nets, "betas": betas}
            s, i, t, _new = HotPathStep.apply(env, s, i, t, *[n.log_beta for n in nets])
        loss = i.sum()                      # cases after the last step (runner.py:167)
        torch.cuda.synchronize()
        t_b = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        return 1e3 * (t_b - t_a) / T, 1e3 * (t_c - t_b) / T, float(loss), [float(v.grad) for v in logb.values()]

    for _ in range(max(1, args.warmup // 2)):
        once()
    runs = [once() for _ in range(max(3, args.repeats))]
    fwd = float(np.median([x[0] for x in runs]))
    bwd = float(np.median([x[1] for x in runs]))
    b_step = algorithmic_bytes(world, networks)
    kb = kernel_bytes(world, networks)
    A = world["n_agents"]
    # a backward step = the two sparse passes transposed (+ the forward's two passes recomputed first in the
    # memory-lean form) + three elementwise adjoints (sampler/epilogue: ~13 arrays, transmission profile: ~9,
    # q-transmission) and the per-venue dot products of d/d log_beta (two reads of every cum)
    sparse = kb["tile_scatter"] + kb["tile_venues"] + (kb["tile_agents"] - 32 * A)
    b_bwd = (1 if keep else 2) * sparse + (0 if keep else kb["transmission"]) + (13 + 9) * 4 * A
    return {
        "metric": "differentiable simulation steps/sec (forward + backward)", "value": 1e3 / (fwd + bwd),
        "unit": "steps/s", "n_gpus": 1, "steps": T, "warmup": max(1, args.warmup // 2), "ms_per_step": fwd + bwd,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.preset}: {A} agents, {len(networks)} infection networks, "
                               f"{network_edges(world, networks)} network-edges, {T} chained differentiable steps, "
                               f"loss = cases after the last step, d/d log_beta of every network"},
        "forward_ms_per_step": fwd, "backward_ms_per_step": bwd, "backward_over_forward": bwd / fwd,
        "backward_form": ("forward sums kept by the step (4 floats per agent and step + the per-venue sums)" if keep
                          else "forward passes recomputed (3 floats per agent and step)"),
        "runs": [{"forward_ms": x[0], "backward_ms": x[1]} for x in runs],
        "algorithmic_bytes": {"forward_step": b_step, "backward_step": b_bwd},
        "roofline": {"bound": "hbm", "kernel": "backward step (all launches)", "achieved": b_bwd / (bwd * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_bwd / (bwd * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": None},
        "loss": runs[-1][2], "grad_log_beta": dict(zip(networks, runs[-1][3])),
    }


def sustained_region(runner, seconds: float, ms_per_step_hint: float, start, t_start: int, block: int = 25):
    """A sustained-clock figure: at least ``seconds`` of GPU work as blocks of ``block`` production steps (gj_step, no
    events between the launches), the epidemic state put back to the block's start state before every block (so every
    block computes the same 25 steps at the headline prevalence instead of drifting into a saturated epidemic).  Each
    block is bracketed by two HIP events on the launch stream; the reset copies (3 arrays) lie OUTSIDE the brackets,
    so ``ms_per_step`` figures are the steps' alone - the GPU still executes them, i.e. the card is busy throughout.
    ``start`` / ``t_start``: the state and step count at the start of the FIRST timed region (after the warm-up), so
    that a block is the first 25 steps of the headline region.
    Returns min / median / max over the blocks and the mean of the last quarter (clocks and temperature settled)."""
    keys = ("is_infected", "susceptibility", "infection_time")
    after = {k: runner.state[k].clone() for k in keys}
    t_after = runner.t
    t0 = t_start
    n_blocks = max(8, int(np.ceil(seconds * 1e3 / max(1e-3, ms_per_step_hint * block))))
    events = []
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    for _ in range(n_blocks):
        for k in keys:
            runner.state[k].copy_(start[k])
        runner.t = t0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(block):
            runner.step()
        b.record()
        events.append((a, b))
    torch.cuda.synchronize()
    wall = time.perf_counter() - w0
    ms = np.array([a.elapsed_time(b) / block for a, b in events])
    infected = float(runner.state["is_infected"].double().sum())
    for k in keys:
        runner.state[k].copy_(after[k])
    runner.t = t_after
    q = max(1, len(ms) // 4)
    return {"seconds_of_gpu_work": float(ms.sum() * block / 1e3), "wall_seconds": wall, "blocks": int(n_blocks),
            "steps_per_block": block, "steps": int(n_blocks * block),
            "ms_per_step": {"min": float(ms.min()), "median": float(np.median(ms)), "max": float(ms.max()),
                            "first_quarter_mean": float(ms[:q].mean()), "last_quarter_mean": float(ms[-q:].mean())},
            "steps_per_s_last_quarter": 1e3 / float(ms[-q:].mean()),
            "infected_after_a_block": infected,
            "method": "blocks of production steps (gj_step) between two HIP events on the launch stream; the state is "
                      "reset before every block, the reset copies lie outside the events"}


def auto_parts(n_agents: int) -> int:
    """One GPU: agent partitions stepped in turn (distributed.PartitionedHotPath).  Round 2 switched to partitions of
    16 M agents above 48 M, unmeasured; the round-3 sweeps (profiles/r03_c3_40m_parts*.json, r03_c5_100m_parts*.json)
    say a single partition is the fastest at every size tried - C3 at 40 M agents 2.62 / 2.78 / 2.96 / 3.00 ms for
    1 / 2 / 3 / 4 partitions in that sweep (2.38 ms with the round's final kernels, r03_c3_40m_parts1.json), C5 at 100 M
    agents 8.4 ms (r03_c5_100m_parts1.json; 12.6 ms is the same world with descriptors only, _parts1_descriptors.json)
    against 18.4 ms for 7 - so partitions are only what --parts asks for (they remain the one-GPU rehearsal of the
    multi-GPU partitioning)."""
    return 1


def self_launch(args) -> int:
    """``python bench.py --gpus N`` (N > 1) outside a torch.distributed environment: start the N ranks as CHILD
    processes (``python -m torch.distributed.run``, one rank per GPU, rendezvous on 127.0.0.1), relay rank 0's JSON
    line and return the children's exit code.  This process never touches the GPU (torch.cuda.device_count() does
    not initialise it) and never replaces itself with another program."""
    import socket
    import subprocess

    if args.backend == "nccl":
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            print(f"bench.py --gpus {args.gpus}: this node shows {n_dev} GPU(s); one rank per GPU is needed "
                  f"(diagnostics with ranks sharing a GPU: --backend gloo)", file=sys.stderr)
            return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:       # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:                      # rank 0 prints exactly one JSON line; anything else goes to stderr
        try:
            if "metric" in json.loads(out):
                line = out
                continue
        except ValueError:
            pass
        sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    elif rc == 0:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        rc = 1
    return rc


def capture_validated(runner, dev, backend, dist):
    """N > 1: capture the production step in a hipGraph and keep it only if ONE replayed step reproduces an eager
    step from the same state bit for bit on every rank.  Returns (graph in use, reason).  Whatever the outcome the
    state has advanced by exactly one step."""
    if backend != "nccl":
        runner.step()
        return False, "gloo backend: collectives are staged through the host and cannot be captured"
    keys = [k for k in ("is_infected", "susceptibility", "infection_time", "transmission", "q_transmission")
            if runner.state.get(k) is not None]
    saved = {k: runner.state[k].clone() for k in keys}
    t_saved = runner.t
    runner.step()                                   # the eager production step
    torch.cuda.synchronize()
    ref = {k: runner.state[k].clone() for k in keys}
    ref_new = runner.new_infected.clone()
    for k in keys:
        runner.state[k].copy_(saved[k])
    runner.t = t_saved
    def agreed(ok: bool) -> bool:                   # every rank takes the same path
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def fall_back(reason):
        runner.drop_graph()
        for k in keys:                              # the eager step's result, so that every rank continues from one state
            runner.state[k].copy_(ref[k])
        runner.t = t_saved + 1
        return False, reason

    reason = "replay of one step == the eager step, every rank"
    try:                                            # capturing executes nothing: no rank can be left waiting in a collective
        runner.capture()
        captured, why = True, ""
    except Exception as e:                          # capture is not supported for something in the step
        captured, why = False, f"capture failed: {type(e).__name__}: {e}"
    if not agreed(captured):                        # BEFORE anyone replays: a replay's collectives need every rank
        return fall_back(why or "capture failed on another rank")
    runner.step()                                   # first replay: the same step again, on every rank
    torch.cuda.synchronize()
    same = all(torch.equal(runner.state[k], ref[k]) for k in keys) and torch.equal(runner.new_infected, ref_new)
    if not agreed(same):
        return fall_back("the replayed step differs from the eager step" + ("" if not same else " on another rank"))
    return True, reason


def faster_of_graph_and_eager(runner, dev, dist, n_steps: int = 20):
    """``--graph auto``, after the captured step was validated: time ``n_steps`` replays and ``n_steps`` eager production
    steps from the same state (barrier + synchronize around each, MAX over ranks - every rank sees the same two
    numbers) and keep the faster form.  On one GPU a replayed rank-sized step measured 8 % SLOWER than the eager launches
    (profiles/r04_*rank_share*: the graph's clock node and its node-to-node gaps), while eager steps need the host to
    issue ~8 launches + 2-3 collectives per 0.12 ms step - which of the two wins depends on the host, so it is
    measured where it runs.  The state, the timestep and the device clock are restored: the timed region starts where
    it would have started.  Returns (graph in use, note)."""
    keys = [k for k in ("is_infected", "susceptibility", "infection_time", "transmission", "q_transmission")
            if runner.state.get(k) is not None]
    saved = {k: runner.state[k].clone() for k in keys}
    t_saved = runner.t

    def restore():
        for k in keys:
            runner.state[k].copy_(saved[k])
        runner.t = t_saved
        runner.sync_clock()

    def timed() -> float:
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            runner.step()
        torch.cuda.synchronize()
        dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return 1e3 * float(el.item()) / n_steps

    runner.step()                       # (one replay outside the clock: the first replay after a capture pays the upload)
    restore()
    g = timed()
    restore()
    token = runner.suspend_graph()
    runner.step()
    restore()
    e = timed()
    restore()
    note = f"{n_steps} steps each from one state: replay {g:.4f} ms, eager {e:.4f} ms per step"
    if g <= e:
        runner.resume_graph(token)
        return True, note + " - replaying"
    return False, note + " - eager launches"


def main():
    args = parse()
    if os.environ.get("GJ_DUMP_STACKS_AFTER"):      # diagnostics: every rank prints its Python stack after that many seconds
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["GJ_DUMP_STACKS_AFTER"]), exit=False)
    if args.preset == "june":
        args.geography = "clustered"          # (the preset is always drawn on a map; the JSON says so)
    if args.generator == "auto":
        n_default = DEFAULT_AGENTS[args.preset]
        args.generator = "torch" if (args.agents or n_default) > 40_000_000 else "numpy"
    if args.only_headline:
        args.no_cpu_baseline, args.high_prevalence = True, 0.0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    # stdout carries exactly ONE line, the JSON result: libraries that print to fd 1 (RCCL's version banner
    # at init, for one) are sent to stderr for the duration of the run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    if args.work_order:
        os.environ["GJ_WORK_ORDER"] = args.work_order
    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: start bench.py plainly (it launches its "
                         f"ranks itself) or under torch.distributed.run with --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path)")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import __graft_entry__ as entry

    if rank == 0:
        entry.build()
    distributed = world_size > 1 or args.force_distributed
    dist = None
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
        dist.barrier()

    from grad_june_amd.synthetic import make_world, algorithmic_bytes, network_edges

    t0 = time.time()

    def progress(msg):      # long set-ups keep writing: a silent command is taken for hung on the GPU pool
        if rank == 0:
            print(f"[bench {time.time() - t0:6.1f}s] {msg}", file=sys.stderr, flush=True)

    if args.multi_net_block_div:
        os.environ["GJ_MULTI_NET_BLOCK_DIV"] = str(args.multi_net_block_div)
    if args.wide_min_share is not None:
        from grad_june_amd import tiling as _TL

        _TL.WIDE_MIN_SHARE = float(args.wide_min_share)
    reorder = args.reorder if args.reorder != "auto" else "household"
    device_compile = not args.host_compile
    share = rw = None
    if world_size > 1:
        # every rank draws the world set by set from the same seeded generator and keeps only ITS share of each set
        # (no rank ever holds the whole COO, nothing is communicated)
        from grad_june_amd.distributed import stream_rank_share
        from grad_june_amd.synthetic import iter_world, iter_world_torch

        if args.generator == "torch":
            n = args.agents or DEFAULT_AGENTS[args.preset]
            pieces = iter_world_torch(args.preset, n, args.seed, dev, infected_fraction=args.infected,
                                      geography=args.geography, progress=progress)
        else:
            pieces = iter_world(args.preset, n_agents=args.agents, seed=args.seed, infected_fraction=args.infected,
                                edge_mult=args.edge_mult, progress=progress, geography=args.geography)
        rw, share = stream_rank_share(pieces, rank, world_size, reorder=None if reorder == "none" else reorder,
                                      progress=progress)
        del pieces
        torch.cuda.empty_cache()
        world = {"n_agents": share["n_agents"], "networks": share["networks"], "state": share["state"],
                 "edge_sets": {k: {"n_edges": e, "n_venues": v} for k, (e, v) in share["sizes"].items()}}
    else:
        world = cached_world(args, progress, make_world)
        if reorder != "none":
            from grad_june_amd.synthetic import reorder_agents

            world = reorder_agents(world, by=reorder)
    networks = world["networks"]
    betas = betas_of(world)
    specs = network_specs(world)
    tables = {s.name: s.table for s in specs if s.table is not None}
    t_gen = time.time() - t0

    if args.backward:
        if distributed:
            raise SystemExit("--backward is a single-GPU measurement")
        out = backward_bench(args, world, specs, networks, dev, progress)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
        return

    if distributed:
        from grad_june_amd.distributed import DistributedHotPath, choose_modes

        modes = None
        if world_size == 1:   # diagnostic: exercise the partial-sum all-reduce with a single rank
            modes = {k: ("partial" if v == "local" and k != "household" else v)
                     for k, v in choose_modes(world, 1).items()}
        runner = DistributedHotPath(world, specs, betas, dev, rank, world_size, seed=args.seed, modes=modes,
                                    progress=progress, quarantine_threshold=args.quarantine, rank_world=rw,
                                    total_edges=None if share is None else share["total_edges"],
                                    device_compile=device_compile,
                                    production_at_one_rank=world_size == 1)   # diagnostic: the overlapped multi-rank step
        extra = {"exchange": {"modes": runner.rw.modes, "venue_classes": None if share is None else share.get("classes"),
                              "halo_agents_rank0": int(runner.rw.n_halo),
                              "halo_bytes_per_step_rank0": runner.halo.bytes_per_step if runner.halo else 0,
                              "partial_sum_floats": int(runner.flat_cum.numel()) if runner.flat_cum is not None else 0}}
    else:
        from grad_june_amd.benchrun import SingleGpuHotPath

        kw = {}
        if device_compile and args.layout == "tiled":
            kw["device_compile"] = True
        if args.layout == "tiled":
            if args.direct == "off":
                kw["direct"] = False
            if args.runs == "off":
                kw["runs"] = False
            if args.presum == "on":
                kw["presum"] = True
            if args.tile_pad > 1:
                kw["tile_pad"] = args.tile_pad
            if args.sv_max:
                kw["sv_max"] = args.sv_max
            if args.eb_target:
                kw["eb_target"] = args.eb_target
            if args.slice_agents:
                sa = args.slice_agents
                kw["slices"] = (-(-world["n_agents"] // sa), sa)
        parts = args.parts if args.parts > 0 else auto_parts(world["n_agents"])
        if parts > 1 and args.layout == "tiled":
            from grad_june_amd.distributed import PartitionedHotPath

            runner = PartitionedHotPath(world, specs, betas, dev, parts, seed=args.seed, progress=progress,
                                        device_compile=device_compile)
            extra = {"partitions_on_one_gpu": parts}
        else:
            set_edges = sum(len(v["agent"]) for v in world["edge_sets"].values())
            tune = args.tune == "on" or (args.tune == "auto" and args.layout == "tiled" and not (set(kw) - {"device_compile", "direct", "runs", "presum"})
                                         and set_edges <= 40_000_000)
            if tune:      # small worlds compile in seconds: measure the candidate tile geometries, keep the best
                from grad_june_amd.benchrun import GEOMETRY_CANDIDATES, tune_geometry

                runner, seen = tune_geometry(world, specs, betas, dev, seed=args.seed, layout=args.layout,
                                             quarantine_threshold=args.quarantine, progress=progress,
                                             **kw, **({"candidates": [c for c in GEOMETRY_CANDIDATES if "direct" not in c]}
                                                      if "direct" in kw else {}))
                extra = {"geometry_tuning_ms": seen}
            else:
                runner = SingleGpuHotPath(world, specs, betas, dev, seed=args.seed, layout=args.layout,
                                          quarantine_threshold=args.quarantine, progress=progress, **kw)
                extra = {}
    t_setup = time.time() - t0

    def sync():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    # The timed region is 10 ms of GPU work after half a minute of host-side world generation: the first ~20 ms after
    # such a pause run ~2 % slower than what follows (five back-to-back regions of one cold run: 0.501, 0.502, 0.492,
    # 0.492, 0.489 ms per step - the GPU's clocks ramping).  The set-up therefore ends with >= 100 ms of STATELESS
    # passes (transmission + both sparse passes + probabilities, no decision: the geometry tuner's measurement loop;
    # the epidemic state, the timestep and the Philox counter stay where they are), then the W warm-up steps follow.
    prewarm = None
    if not distributed and hasattr(runner, "time_stateless") and args.prewarm_ms > 0:
        t0, n_pass = time.perf_counter(), 0
        while 1e3 * (time.perf_counter() - t0) < args.prewarm_ms:
            runner.time_stateless(steps=20, repeats=1)
            n_pass += 22
        prewarm = {"ms": 1e3 * (time.perf_counter() - t0), "stateless_passes": n_pass,
                   "what": "set-up: stateless passes of the hot path before the warm-up steps, so that the timed region "
                           "does not sit on the GPU's clock ramp after the host-side world generation"}
    graph_on, graph_reason = False, "single GPU: the step is one gj_step call"
    want_graph = distributed and hasattr(runner, "capture") and args.graph != "off" and args.warmup > 0
    for _ in range(args.warmup - (1 if want_graph else 0)):
        runner.step()
    sync()
    if want_graph:          # the last warm-up step is the eager step the first replay is compared with
        graph_on, graph_reason = capture_validated(runner, dev, args.backend, dist)
        if graph_on and args.graph == "auto":
            graph_on, note = faster_of_graph_and_eager(runner, dev, dist)
            graph_reason += "; " + note
        sync()
    elif distributed:
        graph_reason = "--graph off" if args.graph == "off" else "no warm-up step to validate a captured step against"
    runner.reset_timers()
    # One GPU: launches are bracketed by HIP events (on the launch stream) INSIDE the timed region - on every
    # `events_every`-th step (every 4th; from 32 steps on, 8 of them): an event between two launches costs the GPU ~4 us (measured: 19 us per step of five
    # events, 3 % of the default workload's step and 10 % of C2's), which the production call (gj_step, no events)
    # does not pay.  N > 1: the production step overlaps collectives with compute and cannot be bracketed.
    events_every = 1 if (args.steps < 12 or distributed) else max(4, args.steps // 8)     # at least 8 bracketed steps from 32 on
    if args.no_events:
        events_every = 1 << 30

    def region():
        """EXACTLY K steps between two (barrier + device synchronise) pairs; max over ranks taken later."""
        t_a = time.perf_counter()
        for i in range(args.steps):
            runner.step(timed=not distributed and i % events_every == 0)
        t_issue = time.perf_counter() - t_a           # host time to issue the K steps (nothing waited for yet)
        sync()
        return time.perf_counter() - t_a, t_issue

    start_state = start_t = None
    if not distributed and hasattr(runner, "load_state") and args.sustained_seconds > 0 and not args.only_headline:
        start_state = {k: runner.state[k].clone() for k in ("is_infected", "susceptibility", "infection_time")}
        start_t = runner.t
        torch.cuda.synchronize()
    elapsed, issue = region()
    infected_first = float(runner.state["is_infected"].double().sum())    # after warm-up + the first K steps
    repeats = [elapsed]
    for _ in range(max(0, args.repeats - 1)):         # the same region again, back to back: run-to-run spread
        repeats.append(region()[0])
    if args.no_events and not distributed:
        for _ in range(args.steps):
            runner.step(timed=True)
        sync()
    if distributed:
        # per-launch durations for N > 1: the same K steps again in the sequential, event-bracketed form
        for _ in range(args.steps):
            runner.step(timed=True)
        sync()
    kt = runner.kernel_ms()          # mean ms per launch over the timed region, HIP events on the launch stream
    import resource

    peak_rss_mb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0     # this process: set-up included
    infected_local = float(runner.state["is_infected"].double().sum())
    if distributed:
        red_dev = dev if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed, peak_rss_mb, issue] + repeats, dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, peak_rss_mb, issue = float(t[0].item()), float(t[1].item()), float(t[2].item())
        repeats = [float(x) for x in t[3:].tolist()]
        t = torch.tensor([infected_local, infected_first], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        infected_local, infected_first = float(t[0].item()), float(t[1].item())

    if rank != 0:
        dist.destroy_process_group()
        return

    def second_region(n, pre=3, **changes):
        """n steps under changed step parameters (runner attributes), HIP events around every launch."""
        kept_attr = {k: getattr(runner, k) for k in changes}
        for k, v in changes.items():
            setattr(runner, k, v)
        for _ in range(pre):
            runner.step()
        torch.cuda.synchronize()
        runner.reset_timers()
        t_a = time.perf_counter()
        for _ in range(n):
            runner.step(timed=True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t_a
        res = {"steps": n, "ms_per_step": 1e3 * el / n, "steps_per_s": n / el, "kernel_ms": runner.kernel_ms()}
        for k, v in kept_attr.items():
            setattr(runner, k, v)
        runner.reset_timers()
        return res

    high = policies = None
    single = not distributed and hasattr(runner, "load_state")
    if single:
        kept = {k: v.clone() for k, v in runner.state.items()}
        kept_t = runner.t
    if single and args.quarantine is None and not args.only_headline:
        # second configuration of SURVEY 8d: a quarantine policy (quarantine_policies.py:13-33: agents at or beyond the
        # stage threshold neither transmit nor receive outside their household) and social distancing
        # (interaction_policies.py:25-31: beta factors) both active, same world, same state
        factors = {n: (1.0 if n == "household" else 0.5) for n in networks}
        policies = second_region(max(10, args.steps // 2), q_thr=4.0,
                                 betas={n: betas[n] * factors[n] for n in networks})
        policies.update({"quarantine_stage_threshold": 4.0, "beta_factors": factors,
                         "what": "quarantine (stage >= 4 masked outside the household) + social distancing "
                                 "(beta x 0.5 on every network but household) active in every step"})
        for k, v in kept.items():
            runner.state[k].copy_(v)
        runner.t = kept_t
    if single and args.high_prevalence > 0:
        from grad_june_amd.synthetic import epidemic_state

        runner.load_state(epidemic_state(world["n_agents"], args.high_prevalence, args.seed + 1))
        high = second_region(max(10, args.steps // 2))
        high.update({"infected_fraction_at_start": args.high_prevalence,
                     "infected_fraction_at_end": float(runner.state["is_infected"].clamp(max=1).mean())})
        for k, v in kept.items():                      # back to the headline run's state for what follows
            runner.state[k].copy_(v)
        runner.t = kept_t
    if single:
        del kept

    sustained = None
    if single and start_state is not None:
        sustained = sustained_region(runner, args.sustained_seconds, 1e3 * elapsed / args.steps, start_state, start_t)

    full = None
    if not distributed and hasattr(runner, "enable_full_step") and not args.only_headline:
        # second timed region: the "full step" of SURVEY section 8d (hot path + symptoms + result reductions)
        runner.enable_full_step()
        for _ in range(3):
            runner.full_step()
        torch.cuda.synchronize()
        t_a = time.perf_counter()
        n_full = max(10, args.steps // 2)
        for _ in range(n_full):
            runner.full_step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t_a
        full = {"steps_per_s": n_full / el, "ms_per_step": 1e3 * el / n_full, "steps": n_full,
                "includes": "hot path a1-a9 + symptoms and per-step result reductions in one kernel (f1 + f2)"}

    # what the timed steps computed: variants of one kernel must agree on these (tools/ab.py prints them)
    # summed over the ranks; "after_first_region" = after warm-up + K steps: what tests/test_gpu_fullsize_properties.py
    # ::test_the_arrangement_bench_times pins for the driver's --warmup 5 --steps 20
    checksum = {"infected": infected_local, "infected_after_first_region": infected_first}
    if getattr(runner, "stamps", None) is not None:      # GJ_DIAG_STAMPS builds: cycles since workgroup start at marked points
        sa = runner.engine.plan.host.slice_agents
        st = runner.stamps[: (world["n_agents"] // sa) * sa].view(-1, sa)[:, :16].cpu().numpy()
        print("stamps (mean cycles per workgroup, slots 0-15):", np.round(st.mean(0)).tolist(), file=sys.stderr)
        print("stamps (max):", np.round(st.max(0)).tolist(), file=sys.stderr)
    sps = args.steps / elapsed
    n_edges = network_edges(world, networks)
    b_step = algorithmic_bytes(world, networks)
    kb = kernel_bytes(world, networks)
    dom = max((k for k in kt if k in kb), key=kt.get)
    share_f = 1.0 / world_size         # each rank streams its own partition
    achieved = kb[dom] * share_f / (kt[dom] * 1e-3) / 1e9
    rep_ms = [1e3 * r / args.steps for r in repeats]
    out = {
        "metric": "simulation steps/sec",
        "value": sps,
        "unit": "steps/s",
        "n_gpus": world_size,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "layout": args.layout,
        "config": {"workload": f"{args.preset}: {world['n_agents']} agents, {len(networks)} infection networks on "
                               f"{len(world['edge_sets'])} edge sets, {n_edges} network-edges, seed {args.seed}, "
                               f"{args.infected:.0%} infected, Philox noise"
                               + ("" if args.geography == "random" else f", {args.geography} geography")
                               + (f", quarantine below stage {args.quarantine:g}" if args.quarantine else ""),
                   "preset": args.preset, "n_agents": world["n_agents"], "network_edges": n_edges,
                   "parallelism": f"agents partitioned over {world_size} GPU(s)", "agent_order": reorder,
                   "geography": args.geography},
        "edges_per_s": sps * n_edges,
        "algorithmic_bytes_per_step": b_step,
        "step_roofline_frac": b_step * sps / (HBM_PEAK_GBS * 1e9 * world_size),
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": kb[dom] * share_f, "ms_per_launch": kt[dom]},
        "kernel_ms": kt,
        "kernel_ms_method": (f"HIP events around every launch of every {events_every}th step of the {len(repeats)} timed "
                             f"regions" if not distributed else
                             "HIP events around every launch and collective of K further steps run in sequence"),
        # the K-step region run `repeats` times back to back: `value` / `ms_per_step` are the FIRST one
        "repeats": {"regions": len(rep_ms), "ms_per_step": rep_ms, "min": min(rep_ms),
                    "median": float(np.median(rep_ms)), "max": max(rep_ms)},
        "host_us_per_step": 1e6 * issue / args.steps,     # host time to ISSUE a step (launches / graph replay), first region
        "state_checksum": checksum,
        "host_peak_rss_mb": peak_rss_mb,     # max over ranks: a rank streams the world and keeps only its share
        "setup_s": {"generate": t_gen, "total": t_setup},
    }
    if distributed:
        kernels_only = sum(v for k, v in kt.items() if k not in ("halo_all_to_all", "partial_all_reduce"))
        out["rccl_ranks"] = dist.get_world_size() if args.backend == "nccl" else 0
        out["backend"] = args.backend
        out["graph"] = graph_on
        out["graph_reason"] = graph_reason
        out["collective_ms_sequential"] = {k: kt[k] for k in ("halo_all_to_all", "partial_all_reduce") if k in kt}
        # what the production step (collectives overlapped with the phases that do not need them) takes beyond
        # this rank's kernels run back to back: collective time that did not hide + launch gaps
        out["exposed_collective_ms_per_step"] = max(0.0, 1e3 * elapsed / args.steps - kernels_only)
        out["kernels_ms_per_step_rank0"] = kernels_only
    out.update(extra)
    if sustained:
        out["sustained"] = sustained
    if prewarm is not None:
        out["prewarm"] = prewarm
    if full:
        out["full_step"] = full
    if high:
        out["high_prevalence"] = high
    if policies:
        out["quarantine_social_distancing"] = policies
    # HBM traffic of the dominant kernel from the committed PMC profile of this exact workload AND these exact kernel
    # sources (tools/pmc_traffic.py stamps the profile with a hash of csrc/ + the ABI header): a profile of other
    # kernels is not reported
    # Two fractions per launch and for the step, side by side: "algorithmic" prices SURVEY 8d's int32-index byte model
    # (what `roofline.achieved` must use), "measured" the bytes the chip actually moved (PMC).  The tiled launches move
    # FEWER bytes than the model (16-bit indices, the direct and run forms; the fused epilogue of k_tile_agents never
    # makes the (8 N + 64) A separate per-agent passes the model prices, which is why its algorithmic fraction can
    # exceed 1) - so the measured fraction is the one that reads as HBM utilisation.
    ms_step = 1e3 * elapsed / args.steps
    by_kernel = {k: {"ms_per_launch": kt[k], "algorithmic_bytes": kb[k] * share_f,
                     "algorithmic_frac": kb[k] * share_f / (kt[k] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "measured_bytes": None, "measured_frac": None} for k in kt if k in kb}
    out["roofline_by_kernel"] = by_kernel
    out["measured_traffic_frac"] = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        w = pmc["workload"]
        same_workload = (w["preset"], w["n_agents"], w["layout"]) == (args.preset, world["n_agents"], args.layout)
        if same_workload and world_size == 1 and pmc.get("csrc_sha256") == csrc_hash():
            out["roofline"]["traffic"] = pmc["per_launch_bytes"][dom]["total"]
            out["roofline"]["traffic_source"] = ("profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, "
                                                 "kernel sources " + pmc["csrc_sha256"][:12] + ")")
            out["roofline"]["measured_achieved"] = out["roofline"]["traffic"] / (kt[dom] * 1e-3) / 1e9
            out["roofline"]["measured_frac"] = out["roofline"]["measured_achieved"] / HBM_PEAK_GBS
            for k, e in by_kernel.items():
                if k in pmc["per_launch_bytes"]:
                    e["measured_bytes"] = pmc["per_launch_bytes"][k]["total"]
                    e["measured_frac"] = e["measured_bytes"] / (kt[k] * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["measured_traffic_bytes_per_step"] = pmc["per_step_total_bytes"]
            out["measured_traffic_frac"] = pmc["per_step_total_bytes"] / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS
        elif same_workload and world_size == 1:
            out["roofline"]["traffic_source"] = "none: profiles/pmc_traffic.json was taken with other kernel sources"
    except (OSError, KeyError):
        pass
    if not args.no_cpu_baseline and world_size == 1:
        out["cpu_baseline"] = cpu_baseline(world, networks, betas, tables, args.cpu_seconds)
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
