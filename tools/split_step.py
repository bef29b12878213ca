#!/usr/bin/env python3
"""Experiment (GPU): the C3 step with phases A + B/C run for two GROUPS of edge sets one after the other, so that the
per-edge workspace a group writes in phase A (~170 MB instead of ~330 MB) is still in the 256 MB Infinity Cache when
its venue launch reads it.

    python tools/split_step.py [--steps 30] [--groups "school,university,pub,gym,grocery|company,care_home,household"]

Sequence per step: transmission (all) -> for each group: phase 8 (A, then B + C) -> phase 3 (D + epilogue, all).
Compared with the fused `gj_step` on the same runner and state; both end in the same state (asserted).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gradabm-june_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--groups", default="school,university,pub,gym,grocery|company,care_home,household")
    ap.add_argument("--world-cache", default="/tmp/gj_worlds")
    a = ap.parse_args()
    import torch

    import bench as B
    from grad_june_amd.benchrun import SingleGpuHotPath
    from grad_june_amd.synthetic import make_world, reorder_agents

    class Args:
        preset, agents, seed, infected, edge_mult, world_cache, generator, geography = "c3", None, 1234, 0.01, 1.0, a.world_cache, "numpy", "random"

    world = reorder_agents(B.cached_world(Args, lambda m: print(m, file=sys.stderr), make_world), by="household")
    dev = torch.device("cuda:0")
    r = SingleGpuHotPath(world, B.network_specs(world), B.betas_of(world), dev, seed=1234, layout="tiled", device_compile=True)
    groups = [g.split(",") for g in a.groups.split("|")]
    assert sorted(n for g in groups for n in g) == sorted(r.networks), (groups, r.networks)

    def params(active):
        return r.engine.params(now=1.0 + r.t, delta_time=1.0, day_type=0, active=active, betas=r.betas,
                               has_quarantine=False, q_threshold=float("inf"), seed=r.seed, step=r.t)

    def split_step():
        e = r.engine
        p_all = params(r.networks)
        e.step_phase(r.bufs, p_all, r.io, 0)
        for g in groups:
            e.step_phase(r.bufs, params([n for n in r.networks if n in g]), r.io, 8)
        e.step_phase(r.bufs, p_all, r.io, 3)
        r.t += 1

    streams = [torch.cuda.Stream(device=dev) for _ in groups[1:]]

    def forked_step():
        """The groups side by side: group 0 on the current stream, the others on streams of their own (fork after the
        transmission launch, join in front of phase D) - one group's tail under another's ramp."""
        e = r.engine
        main = torch.cuda.current_stream(dev)
        p_all = params(r.networks)
        e.step_phase(r.bufs, p_all, r.io, 0)
        ps = [params([n for n in r.networks if n in g]) for g in groups]
        for st, p in zip(streams, ps[1:]):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                e.step_phase(r.bufs, p, r.io, 8)
        e.step_phase(r.bufs, ps[0], r.io, 8)
        for st in streams:
            main.wait_stream(st)
        e.step_phase(r.bufs, p_all, r.io, 3)
        r.t += 1

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for _ in range(5):
        r.step()
    snap = {k: v.clone() for k, v in r.state.items()}
    t_snap = r.t
    out = {}
    for name, fn in (("fused gj_step", r.step), ("groups in sequence", split_step), ("groups side by side", forked_step),
                     ("fused gj_step again", r.step), ("groups in sequence again", split_step),
                     ("groups side by side again", forked_step)):
        for k, v in snap.items():
            r.state[k].copy_(v)
        r.t = t_snap
        ms = timed(fn, a.steps)
        out[name] = (ms, float(r.state["is_infected"].sum()))
        print(f"{name:22s} {ms * 1e3:7.1f} us/step   infected after {a.steps} steps: {out[name][1]:.0f}", flush=True)
    assert len({v[1] for v in out.values()}) == 1, "the split sequence must end in the same state"


if __name__ == "__main__":
    main()
