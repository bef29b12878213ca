#!/usr/bin/env python3
"""Per-workgroup timeline of the venue launch (phases B + C) from a GJ_DIAG_STAMPS build:

    python tools/ab.py --build-only stamps:GJ_DIAG_STAMPS=1
    GJ_LIB_PATH=.../variants/libgj_stamps.so python tools/venue_timeline.py [--work-order heavy]

Runs a few steps of the default workload and prints, per edge set, the workgroups' durations, and for the launch the
makespan, the slot-time (sum of durations) and how many workgroups were resident over time."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gradabm-june_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work-order", default=None)
    ap.add_argument("--world-cache", default="/tmp/gj_worlds")
    a = ap.parse_args()
    if a.work_order:
        os.environ["GJ_WORK_ORDER"] = a.work_order
    import torch

    import bench as B
    from grad_june_amd import _native as N
    from grad_june_amd.benchrun import SingleGpuHotPath
    from grad_june_amd.synthetic import make_world, reorder_agents

    class Args:
        preset, agents, seed, infected, edge_mult, world_cache, generator = "c3", None, 1234, 0.01, 1.0, a.world_cache, "numpy"
        geography = "random"

    world = reorder_agents(B.cached_world(Args, lambda m: print(m, file=sys.stderr), make_world), by="household")
    dev = torch.device("cuda:0")
    r = SingleGpuHotPath(world, B.network_specs(world), B.betas_of(world), dev, seed=1234, layout="tiled", device_compile=True)
    for _ in range(5):
        r.step()
    torch.cuda.synchronize()
    lib = N.load()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    r.engine.step_phase(r.bufs, r.params(), r.io, 2)            # one more venue launch, timed: calibrates the tick
    ev1.record()
    torch.cuda.synchronize()
    launch_us = 1e3 * ev0.elapsed_time(ev1)
    buf = (C.c_ulonglong * (3 * 8192))()
    lib.gj_diag_venue_stamps.restype = C.c_int
    assert lib.gj_diag_venue_stamps(buf) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 3).astype(np.int64)
    work = r.engine.plan.host.work
    n = len(work)                      # one workgroup per work item
    st = st[:n]
    xcd = st[:, 2]
    start, end = st[:, 0].astype(np.float64), st[:, 1].astype(np.float64)
    t0 = start.min()
    start, end = (start - t0) / 100.0, (end - t0) / 100.0        # s_memrealtime: 100 MHz for the whole chip -> microseconds
    dur = end - start
    print(f"launch {launch_us:.1f} us by events; workgroups per XCD: {np.bincount(xcd).tolist()}; "
          f"last end per XCD {[round(float(end[xcd == x].max()), 1) for x in np.unique(xcd)]}")
    print(f"{n} workgroups, makespan {end.max():.1f} us, slot-time {dur.sum() / 1e3:.2f} ms "
          f"= {dur.sum() / end.max():.0f} resident workgroups on average")
    names = [s.name for s in r.engine.plan.host.sets]
    for sid, name in enumerate(names):
        m = work[:, 0] == sid
        if m.any():
            print(f"  {name:12s} {m.sum():5d} wgs  duration mean {dur[m].mean():6.1f} min {dur[m].min():6.1f} max {dur[m].max():6.1f} us"
                  f"   start mean {start[m].mean():6.1f} max {start[m].max():6.1f}   end max {end[m].max():6.1f}")
    for lo in range(0, n, max(1, n // 12)):
        m = slice(lo, min(n, lo + max(1, n // 12)))
        print(f"  work items {lo:5d}+: start {start[m].mean():6.1f} us, duration {dur[m].mean():5.1f} us")
    grid = np.linspace(0, end.max(), 21)
    res = [(int(((start <= t) & (end > t)).sum())) for t in grid]
    print("resident workgroups at 5 % steps of the makespan:", res)


if __name__ == "__main__":
    main()
