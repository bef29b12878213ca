"""Diagnose a tile geometry (GPU): per-launch times next to what the compiled plan looks like.

    python tools/geometry_cliff.py c2 1000000 eb_target=32768 sv_max=16384 slice_agents=2048 [reorder=household]

Written for the 64x cliff of profiles/r02_c2_1m_bench.json (`geometry_tuning_ms`: that candidate 7.03 ms, best 0.108):
prints, per edge set, blocks / tiles / edges per tile, the descriptor format and how many 64-edge chunks take each
slot path (fast two-segment, wide six-segment, table walk), then times the launches with HIP events.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradabm-june_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import bench as B
import __graft_entry__ as entry
from grad_june_amd.benchrun import SingleGpuHotPath
from grad_june_amd.synthetic import make_world, reorder_agents


def describe(plan):
    rows = []
    for hs in plan.sets:
        t = hs.tiled
        host = lambda a: a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
        sptr = host(t.tile_sptr).astype(np.int64)
        lens = np.diff(sptr)
        desc = host(t.chunk_desc).reshape(-1, 8 if t.desc_wide else 4)
        if t.desc_wide:
            walk = ((desc[:, 7].view(np.uint32) >> 8) & 1).astype(bool)
        else:
            walk = (desc[:, 2].view(np.uint32) >> 16).astype(bool)
        nz = lens[lens > 0]
        rows.append({"set": hs.name, "edges": t.n_edges, "blocks": t.n_blocks, "slices": t.n_slices,
                     "tiles_nonempty": int(len(nz)), "edges_per_tile_mean": float(nz.mean()) if len(nz) else 0.0,
                     "edges_per_tile_p10": float(np.percentile(nz, 10)) if len(nz) else 0.0,
                     "desc_wide": bool(t.desc_wide), "chunks": int(len(desc)), "chunks_walk": int(walk.sum()),
                     "walk_share": float(walk.mean()) if len(desc) else 0.0, "direct": bool(t.ell_k),
                     "explicit_slots": t.slot_idx is not None, "run_form": t.runs is not None})
    return rows


def main():
    preset, agents = sys.argv[1], int(sys.argv[2])
    kw = dict(a.split("=") for a in sys.argv[3:])
    reorder = kw.pop("reorder", "household")
    kw = {k: int(v) for k, v in kw.items()}
    for flag in ("desc_explicit", "direct", "runs"):      # booleans: 0 / 1
        if flag in kw:
            kw[flag] = bool(kw[flag])
            if flag != "desc_explicit" and kw[flag]:
                del kw[flag]                               # (True = the compile's own choice)
    entry.build()
    world = make_world(preset, n_agents=agents)
    if reorder != "none":
        world = reorder_agents(world, by=reorder)
    specs, betas = B.network_specs(world), B.betas_of(world)
    dev = torch.device("cuda:0")
    sa = kw.pop("slice_agents", None)
    if sa:
        kw["slices"] = (-(-world["n_agents"] // sa), sa)
    r = SingleGpuHotPath(world, specs, betas, dev, seed=1, device_compile=True, **kw)
    for row in describe(r.engine.plan.host):
        print(row, flush=True)
    for _ in range(2):
        r.step()
    torch.cuda.synchronize()
    r.reset_timers()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        r.step(timed=True)
    torch.cuda.synchronize()
    print("ms/step %.3f" % ((time.perf_counter() - t0) / n * 1e3), {k: round(v, 4) for k, v in r.kernel_ms().items()},
          flush=True)


if __name__ == "__main__":
    main()
