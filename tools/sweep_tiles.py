"""Experiment driver (GPU): time the tiled kernels of one world under several tile geometries.

    python tools/sweep_tiles.py [preset] [agents]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradabm-june_amd"))
sys.path.insert(0, ROOT)

import torch

import bench as B
import __graft_entry__ as entry
from grad_june_amd.benchrun import SingleGpuHotPath
from grad_june_amd.synthetic import make_world


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "c3"
    agents = int(sys.argv[2]) if len(sys.argv) > 2 else None
    entry.build()
    world = make_world(preset, n_agents=agents)
    specs, betas = B.network_specs(world), B.betas_of(world)
    dev = torch.device("cuda:0")
    A = world["n_agents"]

    def run(tag, layout="tiled", **kw):
        r = SingleGpuHotPath(world, specs, betas, dev, seed=1, layout=layout, **kw)
        for _ in range(3):
            r.step()
        torch.cuda.synchronize()
        r.reset_timers()
        t0 = time.perf_counter()
        for _ in range(20):
            r.step(timed=True)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 20 * 1e3
        print(tag, "ms/step %.3f" % el, {k: round(v, 3) for k, v in r.kernel_ms().items()}, flush=True)
        del r
        torch.cuda.empty_cache()

    def sl(sa):
        return (-(-A // sa), sa)

    if os.environ.get("GJ_SWEEP", "geometry") == "grid":
        # GJ_EB="16384,32768" GJ_SA="4928,9856": every combination of edges-per-block and agents-per-slice
        run("default")
        for sa in [int(x) for x in os.environ.get("GJ_SA", "19584").split(",")]:
            for eb in [int(x) for x in os.environ.get("GJ_EB", "131072").split(",")]:
                for sv in [int(x) for x in os.environ.get("GJ_SV", "16384").split(",")]:
                    for wide in [{"auto": None, "0": False, "1": True}[x] for x in os.environ.get("GJ_WIDE", "auto").split(",")]:
                        run(f"SA={sa} EB={eb} SV={sv} wide={wide}", slices=sl(sa), eb_target=eb, sv_max=sv, desc_wide=wide)
        return
    if os.environ.get("GJ_SWEEP", "geometry") == "perset":
        # does running A, B, C set by set keep each set's val workspace in the 256 MiB Infinity Cache?
        from grad_june_amd.synthetic import edge_set_of

        r = SingleGpuHotPath(world, specs, betas, dev, seed=1, layout="tiled")
        e = r.engine
        groups = {}
        for n in r.networks:
            groups.setdefault(edge_set_of(n), []).append(n)

        def p_of(nets):
            return e.params(now=1.0 + r.t, delta_time=1.0, day_type=0, active=nets, betas=r.betas, seed=r.seed, step=r.t)

        def step_all():
            p = p_of(r.networks)
            for ph in (0, 1, 2, 3):
                e.step_phase(r.bufs, p, r.io, ph)
            r.t += 1

        def step_perset(k=1):
            p = p_of(r.networks)
            e.step_phase(r.bufs, p, r.io, 0)
            names = list(groups)
            for i in range(0, len(names), k):
                nets = [n for g in names[i:i + k] for n in groups[g]]
                e.step_phase(r.bufs, p_of(nets), r.io, 8)
            e.step_phase(r.bufs, p, r.io, 3)
            r.t += 1

        for tag, fn in (("all sets per phase", step_all), ("set by set", lambda: step_perset(1)),
                        ("two sets at a time", lambda: step_perset(2)), ("three sets at a time", lambda: step_perset(3)),
                        ("all sets per phase", step_all)):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                fn()
            torch.cuda.synchronize()
            print(tag, "ms/step %.3f" % ((time.perf_counter() - t0) / 40 * 1e3), flush=True)
        return
    if os.environ.get("GJ_SWEEP", "geometry") == "geometry":
        run("default")
        run("split epilogue", split_epilogue=True)
        run("default again")
        run("split epilogue again", split_epilogue=True)
        return
    # phase anatomy: diagnostic phases of gj_step_phase
    r = SingleGpuHotPath(world, specs, betas, dev, seed=1, layout="tiled", slices=sl(19584), sv_max=16384, eb_target=65536)
    for _ in range(3):
        r.step()
    from grad_june_amd.engine import HipTimer
    t = HipTimer()
    for label, phases in (("A scatter", [1]), ("B only", [5]), ("C only", [6]), ("B+C fused", [2]),
                          ("D no sampling", [4]), ("D full", [3])):
        p = r.params()
        for ph in phases:
            r.engine.step_phase(r.bufs, p, r.io, ph)
        torch.cuda.synchronize()
        t.start()
        for _ in range(10):
            for ph in phases:
                r.engine.step_phase(r.bufs, p, r.io, ph)
        t.stop()
        print(label, "%.3f ms" % (t.elapsed_ms() / 10), flush=True)


if __name__ == "__main__":
    main()
