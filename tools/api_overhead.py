"""GPU: per-step cost of the reference-shaped API (GradJune.forward on a HeteroData world) against the
bare engine loop, for a small (769 agents) and a large synthetic world."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradabm-june_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import grad_june_amd as G
from grad_june_amd.defaults import default_parameters
from grad_june_amd.synthetic import make_world


def hetero_from(world, device):
    d = G.HeteroData()
    A = world["n_agents"]
    ag = d["agent"]
    ag.id = torch.arange(A)
    ag.age = torch.from_numpy(world["age"])
    ag.sex = torch.from_numpy(world["sex"])
    for s, es in world["edge_sets"].items():
        d[s].id = torch.arange(len(es["people"]))
        d[s].people = torch.from_numpy(es["people"])
        d["agent", "attends_" + s, s].edge_index = torch.from_numpy(np.vstack((es["agent"], es["venue"])))
    d = d.to(device)
    st = world["state"]
    ag.infection_parameters = {k: torch.from_numpy(st[k]).to(device) for k in ("max_infectiousness", "shape", "rate", "shift")}
    for k in ("is_infected", "susceptibility", "infection_time"):
        ag[k] = torch.from_numpy(st[k]).to(device)
    ag.transmission = torch.zeros(A, device=device)
    ag.symptoms = {"current_stage": torch.from_numpy(st["current_stage"]).to(device),
                   "next_stage": torch.from_numpy(st["current_stage"]).to(device) + 1,
                   "time_to_next_stage": torch.zeros(A, device=device)}
    return d


def main():
    dev = "cuda:0"
    p = default_parameters(dev)
    p["timer"]["total_days"] = 60
    torch.manual_seed(0)
    runner = G.Runner.from_parameters(p)
    with torch.no_grad():
        runner()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner()
        torch.cuda.synchronize()
    print("Runner, 769 agents, 11 networks: %.3f ms per timestep (60 steps)" % ((time.perf_counter() - t0) / 60 * 1e3))

    for n in ([int(a) for a in sys.argv[1:] if a.isdigit()] or [2_000_000]):
        world = make_world("c3", n_agents=n)
        d = hetero_from(world, dev)
        pp = default_parameters(dev)
        pp["timer"]["step_activities"]["weekday"][0] = list(world["networks"])
        pp["timer"]["step_activities"]["weekend"][0] = list(world["networks"])
        pp["networks"] = {k: v for k, v in pp["networks"].items() if k in world["networks"]}
        pp["timer"]["total_days"] = 400
        model = G.GradJune.from_parameters(pp)
        timer = G.Timer.from_parameters(pp)
        with torch.no_grad():
            for _ in range(3):
                next(timer)
                model(d, timer)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                next(timer)
                model(d, timer)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 50
            t0 = time.perf_counter()
            for _ in range(50):
                next(timer)
                model.hot_path(d, timer)
            torch.cuda.synchronize()
            dh = (time.perf_counter() - t0) / 50
        print("GradJune.forward, %d agents: %.3f ms per step (hot path alone %.3f ms)" % (n, dt * 1e3, dh * 1e3))


if __name__ == "__main__":
    main()


def backward_cost(n=2_000_000, steps=8):
    """Row f3: cost of a differentiable run (forward in grad mode + backward) per timestep."""
    dev = "cuda:0"
    world = make_world("c3", n_agents=n)
    d = hetero_from(world, dev)
    pp = default_parameters(dev)
    pp["timer"]["step_activities"]["weekday"][0] = list(world["networks"])
    pp["timer"]["step_activities"]["weekend"][0] = list(world["networks"])
    pp["networks"] = {k: v for k, v in pp["networks"].items() if k in world["networks"]}
    pp["timer"]["total_days"] = 400
    model = G.GradJune.from_parameters(pp)
    timer = G.Timer.from_parameters(pp)
    for net in model.infection_networks.networks.values():
        net.log_beta = torch.nn.Parameter(net.log_beta)
    warm = []
    for _ in range(3):            # warm-up: plan compile, first launches of the adjoint kernels
        next(timer)
        model(d, timer)
        warm.append(d["agent"].is_infected.sum())
    torch.stack(warm).sum().backward()
    for k in ("susceptibility", "is_infected", "infection_time"):
        d["agent"][k] = d["agent"][k].detach()
    for net in model.infection_networks.networks.values():
        net.log_beta.grad = None
    # epochs of `steps` timesteps, as a calibration loop runs them: the first one also pays for the allocator's first
    # blocks (a T-step graph keeps every step's tensors alive until its backward), the later ones reuse them
    for epoch in range(3):
        for k in ("susceptibility", "is_infected", "infection_time"):
            d["agent"][k] = d["agent"][k].detach()
        for k in ("current_stage", "next_stage", "time_to_next_stage"):
            d["agent"].symptoms[k] = d["agent"].symptoms[k].detach()
        for net in model.infection_networks.networks.values():
            net.log_beta.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        series = []
        for _ in range(steps):
            next(timer)
            model(d, timer)
            series.append(d["agent"].is_infected.sum())
        torch.cuda.synchronize()
        tf = (time.perf_counter() - t0) / steps
        loss = torch.stack(series).sum()
        t0 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t0) / steps
        del series, loss
        grads = {k: float(v.log_beta.grad) for k, v in model.infection_networks.networks.items()}
        print("differentiable run, %d agents, epoch %d: forward %.3f ms/step, backward %.3f ms/step; d cases/d log_beta = %s"
              % (n, epoch, tf * 1e3, tb * 1e3, {k: round(v, 1) for k, v in grads.items()}), flush=True)


if __name__ == "__main__" and "--backward" in sys.argv:
    backward_cost()
