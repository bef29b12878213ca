import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradabm-june_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import api_overhead as AO
import grad_june_amd as G
from grad_june_amd.defaults import default_parameters
from grad_june_amd.synthetic import make_world
dev = "cuda:0"
world = make_world("c3", n_agents=500_000)
d = AO.hetero_from(world, dev)
pp = default_parameters(dev)
pp["timer"]["step_activities"]["weekday"][0] = list(world["networks"]); pp["timer"]["step_activities"]["weekend"][0] = list(world["networks"])
pp["networks"] = {k: v for k, v in pp["networks"].items() if k in world["networks"]}; pp["timer"]["total_days"] = 400
model = G.GradJune.from_parameters(pp); timer = G.Timer.from_parameters(pp)
for net in model.infection_networks.networks.values():
    net.log_beta = torch.nn.Parameter(net.log_beta)
next(timer); model(d, timer); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    next(timer); model(d, timer)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
