import sys, os
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in ("gradabm-june_amd","oracle","tests"): sys.path.insert(0, os.path.join(ROOT,p))
sys.path.insert(0, ROOT)
import torch
import grad_june_amd as G
from test_gradients import load_case, step_info, _model_and_timer, _hetero
dev=torch.device("cuda:0")
for case in ("g1","g2"):
    sub, world, tables, names = load_case(case)
    model, timer = _model_and_timer(G, case, dev)
    data = _hetero(G, sub, world, dev)
    for n in names:
        net = model.infection_networks.networks[n]
        net.log_beta = torch.nn.Parameter(net.log_beta.detach().clone())
    series=[]
    for i in range(int(sub["n_steps"])):
        s = step_info(sub, i)
        next(timer)
        data["agent"].symptoms["current_stage"] = s["stage"].to(dev)
        model.hot_path(data, timer, exp_noise=s["noise"])
        series.append(data["agent"].is_infected.sum())
    params=[model.infection_networks.networks[n].log_beta for n in names]
    for tag, loss in (("last", series[-1]), ("series", torch.stack(series).sum())):
        grads = torch.autograd.grad(loss, params, retain_graph=True, allow_unused=True)
        for n,g in zip(names, grads):
            got = 0.0 if g is None else float(g); ref=float(sub[f"grad_{tag}/{n}"])
            print(case, tag, n, "got %.6g ref %.6g rel %.2e" % (got, ref, abs(got-ref)/max(abs(ref),1e-12)))
