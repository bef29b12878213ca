"""CPU ORACLE for the per-timestep infection message-passing path of GradABM-JUNE.

*** TEST INFRASTRUCTURE - NOT PRODUCT CODE. ***
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the timed CPU baseline.  The product path
(``gradabm-june_amd/``) never imports it and fails loudly when the HIP library is missing.

What it is: a restatement, in the same eager ATen ops the reference issues (``index_select``,
``mul``, ``scatter_add_``, elementwise chain, ``exponential_``/softmax for the Gumbel sampler),
of rows a1-a9 of SURVEY.md section 8.  Every function cites the reference lines it follows
(paths relative to /root/reference/).  The sparse gather/scatter follows the published
algorithm of ``torch_geometric.nn.conv.MessagePassing.propagate`` (PyG >= 2.3, pin in
requirements.txt:5; PyG is absent from this image) for ``aggr="add", node_dim=-1,
flow="source_to_target"`` as used at grad_june/infection_networks/base.py:11-13,78-87.

Parity pinning ("Oracle"):
  1. the reference's own known-answer tests, restated in tests/test_oracle_kat.py
     (test/unit/infection_networks/test_base.py:39-44 exact; test_leisure_network.py:61-77;
     test_interaction_policies.py:92-123; test_quarantine_policies.py:40-72;
     test_close_venue_policies.py:46-69);
  2. golden vectors in tests/golden/*.npz produced by running the reference's own .py files
     in the build container (tests/golden/make_golden.py), which this oracle must reproduce.

All functions are dtype-generic: call with float64 tensors to get the fp64 check values used
for giant venues.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

# fixed order in which active networks are accumulated: grad_june/timer.py:14-26,139-157
ACTIVITY_HIERARCHY = [
    "school",
    "university",
    "company",
    "care_home",
    "pub",
    "gym",
    "grocery",
    "visit",
    "care_visit",
    "cinema",
    "household",
]

GUMBEL_TAU = 0.1  # grad_june/infection.py:14-16


# --------------------------------------------------------------------------------------
# a1  TransmissionUpdater.forward            grad_june/transmission.py:39-51
# --------------------------------------------------------------------------------------
def transmission_update(max_infectiousness, shape, rate, shift, infection_time, is_infected, now):
    t = now - infection_time
    sign = (torch.sign(t - shift + 1e-10) + 1) / 2
    aux = torch.exp(-torch.lgamma(shape)) * torch.pow((t - shift) * rate, shape - 1.0)
    aux2 = torch.exp((shift - t) * rate) * rate
    return max_infectiousness * sign * aux * aux2 * is_infected


# --------------------------------------------------------------------------------------
# a2  QuarantinePolicies.apply               grad_june/policies/quarantine_policies.py:13-18,26-33
# --------------------------------------------------------------------------------------
def quarantine_mask(current_stage, thresholds: Sequence[Optional[float]]):
    """``thresholds``: one entry per quarantine policy; ``None`` for a policy that is not
    active at this date (contributes ones)."""
    mask = torch.ones(current_stage.shape, device=current_stage.device)
    for thr in thresholds:
        if thr is None:
            ret = torch.ones(current_stage.shape, device=current_stage.device)
        else:
            ret = (current_stage < thr).to(torch.float)
        mask = mask * ret
    return mask


# --------------------------------------------------------------------------------------
# a3  p_contact                              grad_june/infection_networks/base.py:63-70
# --------------------------------------------------------------------------------------
def p_contact(people):
    one = torch.tensor(1.0, device=people.device)
    zero = torch.tensor(0.0, device=people.device)
    return torch.maximum(torch.minimum(1.0 / (people - 1), one), zero)


# --------------------------------------------------------------------------------------
# a4  leisure tables and masks               grad_june/infection_networks/leisure_network.py:26-42,61-85,107-120
# --------------------------------------------------------------------------------------
def leisure_agent_probabilities(table, sex, age, day_type: int):
    """``table`` [2,2,100] = [day_type, sex, age] (leisure_network.py:26-42)."""
    return table[day_type, sex, age]


# --------------------------------------------------------------------------------------
# a5/a6  MessagePassing(aggr="add").propagate with message x_j*y_i   base.py:78-87
# --------------------------------------------------------------------------------------
def propagate(src_index, dst_index, x, y):
    """out[d] = sum_{e: dst(e)=d} x[src(e)] * y[d], out has the size of y."""
    x_j = x.index_select(-1, src_index)
    y_i = y.index_select(-1, dst_index)
    msg = x_j * y_i
    out = msg.new_zeros(y.shape)
    return out.scatter_add_(-1, dst_index, msg)


def infection_network(
    *,
    kind: str,
    beta,
    people,
    agent_index,
    venue_index,
    transmission,
    susceptibility,
    qmask=1.0,
    leisure_prob=None,
    age=None,
    return_cum: bool = False,
):
    """InfectionNetwork.forward (base.py:61-84) for one network.

    kind: "household" (raw values, base.py:144-149) | "plain" (quarantine mask, base.py:47-59)
          | "leisure" (mask*leisure_prob, leisure_network.py:61-85)
          | "care_visit" (leisure + susceptibility*(age>75), leisure_network.py:107-120)
    beta: scalar (python float or 0-d tensor) = 10**log_beta * policy factors (base.py:36-42)
    """
    n_venues = people.shape[0]
    beta_v = beta * torch.ones(n_venues, device=transmission.device)
    beta_v = beta_v * p_contact(people)
    if kind == "household":
        trans, susc = transmission, susceptibility
    elif kind == "plain":
        trans, susc = qmask * transmission, qmask * susceptibility
    elif kind == "leisure":
        trans = qmask * leisure_prob * transmission
        susc = qmask * leisure_prob * susceptibility
    elif kind == "care_visit":
        trans = qmask * leisure_prob * transmission
        susc = qmask * leisure_prob * susceptibility * (age > 75)
    else:
        raise ValueError(kind)
    if beta_v.dtype != trans.dtype:
        beta_v = beta_v.to(trans.dtype)
    cum = propagate(agent_index, venue_index, trans, beta_v)          # pass 1, size V
    ts = propagate(venue_index, agent_index, cum, susc)               # pass 2, size A
    if return_cum:
        return ts, cum
    return ts


# --------------------------------------------------------------------------------------
# a7  InfectionNetworks.forward epilogue     grad_june/infection_networks/base.py:118-141
# --------------------------------------------------------------------------------------
def not_infected_probabilities(ts_per_network: List[torch.Tensor], n_agents: int, delta_time: float,
                               dtype=torch.float32):
    trans_susc = torch.zeros(n_agents, dtype=dtype)
    for ts in ts_per_network:           # caller passes them in activity-hierarchy order
        trans_susc += ts
    trans_susc = torch.clamp(trans_susc, min=1e-6, max=100)
    p = torch.exp(-trans_susc * delta_time)
    return torch.clamp(p, min=0.0, max=1.0)


# --------------------------------------------------------------------------------------
# a8  IsInfectedSampler.forward              grad_june/infection.py:13-18
#     = torch.nn.functional.gumbel_softmax(logits, tau=0.1, hard=True, dim=0) with the
#       Exponential(1) draws supplied by the caller (so that a GPU run can inject them).
# --------------------------------------------------------------------------------------
def sample_infected(not_infected_probs, exp_noise):
    """``exp_noise``: [2, A] Exponential(1) draws; row 0 pairs with "not infected"."""
    logits = torch.vstack((not_infected_probs, 1.0 - not_infected_probs)).log()
    gumbels = -exp_noise.log()
    gumbels = (logits + gumbels) / GUMBEL_TAU
    y_soft = gumbels.softmax(0)
    index = y_soft.max(0, keepdim=True)[1]
    y_hard = torch.zeros_like(logits).scatter_(0, index, 1.0)
    ret = y_hard - y_soft.detach() + y_soft      # straight-through: forward hard, gradient of y_soft
    return 1.0 - ret[0, :]


def sample_infected_torch(not_infected_probs):
    """The reference's exact call (draws from torch's global CPU generator)."""
    logits = torch.vstack((not_infected_probs, 1.0 - not_infected_probs)).log()
    infection = torch.nn.functional.gumbel_softmax(logits, dim=0, tau=GUMBEL_TAU, hard=True)
    return 1.0 - infection[0, :]


def draw_exp_noise(n_agents: int, generator: Optional[torch.Generator] = None, dtype=torch.float32):
    """The draw gumbel_softmax makes: one exponential_() on a [2, A] tensor, row-major."""
    return torch.empty(2, n_agents, dtype=dtype).exponential_(generator=generator)


# --------------------------------------------------------------------------------------
# a9  GradJune.infect_people                 grad_june/model.py:103-110
# --------------------------------------------------------------------------------------
def infect_people(susceptibility, is_infected, infection_time, new_infected, now):
    zero = torch.tensor(0.0, dtype=susceptibility.dtype)
    susceptibility = torch.maximum(zero, susceptibility - new_infected)
    is_infected = is_infected + new_infected
    infection_time = infection_time + new_infected * (now - infection_time)
    return susceptibility, is_infected, infection_time


# --------------------------------------------------------------------------------------
# f1 ("next" row)  SymptomsUpdater.forward + SymptomsSampler.sample_next_stage
#     grad_june/symptoms.py:204-247, 82-128 - with the randomness supplied by the caller:
#     progresses[a]  = the torch.bernoulli(probs) outcome (symptoms.py:96)
#     dwell[a]       = the stage-time sample the agent would use (T_i or R_i .rsample, :113-126)
# --------------------------------------------------------------------------------------
def symptoms_update(age, current_stage, next_stage, time_to_next_stage, new_infected, time,
                    n_stages: int, progresses, dwell):
    next_stage = next_stage + new_infected * (2.0 - next_stage)
    time_to_next_stage = time_to_next_stage + new_infected * (time - time_to_next_stage)
    mask1 = time >= time_to_next_stage
    mask2 = current_stage < n_stages - 1
    mask_transition = mask1 * mask2
    current_stage = current_stage - (current_stage - next_stage) * mask_transition
    mask_symp_stage = progresses.to(torch.bool)
    mask_recovered_stage = ~mask_symp_stage
    for i in range(2, n_stages - 1):
        mask_stage = current_stage == i
        mask_stage = mask_stage * current_stage / i
        mask_updating = mask_stage * mask_transition
        mask_symp = mask_updating * mask_symp_stage
        next_stage = next_stage + mask_symp
        time_to_next_stage = time_to_next_stage + dwell * mask_symp
        mask_rec = mask_updating * mask_recovered_stage
        next_stage = next_stage - next_stage * mask_rec
        time_to_next_stage = time_to_next_stage + dwell * mask_rec
    return current_stage, next_stage, time_to_next_stage


def adjoint_symptoms(current_stage, next_stage, time_to_next_stage, new_infected, time, n_stages: int,
                     progresses, g_cur, g_nxt, g_ttn=None, dwell=None):
    """Hand-written adjoint of symptoms_update (what gj_adjoint_symptoms computes): given the gradients
    w.r.t. the OUTPUT current / next stage / time, those w.r.t. the INPUT current / next stage / time and
    new_infected.  The gradient paths of symptoms.py:82-128, 231-236 are next += new_infected * (2 - next),
    time += new_infected * (now - time), the transition current -= (current - next) * mask and the value-1
    factor ``(current == i) * current / i`` that multiplies the +1 (onward) / -next (recover) updates of
    next_stage and the dwell time added to time_to_next_stage; times never feed a stage (comparisons only)."""
    f64 = torch.float64
    c0, x0, t0, nw = (v.to(f64) for v in (current_stage, next_stage, time_to_next_stage, new_infected))
    g_cur, g_nxt = g_cur.to(f64), g_nxt.to(f64)
    g_ttn = torch.zeros_like(g_cur) if g_ttn is None else g_ttn.to(f64)
    d = torch.zeros_like(g_cur) if dwell is None else dwell.to(f64)
    x1 = x0 + nw * (2.0 - x0)
    t1 = t0 + nw * (time - t0)
    moving = (time >= t1) & (c0 < n_stages - 1)
    m = moving.to(f64)
    c1 = c0 - (c0 - x1) * m
    s = c1.long().clamp(0, n_stages - 1)
    at = moving & (s >= 2) & (s <= n_stages - 2) & (c1 == s)
    onward = progresses.to(torch.bool)
    sf = s.to(f64).clamp(min=1.0)
    zero = torch.zeros_like(g_cur)
    gc1 = (g_cur + torch.where(at, g_ttn * d / sf, zero) + torch.where(at & onward, g_nxt / sf, zero)
           - torch.where(at & ~onward, g_nxt * x1 / sf, zero))
    gx1 = torch.where(at & ~onward, zero, g_nxt)
    g_cur_in = gc1 * (1.0 - m)
    gx1 = gx1 + gc1 * m
    return g_cur_in, gx1 * (1.0 - nw), g_ttn * (1.0 - nw), gx1 * (2.0 - x0) + g_ttn * (time - t0)


def symptoms_progress_probability(table, age, current_stage, next_stage, time_to_next_stage, new_infected, time,
                                  n_stages: int):
    """probs handed to torch.bernoulli: table[stage after the due transition, age] (symptoms.py:93-95)."""
    next_stage = next_stage + new_infected * (2.0 - next_stage)
    ttn = time_to_next_stage + new_infected * (time - time_to_next_stage)
    mt = (time >= ttn) * (current_stage < n_stages - 1)
    cur = current_stage - (current_stage - next_stage) * mt
    return table[cur.long(), age]


# --------------------------------------------------------------------------------------
# whole hot-path step on a neutral "world" description (rows a1-a9 in model.py:125-138 order)
# --------------------------------------------------------------------------------------
def network_kind(name: str) -> str:
    if name == "household":
        return "household"
    if name == "care_visit":
        return "care_visit"
    if name in ("pub", "gym", "grocery", "visit", "cinema"):
        return "leisure"
    return "plain"


def edge_set_of(name: str) -> str:
    """Leisure-class networks all read ``attends_leisure`` (leisure_network.py:44-48)."""
    return "leisure" if network_kind(name) in ("leisure", "care_visit") else name


def hot_path_step(
    world: Dict,
    state: Dict[str, torch.Tensor],
    *,
    now: float,
    delta_time: float,
    day_type: int,
    active: Sequence[str],
    betas: Dict[str, float],
    leisure_tables: Optional[Dict[str, torch.Tensor]] = None,
    quarantine_thresholds: Optional[Sequence[Optional[float]]] = None,
    exp_noise: Optional[torch.Tensor] = None,
    dtype=torch.float32,
    return_intermediates: bool = False,
):
    """One pass of rows a1-a9.

    world: {"n_agents", "age" i64[A], "sex" i64[A],
            "edge_sets": {set: {"agent" i64[E], "venue" i64[E], "people" [V]}}}
    state: max_infectiousness, shape, rate, shift, infection_time, is_infected, susceptibility,
           current_stage   (all [A])
    active: network names for this step, ANY order; they are accumulated in hierarchy order
            (timer.py:139-157).  quarantine_thresholds: None = no quarantine-policy collection
            at all (mask is the python scalar 1.0, base.py:48-51).
    Returns the new state dict (+ transmission, not_infected_probs, new_infected).
    """
    A = world["n_agents"]
    f = lambda t: t.to(dtype)
    transmission = transmission_update(
        f(state["max_infectiousness"]), f(state["shape"]), f(state["rate"]), f(state["shift"]),
        f(state["infection_time"]), f(state["is_infected"]), now,
    )
    if quarantine_thresholds is None:
        qmask = 1.0
    else:
        qmask = quarantine_mask(state["current_stage"], quarantine_thresholds).to(dtype)
    susceptibility = f(state["susceptibility"])
    order = sorted(active, key=ACTIVITY_HIERARCHY.index)
    ts_list, inter = [], {}
    for name in order:
        es = world["edge_sets"][edge_set_of(name)]
        kind = network_kind(name)
        lp = None
        if kind in ("leisure", "care_visit"):
            lp = leisure_agent_probabilities(leisure_tables[name], world["sex"], world["age"], day_type).to(dtype)
        ts, cum = infection_network(
            kind=kind, beta=betas[name], people=es["people"], agent_index=es["agent"],
            venue_index=es["venue"], transmission=transmission, susceptibility=susceptibility,
            qmask=qmask, leisure_prob=lp, age=world["age"], return_cum=True,
        )
        ts_list.append(ts)
        if return_intermediates:
            inter["cum_" + name] = cum
            inter["ts_" + name] = ts
    p = not_infected_probabilities(ts_list, A, delta_time, dtype=dtype)
    if exp_noise is None:
        new_inf = sample_infected_torch(p)
    else:
        new_inf = sample_infected(p, exp_noise.to(dtype))
    susc2, inf2, time2 = infect_people(susceptibility, f(state["is_infected"]), f(state["infection_time"]), new_inf, now)
    out = dict(state)
    out.update(
        transmission=transmission, not_infected_probs=p, new_infected=new_inf,
        susceptibility=susc2, is_infected=inf2, infection_time=time2,
    )
    if return_intermediates:
        out.update(inter)
        if not isinstance(qmask, float):
            out["qmask"] = qmask
    return out


# --------------------------------------------------------------------------------------
# f3  hand-written adjoint of one hot-path step (what the HIP backward implements), restated with
#     dense torch ops and NO autograd - checked against autograd of hot_path_step and against the
#     reference's recorded gradients (tests/test_gradients.py).
#     Inputs: the step's pre-state and scalars, the noise, and the gradients flowing into its outputs
#     (g_susc, g_inf, g_time w.r.t. the post-step susceptibility / is_infected / infection_time).
#     Returns (grad_susc_in, grad_inf_in, grad_time_in, {network: d loss / d log_beta}).
# --------------------------------------------------------------------------------------
def adjoint_step(world, state, *, now, delta_time, day_type, active, betas, leisure_tables=None,
                 quarantine_thresholds=None, exp_noise, g_susc, g_inf, g_time, g_new=None):
    import math

    A = world["n_agents"]
    f64 = torch.float64
    mx, shp, rt, sh = (state[k].to(f64) for k in ("max_infectiousness", "shape", "rate", "shift"))
    time0, inf0, susc0 = state["infection_time"].to(f64), state["is_infected"].to(f64), state["susceptibility"].to(f64)
    g_susc, g_inf, g_time = g_susc.to(f64), g_inf.to(f64), g_time.to(f64)
    # ---- recompute the forward pieces -----------------------------------------------------------
    t = now - time0
    d = t - sh
    sign = (torch.sign(d + 1e-10) + 1) / 2
    aux = torch.exp(-torch.lgamma(shp)) * torch.pow(d * rt, shp - 1.0)
    aux2 = torch.exp((sh - t) * rt) * rt
    base = mx * sign * aux * aux2                      # d trans / d is_infected
    trans = base * inf0
    dtrans_dt = trans * ((shp - 1.0) / d - rt)         # d trans / d t ;  d t / d infection_time = -1
    q = 1.0 if quarantine_thresholds is None else quarantine_mask(state["current_stage"], quarantine_thresholds).to(f64)
    order = sorted(active, key=ACTIVITY_HIERARCHY.index)
    acc = torch.zeros(A, dtype=f64)                    # sum_n w_n * (L_n (m_n trans)), before * susc
    per_net = {}
    for name in order:
        es = world["edge_sets"][edge_set_of(name)]
        kind = network_kind(name)
        m = w = torch.ones(A, dtype=f64)
        if kind != "household":
            m = w = m * q
        if kind in ("leisure", "care_visit"):
            lp = leisure_agent_probabilities(leisure_tables[name], world["sex"], world["age"], day_type).to(f64)
            m = m * lp
            w = w * lp
        if kind == "care_visit":
            w = w * (world["age"] > 75)
        y = float(betas[name]) * p_contact(es["people"]).to(f64)
        S = torch.zeros(len(y), dtype=f64).scatter_add_(0, es["venue"], (m * trans)[es["agent"]])
        Ln = torch.zeros(A, dtype=f64).scatter_add_(0, es["agent"], (y * S)[es["venue"]])
        acc += w * Ln
        per_net[name] = (es, m, w, y, S)
    ts = susc0 * acc
    inside = (ts >= 1e-6) & (ts <= 100)
    tsc = torch.clamp(ts, 1e-6, 100)
    p = torch.exp(-tsc * delta_time)
    l0, l1 = torch.log(p), torch.log(1 - p)
    g = -torch.log(exp_noise.to(f64))
    z = torch.stack((l0 + g[0], l1 + g[1])) / GUMBEL_TAU
    ysoft = torch.softmax(z, 0)
    nu = (ysoft[1] > ysoft[0]).to(f64)                 # forward value of new_infected
    # ---- adjoints ---------------------------------------------------------------------------------
    x = susc0 - nu
    h = torch.where(x > 0, torch.ones_like(x), torch.where(x == 0, torch.full_like(x, 0.5), torch.zeros_like(x)))
    nu_bar = g_inf + g_time * (now - time0) - g_susc * h
    if g_new is not None:                              # new_infected is also read by the symptoms updater
        nu_bar = nu_bar + g_new.to(f64)
    dnu_dp = -(ysoft[0] * ysoft[1] / GUMBEL_TAU) * (1.0 / p + 1.0 / (1.0 - p))
    dnu_dp = torch.nan_to_num(dnu_dp, nan=0.0, posinf=0.0, neginf=0.0)
    ts_bar = nu_bar * dnu_dp * (-delta_time * p) * inside
    grad_susc = g_susc * h + ts_bar * acc
    xp = susc0 * ts_bar                                # input of the transposed pipeline
    trans_bar = torch.zeros(A, dtype=f64)
    grad_lb = {}
    for name, (es, m, w, y, S) in per_net.items():
        Sp = torch.zeros(len(y), dtype=f64).scatter_add_(0, es["venue"], (w * xp)[es["agent"]])
        trans_bar += m * torch.zeros(A, dtype=f64).scatter_add_(0, es["agent"], (y * Sp)[es["venue"]])
        grad_lb[name] = float((y * S * Sp).sum()) * math.log(10.0)
    grad_inf = g_inf + trans_bar * base
    grad_time = g_time * (1.0 - nu) - trans_bar * dtrans_dt
    new_state = infect_people(state["susceptibility"], state["is_infected"], state["infection_time"], nu.float(), now)
    return grad_susc, grad_inf, grad_time, grad_lb, new_state
