"""Differentiable hot-path step (row f3 of SURVEY section 8): gradients w.r.t. every network's
``log_beta`` and w.r.t. the incoming state, so that a loss on the infection counts can be
back-propagated through the timesteps like with the reference (example_scripts/run_model.py:9-11).

The forward of a step is the ordinary fused HIP step on fresh output tensors.  The backward is
hand-written (``oracle/gj_oracle.py:adjoint_step`` is its dense CPU restatement, checked against the
reference's autograd):

* the two sparse passes are self-adjoint up to exchanging the per-network masks, so the gradient
  w.r.t. the transmissions is the SAME four tiled phases run with ``transpose = 1`` on the vector
  ``susceptibility * ts_bar``;
* d loss / d log_beta_n = ln(10) * sum_v cum_n[v] * cum'_n[v] / (beta_n * p_contact[v]) - a dot
  product of the forward and transposed per-venue sums;
* the rest (straight-through Gumbel-softmax, clamp/exp/clamp, infect_people, transmission profile)
  is elementwise: ``gj_adjoint_sample`` and ``gj_adjoint_transmission``.

What a step keeps for its backward (``KEEP_FORWARD_SUMS``): the pre-state (3 per-agent floats) and - by default -
the two products of the forward's sparse passes that the adjoint needs, the per-agent sums before the susceptibility
factor (``gj_step_io.agent_sums``, one float per agent) and the per-venue sums ``cum`` (a clone, a few MB): 4 floats
per agent and step, 160 MB per step of a 10 M-agent world out of 288 GB.  With ``GJ_BACKWARD_RECOMPUTE=1`` (or
``autograd.KEEP_FORWARD_SUMS = False``) a step keeps the pre-state only and its backward recomputes the two passes
first (round 2's form: 3 floats per agent and step; C3 at 10 M agents: a backward of 1.42 ms = 2.5x the forward instead
of 0.91 ms = 1.6x, profiles/r04_c3_10m_backward*.json).  Both give the same gradients bit for bit - the kept sums ARE
what the recomputation produces (tests/test_gradients.py::test_kept_forward_sums_equal_the_recomputation).
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

import os

from . import _native as N
from .engine import AgentBuffers
from .plan import SPLIT_SUFFIX

KEEP_FORWARD_SUMS = os.environ.get("GJ_BACKWARD_RECOMPUTE", "0") in ("", "0")


def _keep_sums(env) -> bool:
    return bool(env.get("keep_sums", KEEP_FORWARD_SUMS))


def _ones(plan, n):
    """A constant vector of ones on the plan's device (the adjoint runs the passes with susceptibility = 1; the phases
    it calls only read it) - one per plan, not a fill launch per backward step."""
    t = getattr(plan, "_adjoint_ones", None)
    if t is None or t.numel() != n:
        t = torch.ones(n, dtype=torch.float32, device=plan.device)
        plan._adjoint_ones = t
    return t


def _clone_forward_cum(plan, nets):
    """{edge set: clone of its per-venue sums} for every set a network of ``nets`` (or its twin on a split set) runs on."""
    cum_fwd = {}
    for _, names in _names_with_twins(plan, nets):
        for name in names:
            es = plan.networks[name].edge_set
            if es not in cum_fwd:
                cum_fwd[es] = plan.cum_of(es).clone()
    return cum_fwd


def _forward_sums(engine, p, bufs, acc, nets, compute_transmission: bool):
    """Forward of the two sparse passes with susceptibility = 1 in ``bufs``: fills ``acc`` with
    sum_n w_n * (L_n (m_n trans)) and returns {edge set: clone of its per-venue sums}."""
    plan = engine.plan
    io = engine.io(trans_susc=acc)
    p.transpose = 0
    if compute_transmission:
        engine.step_phase(bufs, p, io, 0)                      # transmission (+ q * transmission)
    else:
        engine.quarantine_transmission(bufs, p)                # the caller supplied the transmissions
    engine.step_phase(bufs, p, io, 8)                          # phase A, then B + C in one launch (cum stays in place)
    cum_fwd = _clone_forward_cum(plan, nets)
    engine.step_phase(bufs, p, io, 4)
    return cum_fwd


def _transposed_passes(engine, p, bufs, scratch, x, nets, betas, cum_fwd):
    """The transposed pipeline on x = susceptibility * ts_bar: returns (d loss / d transmission,
    [d loss / d log_beta per network of ``nets``]).  ``scratch`` is the transmission buffer of ``bufs``."""
    plan = engine.plan
    n = plan.host.n_agents
    # The tiled passes sum in fixed point (2^-36 / 2^-32 resolution, |value| <= 16384 / 262144): scales chosen for the
    # forward's transmissions.  A cotangent has whatever magnitude the user's loss gives it (an MSE on case counts:
    # 1e5; a normalised loss: 1e-10), so x is brought to max |x| in [0.5, 1) by a power of two first and the results
    # are scaled back - exact, the passes being linear - without a host synchronisation.
    lo, hi = torch.aminmax(x)                                  # (one reduction launch; max |x| = max(-lo, hi))
    scale = _power_of_two_scale(torch.maximum(-lo, hi))
    torch.div(x, scale, out=scratch[:n])
    tbar = torch.empty(n, dtype=torch.float32, device=plan.device)
    io_t = engine.io(trans_susc=tbar)
    p.transpose = 1
    try:
        engine.quarantine_transmission(bufs, p)                # q * x for the masked sets
        engine.step_phase(bufs, p, io_t, 8)                    # phase A, then B + C in one launch: cum' is complete
        grads = [g.to(torch.float32) for g in _beta_gradients(plan, nets, betas, cum_fwd, scale)]
        engine.step_phase(bufs, p, io_t, 4)                    # tbar = d loss / d transmission (of x / scale)
    finally:
        p.transpose = 0
    return tbar.mul_(scale), grads


def _names_with_twins(plan, nets):
    """(network, [its name and - on a set the multi-GPU partition split - its twin's]) for every network."""
    return [(net, [nm for nm in (net.name, net.name + SPLIT_SUFFIX) if nm in plan.networks]) for net in nets]


def _beta_gradients(plan, nets, betas, cum_fwd, scale, weights_of=None) -> List[torch.Tensor]:
    """ln(10) * sum_v cum_n[v] * cum'_n[v] / (beta_n * p_contact[v]) per network, from the forward's and the
    transposed pass's per-venue sums (``plan.cum_of`` holds the latter).  ``weights_of(edge set)``: this rank's
    weight of every venue (multi-GPU: the ranks' values are summed by the caller).  One launch per edge set + one to
    finish (gj_adjoint_beta_*: fp64, summed in a fixed order) - as torch ops this was ~10 small launches per network
    and what a backward step spent most of its host time on."""
    lib, dev = N.load(), plan.device
    partial = torch.zeros(N.GJ_ADJ_BETA_BLOCKS * N.GJ_MAX_NETS, dtype=torch.float64, device=dev)
    out = torch.zeros(max(1, len(nets)), dtype=torch.float64, device=dev)
    per_set, per_set_k = {}, {}
    for col, (net, names) in enumerate(_names_with_twins(plan, nets)):
        for name in names:
            es = plan.networks[name].edge_set
            k = per_set_k.get(es, 0)
            per_set_k[es] = k + 1
            per_set.setdefault(es, []).append((k, col, float(betas[net.name])))
    scale32 = scale.to(device=dev, dtype=torch.float32).reshape(1)
    keep = []
    for es, items in per_set.items():
        i = plan.host.set_index[es]
        nk = len(items)
        assert [k for k, _, _ in items] == list(range(nk))
        beta = (C.c_float * nk)(*[b for _, _, b in items])
        cols = (C.c_int32 * nk)(*[c for _, c, _ in items])
        w = weights_of(es) if weights_of is not None else None
        if w is not None:
            w = w.to(device=dev, dtype=torch.float64).contiguous()
        fwd, bwd = cum_fwd[es].contiguous(), plan.cum_of(es)
        keep.append((w, fwd))
        N.check(lib.gj_adjoint_beta_partial(plan.host.sets[i].n_venues, int(plan.c.sets[i].cum_stride), nk, N.ptr(fwd),
                                            N.ptr(bwd), N.ptr(plan.keep[i]["v_pc"]), N.ptr(w), beta, cols, N.ptr(partial),
                                            N.current_stream()), "gj_adjoint_beta_partial")
    N.check(lib.gj_adjoint_beta_finish(len(nets), N.ptr(partial), N.ptr(scale32), N.ptr(out), N.current_stream()),
            "gj_adjoint_beta_finish")
    return [out[i] for i in range(len(nets))]


def _power_of_two_scale(peak: torch.Tensor) -> torch.Tensor:
    scale = torch.where((peak > 0) & torch.isfinite(peak), torch.exp2(torch.ceil(torch.log2(peak.clamp_min(1e-45)))),
                        torch.ones_like(peak))
    return torch.where(torch.isfinite(scale) & (scale > 0), scale, torch.ones_like(scale))


def _param_grads(nets, grads):
    out = []
    for net, g in zip(nets, grads):
        lb = net.log_beta
        out.append(g.to(lb.device).reshape(lb.shape) if isinstance(lb, torch.Tensor) and lb.requires_grad else None)
    return out


class HotPathStep(torch.autograd.Function):
    """(susceptibility, is_infected, infection_time, *log_betas) -> (susceptibility', is_infected',
    infection_time', new_infected)."""

    @staticmethod
    def forward(ctx, env, susc, inf, time, *log_betas):
        engine, params, fixed, stage, exp_noise, nets = (env[k] for k in ("engine", "params", "fixed", "stage",
                                                                           "exp_noise", "nets"))
        plan = engine.plan
        n = plan.host.n_agents
        out_s, out_i, out_t = (t.detach().to(torch.float32).clone().contiguous() for t in (susc, inf, time))
        n_ext = plan.host.n_ext_agents                # (one GPU: no halo slots, the step writes every transmission)
        trans = (torch.empty if n_ext == n else torch.zeros)(n_ext, dtype=torch.float32, device=plan.device)
        new_inf = torch.empty(n, dtype=torch.float32, device=plan.device)
        bufs = AgentBuffers(plan, **fixed, infection_time=out_t, is_infected=out_i, susceptibility=out_s,
                            transmission=trans, current_stage=stage)
        keep = _keep_sums(env) and plan.c.tiled is not None and bool(plan.c.tiled)
        acc = torch.empty(n, dtype=torch.float32, device=plan.device) if keep else None
        engine.step(bufs, params, engine.io(new_infected=new_inf, exp_noise=exp_noise, agent_sums=acc))
        ctx.env = env
        ctx.save_for_backward(susc.detach().to(torch.float32).contiguous(), inf.detach().to(torch.float32).contiguous(),
                              time.detach().to(torch.float32).contiguous())
        ctx.transmission = trans
        ctx.kept = (acc, _clone_forward_cum(plan, nets)) if keep else None
        return out_s, out_i, out_t, new_inf

    @staticmethod
    def backward(ctx, g_susc, g_inf, g_time, g_new):
        env = ctx.env
        engine, params, fixed, stage, exp_noise, nets = (env[k] for k in ("engine", "params", "fixed", "stage",
                                                                           "exp_noise", "nets"))
        susc0, inf0, time0 = ctx.saved_tensors
        plan, lib, dev = engine.plan, N.load(), engine.plan.device
        n = plan.host.n_agents
        if plan.c.tiled is None or not bool(plan.c.tiled):
            raise NotImplementedError("the backward pass runs on the tiled layout")

        def f32(g):
            return None if g is None else g.detach().to(torch.float32).contiguous()

        g_susc, g_inf, g_time, g_new = f32(g_susc), f32(g_inf), f32(g_time), f32(g_new)
        ones = _ones(plan, n)
        n_ext = plan.host.n_ext_agents                # (no halo slots on one GPU: the transposed pass overwrites all of it)
        scratch = (torch.empty if n_ext == n else torch.zeros)(n_ext, dtype=torch.float32, device=dev)
        bufs = AgentBuffers(plan, **fixed, infection_time=time0, is_infected=inf0, susceptibility=ones,
                            transmission=scratch, current_stage=stage)
        p = params
        if ctx.kept is not None:      # the forward's per-agent and per-venue sums, kept by the step
            acc, cum_fwd = ctx.kept
        else:                         # recompute the forward of the two passes from the saved pre-state
            acc = torch.empty(n, dtype=torch.float32, device=dev)
            cum_fwd = _forward_sums(engine, p, bufs, acc, nets, compute_transmission=True)
        # ---- elementwise adjoint of epilogue + sampler + infect_people ------------------------------------
        x = torch.empty(n, dtype=torch.float32, device=dev)
        grad_susc = torch.empty(n, dtype=torch.float32, device=dev)
        grad_time = torch.empty(n, dtype=torch.float32, device=dev)
        N.check(lib.gj_adjoint_sample(n, N.ptr(susc0), N.ptr(time0), N.ptr(acc), N.ptr(exp_noise), int(p.seed),
                                      int(p.step), int(p.agent_offset), float(p.now), float(p.delta_time),
                                      N.ptr(g_susc), N.ptr(g_inf), N.ptr(g_time), N.ptr(g_new), N.ptr(x),
                                      N.ptr(grad_susc), N.ptr(grad_time), N.current_stream()), "gj_adjoint_sample")
        # ---- transposed passes on x = susc0 * ts_bar --------------------------------------------------------
        tbar, grads = _transposed_passes(engine, p, bufs, scratch, x, nets, env["betas"], cum_fwd)
        # ---- through the transmission profile ------------------------------------------------------------------
        grad_inf = torch.empty(n, dtype=torch.float32, device=dev)
        st0 = AgentBuffers(plan, **fixed, infection_time=time0, is_infected=inf0, susceptibility=ones,
                           transmission=scratch)
        N.check(lib.gj_adjoint_transmission(n, C.byref(st0.c), float(p.now), N.ptr(tbar), N.ptr(g_inf),
                                            N.ptr(grad_inf), N.ptr(grad_time), N.current_stream()),
                "gj_adjoint_transmission")
        return (None, grad_susc, grad_inf, grad_time, *_param_grads(nets, grads))


class DistributedHotPathStep(torch.autograd.Function):
    """``HotPathStep`` on one rank of a multi-GPU job (``distributed.DistributedHotPath``): the same inputs and
    outputs for the rank's OWNED agents.  The forward is the multi-rank launch sequence; the backward runs the
    transposed passes with the forward's communication pattern (the cotangents of halo agents travel by the same
    all-to-all as their transmissions, the transposed per-venue sums by the same all-reduce), and ends with one
    all-reduce of the step's d loss / d log_beta - so every rank's ``log_beta.grad`` is the whole world's gradient,
    equal to the single-GPU run's.  Every rank must back-propagate the same graph (a loss built from the rank-summed
    result series: ``distributed_api.DistributedRunner``)."""

    @staticmethod
    def forward(ctx, env, susc, inf, time, *log_betas):
        hp, params_of, fixed, stage, exp_noise = (env[k] for k in ("hp", "params_of", "fixed", "stage", "exp_noise"))
        plan = hp.engine.plan
        n = plan.host.n_agents
        out_s, out_i, out_t = (t.detach().to(torch.float32).clone().contiguous() for t in (susc, inf, time))
        new_inf = torch.empty(n, dtype=torch.float32, device=plan.device)
        bufs = AgentBuffers(plan, **fixed, infection_time=out_t, is_infected=out_i, susceptibility=out_s,
                            transmission=hp.state["transmission"], q_transmission=hp.state["q_transmission"],
                            current_stage=stage)
        keep = _keep_sums(env)
        acc = torch.empty(n, dtype=torch.float32, device=plan.device) if keep else None
        hp.run_step(bufs, hp.engine.io(new_infected=new_inf, exp_noise=exp_noise, agent_sums=acc), params_of)
        ctx.env = env
        ctx.save_for_backward(susc.detach().to(torch.float32).contiguous(), inf.detach().to(torch.float32).contiguous(),
                              time.detach().to(torch.float32).contiguous())
        ctx.kept = (acc, _clone_forward_cum(plan, env["nets"])) if keep else None      # (cum: complete after the all-reduce)
        return out_s, out_i, out_t, new_inf

    @staticmethod
    def backward(ctx, g_susc, g_inf, g_time, g_new):
        env = ctx.env
        hp, params_of, fixed, stage, exp_noise, nets = (env[k] for k in ("hp", "params_of", "fixed", "stage",
                                                                          "exp_noise", "nets"))
        susc0, inf0, time0 = ctx.saved_tensors
        engine = hp.engine
        plan, lib, dev = engine.plan, N.load(), engine.plan.device
        n, n_ext = plan.host.n_agents, plan.host.n_ext_agents
        p = params_of(None)

        def f32(g):
            return None if g is None else g.detach().to(torch.float32).contiguous()

        g_susc, g_inf, g_time, g_new = f32(g_susc), f32(g_inf), f32(g_time), f32(g_new)
        ones = _ones(plan, n)
        scratch = torch.zeros(n_ext, dtype=torch.float32, device=dev)
        scratch_q = torch.zeros(n_ext, dtype=torch.float32, device=dev) if p.has_quarantine else None
        bufs = AgentBuffers(plan, **fixed, infection_time=time0, is_infected=inf0, susceptibility=ones,
                            transmission=scratch, q_transmission=scratch_q, current_stage=stage)
        if ctx.kept is not None:      # the forward's per-agent and per-venue sums, kept by the step
            acc, cum_fwd = ctx.kept
        else:
            acc = torch.empty(n, dtype=torch.float32, device=dev)
            cum_fwd = {}

            def keep_forward_sums():
                cum_fwd.update(_clone_forward_cum(plan, nets))

            # ---- recompute the forward of the two passes from the saved pre-state, across the ranks -----------
            p.transpose = 0
            engine.step_phase(bufs, p, engine.io(trans_susc=acc), 0)          # transmission (+ q * transmission)
            hp.sparse_passes(bufs, engine.io(trans_susc=acc), p, between=keep_forward_sums)
        # ---- elementwise adjoint of epilogue + sampler + infect_people (owned agents) ----------------------------
        x = torch.empty(n, dtype=torch.float32, device=dev)
        grad_susc = torch.empty(n, dtype=torch.float32, device=dev)
        grad_time = torch.empty(n, dtype=torch.float32, device=dev)
        N.check(lib.gj_adjoint_sample(n, N.ptr(susc0), N.ptr(time0), N.ptr(acc), N.ptr(exp_noise), int(p.seed),
                                      int(p.step), int(p.agent_offset), float(p.now), float(p.delta_time),
                                      N.ptr(g_susc), N.ptr(g_inf), N.ptr(g_time), N.ptr(g_new), N.ptr(x),
                                      N.ptr(grad_susc), N.ptr(grad_time), N.current_stream()), "gj_adjoint_sample")
        # ---- transposed passes on x = susc0 * ts_bar: one scale for the whole world -------------------------------
        scale = _power_of_two_scale(hp.all_reduce_max(x.abs().max().reshape(1)).reshape(()))
        scratch.zero_()
        scratch[:n].copy_(x / scale)
        tbar = torch.empty(n, dtype=torch.float32, device=dev)
        grads: List[torch.Tensor] = []

        def beta_gradients():
            grads.extend(_beta_gradients(plan, nets, env["betas"], cum_fwd, scale, weights_of=hp.venue_weights))

        p.transpose = 1
        try:
            engine.quarantine_transmission(bufs, p)                            # q * x for the masked sets
            hp.sparse_passes(bufs, engine.io(trans_susc=tbar), p, between=beta_gradients)
        finally:
            p.transpose = 0
        tbar = tbar * scale
        total = hp.all_reduce_sum(torch.stack(grads)) if grads else None        # fp64: the world's gradient
        grads32 = [total[i].to(torch.float32) for i in range(len(grads))]
        # ---- through the transmission profile (owned agents) --------------------------------------------------------
        grad_inf = torch.empty(n, dtype=torch.float32, device=dev)
        st0 = AgentBuffers(plan, **fixed, infection_time=time0, is_infected=inf0, susceptibility=ones,
                           transmission=scratch)
        N.check(lib.gj_adjoint_transmission(n, C.byref(st0.c), float(p.now), N.ptr(tbar), N.ptr(g_inf),
                                            N.ptr(grad_inf), N.ptr(grad_time), N.current_stream()),
                "gj_adjoint_transmission")
        return (None, grad_susc, grad_inf, grad_time, *_param_grads(nets, grads32))


class AllReduceSum(torch.autograd.Function):
    """Sum of a tensor over the ranks whose gradient is the identity: every rank evaluates the SAME loss on the
    summed value, so d loss / d (this rank's term) = d loss / d sum, which every rank already holds."""

    @staticmethod
    def forward(ctx, hp, x):
        return hp.all_reduce_sum(x.detach())

    @staticmethod
    def backward(ctx, g):
        return None, g


class NetworksForward(torch.autograd.Function):
    """The stand-alone ``InfectionNetworks.forward`` / ``InfectionNetwork.forward`` (base.py:61-84,118-141) as
    an autograd node: (transmission, susceptibility, *log_betas) -> not_infected_probs (or one network's
    trans_susc), differentiable w.r.t. all of them.  Same machinery as HotPathStep without the sampler."""

    @staticmethod
    def forward(ctx, env, transmission, susceptibility, *log_betas):
        engine, p, want = env["engine"], env["params"], env["want"]
        plan = engine.plan
        n = plan.host.n_agents
        dev = plan.device
        trans = torch.zeros(plan.host.n_ext_agents, dtype=torch.float32, device=dev)
        trans[:n].copy_(transmission.detach())
        susc = susceptibility.detach().to(device=dev, dtype=torch.float32).contiguous()
        bufs = AgentBuffers(plan, susceptibility=susc, transmission=trans, current_stage=env["stage"])
        out = torch.empty(n, dtype=torch.float32, device=dev)
        keep = _keep_sums(env) and plan.c.tiled is not None and bool(plan.c.tiled)
        acc = torch.empty(n, dtype=torch.float32, device=dev) if keep else None
        engine.quarantine_transmission(bufs, p)
        engine.venue_reduce(bufs, p)
        engine.agent_gather(bufs, p, engine.io(not_infected_probs=out, agent_sums=acc) if want == "probs"
                            else engine.io(trans_susc=out, agent_sums=acc), sample=False)
        snap = N.StepParams()
        C.memmove(C.byref(snap), C.byref(p), C.sizeof(N.StepParams))
        ctx.env = dict(env, params=snap)
        ctx.save_for_backward(trans, susc)
        ctx.kept = (acc, _clone_forward_cum(plan, env["nets"])) if keep else None
        return out

    @staticmethod
    def backward(ctx, g_out):
        env = ctx.env
        engine, p, nets, want = env["engine"], env["params"], env["nets"], env["want"]
        trans, susc = ctx.saved_tensors
        plan, dev = engine.plan, engine.plan.device
        n = plan.host.n_agents
        if plan.c.tiled is None or not bool(plan.c.tiled):
            raise NotImplementedError("the backward pass runs on the tiled layout")
        g = g_out.detach().to(torch.float32).contiguous()
        ones = _ones(plan, n)
        scratch = trans.clone()
        bufs = AgentBuffers(plan, susceptibility=ones, transmission=scratch, current_stage=env["stage"])
        if ctx.kept is not None:
            acc, cum_fwd = ctx.kept
        else:
            acc = torch.empty(n, dtype=torch.float32, device=dev)
            cum_fwd = _forward_sums(engine, p, bufs, acc, nets, compute_transmission=False)
        if want == "probs":                                       # clamp(exp(-clamp(ts, 1e-6, 100) * dt), 0, 1)
            ts = susc * acc
            inside = (ts >= 1e-6) & (ts <= 100.0)
            prob = torch.exp(-torch.clamp(ts, 1e-6, 100.0) * float(p.delta_time))
            ts_bar = torch.where(inside, g * (-float(p.delta_time)) * prob, torch.zeros_like(g))
        else:
            ts_bar = g
        grad_susc = ts_bar * acc
        tbar, grads = _transposed_passes(engine, p, bufs, scratch, (susc * ts_bar).contiguous(), nets, env["betas"],
                                         cum_fwd)
        return (None, tbar, grad_susc, *_param_grads(nets, grads))


class SymptomsStep(torch.autograd.Function):
    """(new_infected, current_stage, next_stage, time_to_next_stage) -> the three updated arrays, with the
    reference's gradient paths (symptoms.py:98,105-124,231-236): a loss on the stages - the deaths series of
    runner.py:198-215 - reaches ``log_beta`` through ``new_infected``.  All three outputs carry a gradient,
    as test_symptoms.py:208-231 expects.  Forward = ``gj_symptoms_update`` on
    fresh tensors; backward = ``gj_adjoint_symptoms`` replaying the branch each agent took."""

    @staticmethod
    def forward(ctx, env, new_infected, cur, nxt, ttn):
        lib = N.load()
        n = new_infected.numel()
        nw = new_infected.detach().to(torch.float32).contiguous()
        cur0, nxt0, ttn0 = (t.detach().to(torch.float32).contiguous() for t in (cur, nxt, ttn))
        out_c, out_x, out_t = cur0.clone(), nxt0.clone(), ttn0.clone()
        p = env["params"]
        N.check(lib.gj_symptoms_update(n, N.ptr(env["cls"]), N.ptr(nw), N.ptr(out_c), N.ptr(out_x), N.ptr(out_t),
                                       C.byref(p), N.ptr(env["progresses"]), N.ptr(env["dwell"]), N.current_stream()),
                "gj_symptoms_update")
        snap = N.SymptomsParams()                       # the caller reuses its struct: keep this call's clock / key
        C.memmove(C.byref(snap), C.byref(p), C.sizeof(N.SymptomsParams))
        ctx.env = {"cls": env["cls"], "progresses": env["progresses"], "dwell": env["dwell"], "params": snap,
                   "table": env.get("table")}
        ctx.save_for_backward(nw, cur0, nxt0, ttn0)
        return out_c, out_x, out_t

    @staticmethod
    def backward(ctx, g_cur, g_nxt, g_ttn):
        nw, cur0, nxt0, ttn0 = ctx.saved_tensors
        env = ctx.env
        n = nw.numel()

        def f32(g):
            return None if g is None else g.detach().to(torch.float32).contiguous()

        g_cur, g_nxt, g_ttn = f32(g_cur), f32(g_nxt), f32(g_ttn)
        g_cur_in, g_nxt_in, g_ttn_in, g_new = (torch.empty_like(nw) for _ in range(4))
        N.check(N.load().gj_adjoint_symptoms(n, N.ptr(env["cls"]), N.ptr(nw), N.ptr(cur0), N.ptr(nxt0), N.ptr(ttn0),
                                             C.byref(env["params"]), N.ptr(env["progresses"]), N.ptr(env["dwell"]),
                                             N.ptr(g_cur), N.ptr(g_nxt), N.ptr(g_ttn), N.ptr(g_cur_in),
                                             N.ptr(g_nxt_in), N.ptr(g_ttn_in), N.ptr(g_new), N.current_stream()),
                "gj_adjoint_symptoms")
        return None, g_new, g_cur_in, g_nxt_in, g_ttn_in
