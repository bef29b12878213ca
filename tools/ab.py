#!/usr/bin/env python3
"""Kernel A/B on ONE GPU box: run bench.py once per variant (a build of the library made with different -D flags
and/or different bench.py arguments), back to back, sharing a world cache, and print one table.  Boxes differ by
up to 10 % on the same binary, so variants are only comparable inside one invocation.

    python tools/ab.py [--steps 30] [--rounds 2] label[:DEFINE=VAL,...][@bench args] ...

e.g.  python tools/ab.py base unroll8:GJ_UNROLL_NARROW=8 "nodirect@--direct off"
Variants are built here (hipcc cross-compiles without a GPU) by `--build-only`; on the GPU box the prebuilt
libraries under gradabm-june_amd/grad_june_amd/lib/variants/ are used.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gradabm-june_amd", "csrc")
VAR = os.path.join(ROOT, "gradabm-june_amd", "grad_june_amd", "lib", "variants")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def parse_variant(v):
    bench_args = []
    if "@" in v:
        v, rest = v.split("@", 1)
        bench_args = rest.split()
    label, _, defs = v.partition(":")
    return label, [d for d in defs.split(",") if d], bench_args


def lib_path(label):
    return os.path.join(VAR, f"libgj_{label}.so")


def lib_of(label):
    p = lib_path(label)
    return p if os.path.exists(p) or label == "base" else lib_path("base")     # argument-only variants run the base build


def build(label, defs):
    os.makedirs(VAR, exist_ok=True)
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *FLAGS, *[f"-D{d}" for d in defs], "-o", lib_path(label),
           os.path.join(CSRC, "gradjune_hip.hip"), os.path.join(CSRC, "gj_compile.hip")]
    subprocess.run(cmd, check=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--common", default="--device-compile --no-cpu-baseline", help="bench.py arguments of every variant")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ab.jsonl"))
    a = ap.parse_args()
    vs = [parse_variant(v) for v in a.variants]
    if a.build_only:
        for label, defs, bargs in vs:
            if defs or label == "base" or not bargs:
                build(label, defs)
                print("built", lib_path(label))
        return
    rows = []
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    for rnd in range(a.rounds):
        for label, defs, bargs in vs:
            env = dict(os.environ, GJ_LIB_PATH=lib_of(label))
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", "5",
                   "--world-cache", "/tmp/gj_worlds", *a.common.split(), *bargs]
            r = subprocess.run(cmd, env=env, capture_output=True, text=True)
            if r.returncode != 0:
                print(f"{label}: FAILED\n{r.stderr[-2000:]}", flush=True)
                continue
            d = json.loads(r.stdout.strip().splitlines()[-1])
            row = {"label": label, "round": rnd, "ms_per_step": d["ms_per_step"], "kernel_ms": d["kernel_ms"],
                   "full_ms": (d.get("full_step") or {}).get("ms_per_step"), "checksum": d.get("state_checksum")}
            rows.append(row)
            with open(a.out, "a") as f:
                f.write(json.dumps(row) + "\n")
            km = " ".join(f"{k[:9]}={v * 1e3:6.1f}" for k, v in d["kernel_ms"].items())
            full = f"   full step {row['full_ms'] * 1e3:7.1f}" if row["full_ms"] else ""
            print(f"[r{rnd}] {label:24s} {d['ms_per_step'] * 1e3:7.1f} us/step   {km}{full}   infected={row['checksum']['infected']:.0f}",
                  flush=True)
    print("\nbest of rounds:")
    for label, _, _ in vs:
        mine = [r for r in rows if r["label"] == label]
        if mine:
            b = min(mine, key=lambda r: r["ms_per_step"])
            km = " ".join(f"{k[:9]}={v * 1e3:6.1f}" for k, v in b["kernel_ms"].items())
            print(f"{label:24s} {b['ms_per_step'] * 1e3:7.1f} us/step   {km}")


if __name__ == "__main__":
    main()
