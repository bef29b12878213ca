#!/usr/bin/env python3
"""Copy what `tools/profile_round.sh <tag>` left under gpurun_out/final/ into profiles/ (the tracked summaries):

    python tools/collect_profiles.py r03

  bench.json                  -> profiles/<tag>_c3_10m_bench.json            (the one JSON line, indented)
  stats/**/*kernel_stats.csv  -> profiles/<tag>_c3_10m_kernel_stats.csv
  pmc_fetch/, pmc_write/      -> profiles/pmc_traffic.json + <tag>_c3_10m_pmc_counters.csv   (tools/pmc_traffic.py)
  backward*.json              -> profiles/<tag>_c3_10m_backward.json, _backward_recompute.json
  backward_stats/**           -> profiles/<tag>_c3_10m_backward_kernel_stats.csv
  c2.json                     -> profiles/<tag>_c2_1m_bench.json
"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
DST = os.path.join(ROOT, "profiles")


def last_json_line(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.strip().startswith("{")]
    return json.loads(lines[-1])


def keep_json(name, out):
    p = os.path.join(SRC, name)
    if not os.path.exists(p):
        print("missing", name)
        return None
    d = last_json_line(p)
    with open(os.path.join(DST, out), "w") as f:
        json.dump(d, f, indent=1)
        f.write("\n")
    print("wrote", out)
    return d


def keep_stats(directory, out):
    found = glob.glob(os.path.join(SRC, directory, "**", "*kernel_stats.csv"), recursive=True)
    if not found:
        print("missing", directory)
        return
    shutil.copyfile(found[-1], os.path.join(DST, out))
    print("wrote", out)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    b = keep_json("bench.json", f"{tag}_c3_10m_bench.json")
    keep_stats("stats", f"{tag}_c3_10m_kernel_stats.csv")
    if os.path.isdir(os.path.join(SRC, "pmc_fetch")) and os.path.isdir(os.path.join(SRC, "pmc_write")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"),
                               os.path.join(SRC, "pmc_fetch"), os.path.join(SRC, "pmc_write"), tag])
        made = os.path.join(DST, f"{tag}_pmc_counters.csv")
        if os.path.exists(made):
            os.replace(made, os.path.join(DST, f"{tag}_c3_10m_pmc_counters.csv"))
    keep_json("backward.json", f"{tag}_c3_10m_backward.json")
    keep_json("backward_recompute.json", f"{tag}_c3_10m_backward_recompute.json")
    keep_stats("backward_stats", f"{tag}_c3_10m_backward_kernel_stats.csv")
    keep_json("c2.json", f"{tag}_c2_1m_bench.json")
    if b:
        r = b.get("roofline", {})
        print(f"headline: {b['ms_per_step']:.4f} ms/step = {b['value']:.0f} steps/s, step roofline "
              f"{b.get('step_roofline_frac', 0):.3f}; dominant {r.get('kernel')} frac {r.get('frac', 0):.3f}, traffic {r.get('traffic')}")


if __name__ == "__main__":
    main()
