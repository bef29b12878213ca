"""The reference's example_scripts/run_model.py flow on N MI355X GPUs (one process per GPU):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        examples/run_distributed.py [config.yaml]

Without a config the packaged default parameters and the reference's 769-agent test world are used.
"""
import os
import sys

import torch
import torch.distributed as dist
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradabm-june_amd"))

from grad_june_amd.defaults import default_parameters  # noqa: E402
from grad_june_amd.distributed_api import DistributedRunner  # noqa: E402


def main():
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=int(os.environ.get("RANK", "0")),
                            world_size=int(os.environ.get("WORLD_SIZE", "1")), device_id=device)
    if len(sys.argv) > 1:
        with open(sys.argv[1]) as f:
            params = yaml.safe_load(f)
    else:
        params = default_parameters(str(device))
    params["system"]["device"] = str(device)
    torch.manual_seed(0)                                   # the same seed on every rank
    runner = DistributedRunner.from_parameters(params)
    with torch.no_grad():
        results, is_infected_local = runner()              # results: identical on every rank
    if dist.get_rank() == 0:
        print("cases per timestep:", [int(c) for c in results["cases_per_timestep"].tolist()])
        print("deaths per timestep:", [int(c) for c in results["deaths_per_timestep"].tolist()])
    sys.stdout.flush()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
