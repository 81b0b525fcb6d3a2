"""Training driver -- mirror of run/training.py:13-41:
    python -m snerf_amd.run.training start_training <run.toml> <pipeline.toml>
(under torch.distributed.run for several GPUs of one node)."""
import sys

import torch

from .. import parallel
from ..framework.configs import load_configs
from ..framework.pipelines import load_pipeline, run_pipeline


def start_training(run_config_fp, pipeline_config_fp):
    cfgs = load_configs(run_config_fp, pipeline_config_fp)
    start_pipeline_cfgs(cfgs)


def start_pipeline_cfgs(cfgs, max_steps=None):
    rank, ws, device = parallel.init_distributed()
    if cfgs.run.deterministic:
        torch.manual_seed(0)
    pipeline = load_pipeline(cfgs)
    return run_pipeline(pipeline, cfgs, device=device, max_steps=max_steps)


if __name__ == "__main__":
    if len(sys.argv) != 4 or sys.argv[1] != "start_training":
        sys.exit("usage: python -m snerf_amd.run.training start_training <run.toml> <pipeline.toml>")
    start_training(sys.argv[2], sys.argv[3])
