"""Run / pipeline configuration -- mirror of framework/configs.py:15-176 (field names and defaults of
RunConfig; the pipeline config class is chosen by the `pipeline = "pkg.mod.Class"` string and built
by Class.init_config, exactly as MainConfig does at :69-75).  TOML is read with tomli."""
import importlib
import os
from typing import List, Literal, Optional, Union

import tomli
from pydantic import BaseModel


class RunConfig(BaseModel):
    gpu_id: Union[int, List[int]] = 0
    device_req_free: Union[bool, int] = True
    gpu_max_memory_fraction: float = 0.95
    max_train_steps: int = 100000
    save_every_n_epochs: int = 1
    train_n_workers: int = 0
    val_n_workers: int = 4
    num_sanity_val_steps: int = 1
    check_val_every_n_epoch: int = 1
    shuffle_dataset: Union[bool, int] = True
    float32_matmul_precision: Literal["highest", "high", "medium"] = "high"
    deterministic: Union[bool, int] = False
    render_solid_background: Union[bool, int] = False
    run_name_postfix: str = ""
    experiment_category: str = ""
    resume_from_ckpoint: Union[bool, int] = False
    ckpoint_fp: Optional[str] = None
    dataset_name: Optional[str] = None
    dataset_limit_train_images: Union[int, bool] = False
    run_name: Optional[str] = None
    workspace_dp: Optional[str] = None
    cache_dp: Optional[str] = None
    datasets_dp: Optional[str] = None
    run_dp: Optional[str] = None
    # --- additions of this build (synthetic GPU-resident ray bank; no DFC2019 data in this environment)
    synthetic_rays: int = 1 << 20
    synthetic_images: int = 19
    synthetic_seed: int = 0


def _load_toml(fp):
    with open(fp, "rb") as f:
        return tomli.load(f)


class MainConfig:
    def __init__(self, run_ifp=None, pipeline_ifp=None, run: dict = None, pipeline: dict = None) -> None:
        self.run = RunConfig(**(run if run is not None else _load_toml(run_ifp)))
        data = pipeline if pipeline is not None else _load_toml(pipeline_ifp)
        name = data["pipeline"].split(".")
        module = importlib.import_module(".".join(name[:-1]))
        self.pipeline = getattr(module, name[-1]).init_config(data)

    def __str__(self) -> str:
        return f"RunConfig({self.run}) \nPipelineConfig({self.pipeline})"


def load_configs(run_config_fp: str, pipeline_config_fp: str):
    for fp in (run_config_fp, pipeline_config_fp):
        if not os.path.isfile(fp):
            raise FileNotFoundError(f"config file not found: {fp}")
    return MainConfig(run_config_fp, pipeline_config_fp)
