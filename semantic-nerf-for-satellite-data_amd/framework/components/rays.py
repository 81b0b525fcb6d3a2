"""Named column access into the ray tensors -- mirror of framework/components/rays.py:7-64.

rays (N,8): origin 0:3 | dir 3:6 | near 6 | far 7;  extras (N,4): sun_d 0:3 | ts 3.
Names are matched by prefix exactly like the reference ("origins", "directions", "fars" work)."""
import torch

_RAY_COLS = (("origin", 0, 3), ("dir", 3, 6), ("near", 6, 7), ("far", 7, 8), ("sun_direction", 8, 11))
_EXTRA_COLS = (("sun_d", 0, 3), ("ts", 3, 4))


def _component(t: torch.Tensor, table, kind: str, name: str, value=None):
    for key, a, b in table:
        if name.startswith(key):
            if value is not None:
                t[:, a:b] = value
            return t[:, a:b]
    raise KeyError(f"Trying to access {kind} component with a unknown name: {name}")


def _satnerf_ray_component(rays: torch.Tensor, name: str, value=None):
    return _component(rays, _RAY_COLS, "ray", name, value)


def _satnerf_extras_component(extras: torch.Tensor, name: str, value=None):
    return _component(extras, _EXTRA_COLS, "extra", name, value)


ray_component_fn = _satnerf_ray_component
extras_component_fn = _satnerf_extras_component
