"""GPU-resident ray bank + on-device batch sampler.

Replaces the DataLoader over BaseRaysDataset (framework/pipelines.py:107-118, framework/datasets.py:214-266,
baseline/dataset/satnerf_dataset.py:122-133, semantic/dataset/semantic_dataset.py:83-90) whose per-ray
Python __getitem__ caps the reference's feed rate.  Row layout is the reference's: rays (R,8) f32,
rgbs (R,3) f32, extras (R,4) f32, semantic (R,1) uint8, semantic_sparsity_mask (R,) bool; depth set:
rays / depths (R,1) / weights (R,) / extras.  Shuffle = one randperm per epoch, epoch length = R // batch
(framework/util/train_util.py:15-16).  Under data parallelism every rank draws the SAME permutation
(seeded by epoch) and takes its contiguous slice of each global batch, so the union over ranks is
exactly the single-process batch."""
import numpy as np
import torch


def shard_bounds(n: int, rank: int, world: int):
    """contiguous shard [lo, hi) of a global batch of n rays; n must divide evenly (equal work per rank)."""
    if n % world != 0:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return rank * per, (rank + 1) * per


class GpuRayBank:
    def __init__(self, tensors: dict, n_classes: int = 5, car_cls_idx: int = 4, seed: int = 0, device=None, image_sizes=None):
        """`image_sizes`: H*W of every image of a test bank, in row order (the reference's test DataLoader hands over one image
        per step, framework/pipelines.py:120-129); without it `image()` cuts equal synthetic slices."""
        self.t = {k: (v.to(device) if device is not None else v) for k, v in tensors.items()}
        self.image_sizes = [int(x) for x in image_sizes] if image_sizes is not None else None
        if self.image_sizes is not None:
            if sum(self.image_sizes) != int(self.t["rays"].shape[0]) or min(self.image_sizes) <= 0:
                raise ValueError("image_sizes must be positive and sum to the number of rays in the bank")
            self._image_lo = np.concatenate([[0], np.cumsum(self.image_sizes)]).astype(np.int64)
        self.semantic_n_classes = n_classes
        self.car_cls_idx = car_cls_idx
        self.seed = seed
        self._perm = None
        self._perm_epoch = -1

    def __len__(self):
        return int(self.t["rays"].shape[0])

    def to(self, device):
        self.t = {k: v.to(device) for k, v in self.t.items()}
        self._perm = None
        return self

    @property
    def device(self):
        return self.t["rays"].device

    @staticmethod
    def synthetic(n_rays: int, n_images: int = 19, n_classes: int = 5, seed: int = 0, depth: bool = False, device=None):
        """SURVEY 8(d) synthetic distribution: origin~U(-1,1)^3, unit dir, near=0, far~U(0.5,1.5), one sun
        direction and ts per image, rgbs~U(0,1), P(car)=0.03."""
        rng = np.random.default_rng(seed)
        o = rng.uniform(-1, 1, (n_rays, 3))
        d = rng.standard_normal((n_rays, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        far = rng.uniform(0.5, 1.5, (n_rays, 1))
        rays = np.concatenate([o, d, np.zeros((n_rays, 1)), far], 1).astype(np.float32)
        el, az = np.radians(rng.uniform(30, 70, n_images)), np.radians(rng.uniform(90, 180, n_images))
        sun = np.stack([np.sin(az) * np.cos(el), np.cos(az) * np.cos(el), np.sin(el)], 1)
        img = rng.integers(0, n_images, n_rays)
        extras = np.concatenate([sun[img], img[:, None]], 1).astype(np.float32)
        t = {"rays": torch.from_numpy(rays), "extras": torch.from_numpy(extras)}
        if depth:
            t["depths"] = torch.from_numpy((rng.uniform(0.2, 1.0, (n_rays, 1)) * far).astype(np.float32))
            t["weights"] = torch.from_numpy(rng.uniform(0, 1, n_rays).astype(np.float32))
        else:
            t["rgbs"] = torch.from_numpy(rng.uniform(0, 1, (n_rays, 3)).astype(np.float32))
            sem = rng.integers(0, max(n_classes - 1, 1), (n_rays, 1))
            sem[rng.uniform(size=(n_rays, 1)) < 0.03] = n_classes - 1
            t["semantic"] = torch.from_numpy(sem.astype(np.uint8))
            t["semantic_sparsity_mask"] = torch.ones(n_rays, dtype=torch.bool)
        return GpuRayBank(t, n_classes=n_classes, car_cls_idx=n_classes - 1, seed=seed, device=device)

    def n_images(self, rays_per_image: int = None) -> int:
        if rays_per_image is None:
            if self.image_sizes is None:
                raise ValueError("this bank carries no image sizes: pass rays_per_image")
            return len(self.image_sizes)
        return len(self) // rays_per_image

    def image(self, i: int, rays_per_image: int = None, rank: int = 0, world: int = 1) -> dict:
        """rows of validation image i: the bank's own image i (`image_sizes`) when `rays_per_image` is None, else the i-th
        slice of `rays_per_image` rows (synthetic banks).  Under data parallelism each rank takes a contiguous slice of the
        image's rays (ragged tails allowed: validation sums carry their counts); an image with fewer rays than would give
        every rank at least one is refused (a zero-size launch has no defined result)."""
        if rays_per_image is None:
            lo, hi = int(self._image_lo[i]), int(self._image_lo[i + 1])
        else:
            lo, hi = i * rays_per_image, (i + 1) * rays_per_image
        per = -(-(hi - lo) // world)
        a, b = min(lo + rank * per, hi), min(lo + (rank + 1) * per, hi)
        if (hi - lo) <= (world - 1) * per:
            raise ValueError(f"validation image {i} has {hi - lo} rays: too few to give each of {world} ranks a slice")
        return {k: v[a:b] for k, v in self.t.items()}

    def steps_per_epoch(self, global_batch: int) -> int:
        return max(1, len(self) // global_batch)

    def batch(self, step: int, global_batch: int, rank: int = 0, world: int = 1, shuffle: bool = True) -> dict:
        """rows of global batch `step`, restricted to this rank's shard"""
        spe = self.steps_per_epoch(global_batch)
        epoch, it = divmod(step, spe)
        lo, hi = shard_bounds(global_batch, rank, world)
        base = it * global_batch
        if shuffle:
            if self._perm_epoch != epoch or self._perm is None:
                # drawn ON the bank's device (no host randperm + upload per epoch); every rank seeds the same generator
                # of the same device type, so all ranks hold the same permutation
                g = torch.Generator(device=self.device).manual_seed(self.seed * 1000003 + epoch)
                self._perm = torch.randperm(len(self), generator=g, device=self.device)
                self._perm_epoch = epoch
            idx = self._perm[base + lo: base + hi]
        else:
            idx = torch.arange(base + lo, base + hi, device=self.device) % len(self)
        return {k: v.index_select(0, idx) for k, v in self.t.items()}
