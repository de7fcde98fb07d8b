"""Training / test data for the CLI -- contract of torchsr/dataset.py:364-428.

``initialize_datasets`` returns ``(train_loader, test_loader, train_len, test_len)``; train batches
are ``(low_res, high_res)``, test batches ``(low_res, bicubic, high_res)``, float NCHW in [0, 1].
The reference's PIL / torchvision / sklearn pipeline is CPU image I/O and outside the accelerated
hot path; this module restates its behaviour with PIL + torch only (random
crop, horizontal / vertical flips, PIL bicubic x1/4 which quantises LR to 8 bits, 90/10 split) and
adds a synthetic source (``train_dir='synthetic:N'``) used by smoke runs and benchmarks.
"""
import os
import random
from typing import List, Tuple

import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

SUPPORTED_IMAGES = ('.jpg', '.jpeg', '.png', '.bmp')  # dataset.py:29


def _image_dataset(directory: str) -> List[str]:
    """dataset.py:32-52."""
    return [os.path.join(directory, f) for f in sorted(os.listdir(directory)) if f.lower().endswith(SUPPORTED_IMAGES)]


class _Pairs(Dataset):
    def __init__(self, images, crop_size, upscale_factor, multiplier, test):
        self.images, self.crop, self.up, self.mult, self.test = images, crop_size, upscale_factor, multiplier, test

    def __len__(self):
        return len(self.images) * self.mult

    def __getitem__(self, index):
        import numpy as np
        from PIL import Image
        img = Image.open(self.images[index % len(self.images)]).convert('RGB')
        w, h = img.size
        if w < self.crop or h < self.crop:
            img = img.resize((max(w, self.crop), max(h, self.crop)), Image.BICUBIC)
            w, h = img.size
        x0, y0 = random.randint(0, w - self.crop), random.randint(0, h - self.crop)
        hr = img.crop((x0, y0, x0 + self.crop, y0 + self.crop))
        if not self.test:
            if random.random() < 0.5:
                hr = hr.transpose(Image.FLIP_LEFT_RIGHT)
            if random.random() < 0.5:
                hr = hr.transpose(Image.FLIP_TOP_BOTTOM)
        lr = hr.resize((self.crop // self.up, self.crop // self.up), Image.BICUBIC)
        to_t = lambda im: torch.from_numpy(np.asarray(im, dtype='float32') / 255.0).permute(2, 0, 1).contiguous()  # noqa: E731
        if self.test:
            return to_t(lr), to_t(lr.resize((self.crop, self.crop), Image.BICUBIC)), to_t(hr)
        return to_t(lr), to_t(hr)


class _Synthetic(Dataset):
    def __init__(self, n, crop_size, upscale_factor, test, seed):
        self.n, self.crop, self.up, self.test, self.seed = n, crop_size, upscale_factor, test, seed

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        hr = torch.rand(3, self.crop, self.crop, generator=g)
        lr = torch.nn.functional.interpolate(hr[None], scale_factor=1 / self.up, mode='bicubic', antialias=True,
                                             align_corners=False).clamp(0, 1)[0]
        if self.test:
            bic = torch.nn.functional.interpolate(lr[None], scale_factor=self.up, mode='bicubic',
                                                  align_corners=False).clamp(0, 1)[0]
            return lr, bic, hr
        return lr, hr


class DeviceLoader:
    """Train / test batches produced ON the MI355X.

    The reference runs RandomCrop + flips + PIL bicubic x1/4 in 16 DataLoader workers per GPU for every
    sample (dataset.py:88-99,121-125); at ~11 ms per batch-16 step that pipeline, not the GPU, sets the
    pace.  Here every image is decoded once, kept in HBM as uint8 HWC, and a batch is two kernels:
    ``srx_crop_flip_u8`` (crop + flips + ToTensor) and ``srx_bicubic_down`` (antialiased Keys bicubic, the
    filter PIL's BICUBIC and ``F.interpolate(..., antialias=True)`` use; LR rounded to 8 bits like the PIL
    image the reference converts back).  PIL's resampler also rounds its horizontal pass to 8 bits, so
    low-resolution pixels agree with the reference's to the last 8-bit step, not bit for bit.
    Same contract as the DataLoader it replaces: ``len()``, iteration yields ``(low_res, high_res)`` or
    ``(low_res, bicubic, high_res)`` float NCHW in [0, 1]; the last partial batch is dropped when training
    (fixed hipGraph shapes) and kept when testing.
    """

    def __init__(self, images, device, batch_size: int, crop_size: int, upscale_factor: int, test: bool, seed: int,
                 multiplier: int = 1, rank: int = 0, world_size: int = 1):
        from . import _lib
        self._lib = _lib
        self.device, self.batch, self.crop, self.up, self.test = device, batch_size, crop_size, upscale_factor, test
        self.images = [self._fit(im).to(device) for im in images]          # uint8 [H][W][3]
        self.sizes = [(int(im.shape[0]), int(im.shape[1])) for im in self.images]
        self.order = [i for _ in range(multiplier) for i in range(len(self.images))]
        self.rank, self.world = rank, world_size
        self.rng = random.Random(seed * 7919 + rank)
        self.epoch = 0

    def _fit(self, im: torch.Tensor) -> torch.Tensor:
        """Images smaller than the crop are enlarged first (the reference's RandomCrop would raise)."""
        h, w = int(im.shape[0]), int(im.shape[1])
        if h >= self.crop and w >= self.crop:
            return im.contiguous()
        f = torch.nn.functional.interpolate(im.permute(2, 0, 1)[None].float(), size=(max(h, self.crop), max(w, self.crop)),
                                            mode='bicubic', align_corners=False)
        return f[0].permute(1, 2, 0).clamp(0, 255).round().to(torch.uint8).contiguous()

    def _shard(self, order):
        """This rank's samples: what ``DistributedSampler`` hands out (dataset.py:279,343-360) -- the order padded to a
        multiple of the world size by wrapping around, then every world-th sample -- so that every rank holds the
        same number of samples (none is empty, none runs an extra batch and waits in a collective alone)."""
        if self.world > 1 and len(order) % self.world:
            pad = self.world - len(order) % self.world
            order = order + (order * (-(-pad // len(order))))[:pad]
        return order[self.rank::self.world]

    def __len__(self) -> int:
        n = -(-len(self.order) // self.world)  # the same on every rank
        # the reference's train loader drops nothing either (dataset.py:280-293), but a replayed hipGraph needs
        # one batch shape, so the last partial TRAIN batch is dropped; the eager test pass keeps it (:345-360)
        return -(-n // self.batch) if self.test else n // self.batch

    def __iter__(self):
        order = list(self.order)
        if not self.test:
            random.Random(self.epoch * 104729 + 17).shuffle(order)  # same permutation on every rank
        self.epoch += 1
        order = self._shard(order)
        call, dev = self._lib.call, self.device
        for b in range(len(self)):
            idx = order[b * self.batch:(b + 1) * self.batch]
            meta = []
            for i in idx:
                h, w = self.sizes[i]
                top, left = self.rng.randint(0, h - self.crop), self.rng.randint(0, w - self.crop)
                hflip = 0 if self.test else int(self.rng.random() < 0.5)
                vflip = 0 if self.test else int(self.rng.random() < 0.5)
                meta.append([h, w, top, left, hflip, vflip])
            ptrs = torch.tensor([self.images[i].data_ptr() for i in idx], dtype=torch.int64).to(dev)
            meta_t = torch.tensor(meta, dtype=torch.int32).to(dev)
            stream = torch.cuda.current_stream().cuda_stream
            hr = torch.empty((len(idx), 3, self.crop, self.crop), dtype=torch.float32, device=dev)
            call('srx_crop_flip_u8', ptrs.data_ptr(), meta_t.data_ptr(), hr.data_ptr(), len(idx), self.crop, stream)
            lc = self.crop // self.up
            lr = torch.empty((len(idx), 3, lc, lc), dtype=torch.float32, device=dev)
            call('srx_bicubic_down', hr.data_ptr(), lr.data_ptr(), len(idx), 3, self.crop, self.crop, self.up, 1, stream)
            if self.test:  # the bicubic up-sampled image of TestData (dataset.py:175-190): only ever displayed
                bic = torch.nn.functional.interpolate(lr, size=(self.crop, self.crop), mode='bicubic',
                                                      align_corners=False).clamp(0, 1)
                yield lr, bic, hr
            else:
                yield lr, hr


def _decode(path: str) -> torch.Tensor:
    import numpy as np
    from PIL import Image
    return torch.from_numpy(np.asarray(Image.open(path).convert('RGB'), dtype='uint8').copy())


def initialize_device_datasets(train_directory: str, device, batch_size: int = 64, crop_size: int = 96,
                               upscale_factor: int = 4, dataset_multiplier: int = 1, distributed: bool = False,
                               seed: int = 0, rank: int = 0, world_size: int = 1):
    """``initialize_datasets`` with the augmentation on the device (``--device-data``).  ``synthetic:N`` draws N
    seeded random 2*crop x 2*crop images."""
    if train_directory.startswith('synthetic:'):
        n = int(train_directory.split(':')[1])
        g = torch.Generator().manual_seed(seed)
        images = [torch.randint(0, 256, (2 * crop_size, 2 * crop_size, 3), generator=g, dtype=torch.uint8)
                  for _ in range(n + max(1, n // 10))]
    else:
        paths = _image_dataset(train_directory)
        if not paths:
            raise RuntimeError(f'no images ({", ".join(SUPPORTED_IMAGES)}) found in {train_directory}')
        random.Random(seed or None).shuffle(paths)
        images = [_decode(p) for p in paths]
    n_test = max(1, len(images) // 10)
    if not distributed:
        rank, world_size = 0, 1
    train = DeviceLoader(images[n_test:] or images, device, batch_size, crop_size, upscale_factor, False, seed,
                         dataset_multiplier, rank, world_size)
    # dataset.py:343-360: the test set is multiplied like the train set, sharded over the ranks, nothing dropped
    test = DeviceLoader(images[:n_test], device, batch_size, crop_size, upscale_factor, True, seed + 1,
                        dataset_multiplier, rank, world_size)
    return train, test, len(train.order), len(test.order)


def initialize_datasets(train_directory: str, batch_size: int = 64, crop_size: int = 96, upscale_factor: int = 4,
                        dataset_multiplier: int = 1, workers: int = 16, distributed: bool = False,
                        seed: int = 0) -> Tuple[DataLoader, DataLoader, int, int]:
    """dataset.py:364-428."""
    if train_directory.startswith('synthetic:'):
        n = int(train_directory.split(':')[1])
        n_test = max(1, n // 10)
        train = _Synthetic(n, crop_size, upscale_factor, False, seed)
        test = _Synthetic(n_test, crop_size, upscale_factor, True, seed + 1)
        workers = 0
    else:
        images = _image_dataset(train_directory)
        if not images:
            raise RuntimeError(f'no images ({", ".join(SUPPORTED_IMAGES)}) found in {train_directory}')
        rng = random.Random(seed or None)
        rng.shuffle(images)
        n_test = max(1, len(images) // 10)  # sklearn train_test_split(test_size=0.1), dataset.py:412
        train = _Pairs(images[n_test:] or images, crop_size, upscale_factor, dataset_multiplier, False)
        test = _Pairs(images[:n_test], crop_size, upscale_factor, dataset_multiplier, True)  # dataset.py:343-345

    def loader(ds, shuffle, drop_last):
        sampler = DistributedSampler(ds, seed=seed, shuffle=shuffle) if distributed else None
        return DataLoader(ds, batch_size=batch_size, shuffle=shuffle and sampler is None, sampler=sampler,
                          num_workers=workers, drop_last=drop_last, pin_memory=True,
                          persistent_workers=distributed and workers > 0)

    # train: the replayed hipGraph needs one batch shape, so the last partial batch is dropped; test: eager, and the
    # reference's test loader drops nothing (dataset.py:345-360) -- a shard smaller than --batch-size still gets tested
    return loader(train, True, True), loader(test, False, False), len(train), len(test)
