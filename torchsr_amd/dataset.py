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


def initialize_datasets(train_directory: str, batch_size: int = 64, crop_size: int = 96, upscale_factor: int = 4,
                        dataset_multiplier: int = 1, workers: int = 16, distributed: bool = False,
                        seed: int = 0) -> Tuple[DataLoader, DataLoader, int, int]:
    """dataset.py:364-428."""
    if train_directory.startswith('synthetic:'):
        n = int(train_directory.split(':')[1])
        n_test = max(batch_size, n // 10)
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
        test = _Pairs(images[:n_test], crop_size, upscale_factor, 1, True)

    def loader(ds, shuffle):
        sampler = DistributedSampler(ds, seed=seed, shuffle=shuffle) if distributed else None
        return DataLoader(ds, batch_size=batch_size, shuffle=shuffle and sampler is None, sampler=sampler,
                          num_workers=workers, drop_last=True, pin_memory=True,
                          persistent_workers=distributed and workers > 0)

    return loader(train, True), loader(test, False), len(train), len(test)
