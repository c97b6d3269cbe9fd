"""Synthetic stand-ins for the datasets BASELINE.json's configs name (STL10, ImageNet-val), shaped like the
real ones -- image sizes, counts, label range, the torchvision ``Dataset(transform=...)`` protocol -- but made
of generated pixels: the real files cannot be fetched offline.  Used by tests/, bench.py and tools/rate_sweep.py.
"""
import numpy as np
import torch
from PIL import Image

# (H, W), weight -- the shape mix of ImageNet-val (50 000 JPEGs): about two thirds are 500 wide and 375 / 333 high,
# portraits are the transposes, the rest is a long tail from thumbnails to multi-megapixel photos.  The weights
# are approximate (the dataset is not available here); what matters for the path is that one batch holds many
# different sizes, both orientations, up- and down-scaling, and the occasional photo too large for one LDS band.
IMAGENET_VAL_SIZE_MIX = (
    ((375, 500), 0.52), ((333, 500), 0.11), ((500, 375), 0.09), ((500, 333), 0.03), ((334, 500), 0.03),
    ((332, 500), 0.02), ((500, 500), 0.015), ((400, 500), 0.02), ((480, 640), 0.02), ((281, 500), 0.01),
    ((357, 500), 0.01), ((376, 500), 0.01), ((500, 400), 0.01), ((768, 1024), 0.015), ((600, 800), 0.01),
    ((1200, 1600), 0.008), ((1536, 2048), 0.004), ((2448, 3264), 0.002), ((150, 200), 0.01), ((96, 128), 0.005),
    ((224, 224), 0.005), ((300, 225), 0.01), ((213, 320), 0.01), ((500, 313), 0.01), ((640, 427), 0.01),
)


class MixedSizeImages(torch.utils.data.Dataset):
    """ImageNet-val-shaped: image i has a size drawn from ``IMAGENET_VAL_SIZE_MIX`` by a hash of (seed, i) and
    pixels = a per-size base image XOR a per-image byte (cheap to make, all different).  ``__getitem__`` follows
    torchvision's ImageFolder: ``(transform(PIL image), target)``."""

    def __init__(self, n, transform=None, seed=0, pool=4, classes=1000):
        self.n, self.transform, self.seed, self.classes = int(n), transform, int(seed), int(classes)
        shapes = [s for s, _ in IMAGENET_VAL_SIZE_MIX]
        w = np.array([p for _, p in IMAGENET_VAL_SIZE_MIX], dtype=np.float64)
        rng = np.random.default_rng(seed)
        self.shape_of = rng.choice(len(shapes), size=self.n, p=w / w.sum())
        self.shapes = shapes
        self.pool = {}
        for k in np.unique(self.shape_of):
            h, wd = shapes[k]
            self.pool[int(k)] = [np.random.default_rng(seed * 1000 + int(k) * 10 + j)
                                 .integers(0, 256, (h, wd, 3), dtype=np.uint8) for j in range(pool)]
        self.targets = ((np.arange(self.n, dtype=np.int64) * 2654435761 + seed) >> 7) % self.classes

    def __len__(self):
        return self.n

    def array(self, i):
        base = self.pool[int(self.shape_of[i])]
        return base[i % len(base)] ^ np.uint8((i * 37 + 11) & 0xFF)

    def __getitem__(self, i):
        img = Image.fromarray(self.array(i))
        return (self.transform(img) if self.transform is not None else img), int(self.targets[i])


class Stl10Shaped(torch.utils.data.Dataset):
    """STL10-shaped: ``n`` RGB 96x96 images in 10 classes (STL10: 5 000 train / 8 000 test), held as one uint8
    array [N,3,96,96] like torchvision's STL10, ``__getitem__`` = ``(transform(PIL image), target)`` (torchvision
    STL10.__getitem__: ``Image.fromarray(np.transpose(img, (1, 2, 0)))``).  The class is recoverable from the
    pixels (a colour cast + a block pattern per class over noise), so a downstream LinearSVC has something to
    learn even through a random-weight tower."""

    def __init__(self, n, transform=None, seed=0, split_seed=0):
        rng = np.random.default_rng(seed * 7919 + split_seed)
        self.labels = rng.integers(0, 10, size=n).astype(np.int64)
        x = rng.integers(0, 256, size=(n, 96, 96, 3), dtype=np.uint8)
        for k in range(10):
            m = self.labels == k
            c = k % 3
            x[m, :, :, c] = (x[m, :, :, c] // 4 + 120 + 12 * (k // 3)).astype(np.uint8)
            r0 = 8 * k
            x[m, r0:r0 + 16, :, (c + 1) % 3] //= 8
        self.data = np.ascontiguousarray(x.transpose(0, 3, 1, 2))
        self.transform = transform

    def __len__(self):
        return len(self.data)

    def hwc(self):
        """-> uint8 [N,96,96,3] (for the tensor fast path)"""
        return np.ascontiguousarray(self.data.transpose(0, 2, 3, 1))

    def __getitem__(self, i):
        img = Image.fromarray(np.transpose(self.data[i], (1, 2, 0)))
        return (self.transform(img) if self.transform is not None else img), int(self.labels[i])
