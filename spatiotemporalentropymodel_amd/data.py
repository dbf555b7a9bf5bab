"""Training-data pipeline: Vimeo-septuplet datasets with the reference's augmentation semantics,
re-cut for the MI355X: the host decodes PNGs and draws the augmentation *scalars* (same RNG calls, same order as the
reference, so a seeded run picks the same crops / flips / quality maps), the device does everything per pixel.

    VimeoSepTuplet        stem/dataset_vidseq.py:24-96       shared random crop (torch RNG) + temporal flip (`random`)
    VimeoSepTuplet_QMap   stem_roi/stem_roi_dataset.py:13-163 crop + flip + quality map: uniform / gradation / Gaussians
    get_loader / get_loader_roi                               iterate device batches: (list of 7 x [B,3,c,c]) [, qmap [B,1,c,c]]

`__getitem__` returns the decoded uint8 frames plus the drawn parameters (no per-pixel host work); `device_batch`
uploads one pinned uint8 block and launches `stem_crop_u8_to_f32` (+ `stem_qmap_render`).  A 16-septuplet batch is
16 x 7 x 448 x 256 x 3 B = 38.5 MB over PCIe (0.6 ms) instead of 88 MB of cropped floats, and the loader threads only
run the PNG decoder.
"""
from __future__ import annotations

import os
import queue
import random
import threading

import numpy as np
import torch

from . import _lib
from . import functional as F

MODE_UNIFORM, MODE_GRADATION, MODE_GAUSSIAN = 0, 1, 2
MAX_GAUSSIANS = 20
QP = 4 + 4 * MAX_GAUSSIANS           # doubles per sample, layout in include/stem_hip.h (stem_qmap_render)


# ----------------------------------------------------------------------------- host: the reference's random draws
def draw_crop_flip_qmap(img_h, img_w, cropsize, training=True, p=0.3, level_range=(0, 100), level=0):
    """The draws of VimeoSepTuplet_QMap.__getitem__ in the reference's order (stem_roi_dataset.py:49-59, 88-93, 100-101,
    106-148), from Python's `random`: -> (top, left, flip, qmap parameter block [QP] float64)."""
    if training:
        top = random.randint(0, img_h - cropsize)
        left = random.randint(0, img_w - cropsize)
    else:
        top = int(round((img_h - cropsize) / 2.0))
        left = int(round((img_w - cropsize) / 2.0))
    flip = random.random() >= 0.5
    q = np.zeros(QP, dtype=np.float64)
    sample = random.random()
    if not training:
        q[0], q[1] = MODE_UNIFORM, level
        return top, left, flip, q
    hi = level_range[1]
    if sample < p:
        q[0] = MODE_UNIFORM
        tmp = random.random()
        if tmp < 0.01:
            q[1] = 0.0
        elif tmp < 0.20:
            q[1] = (hi + 1) * (1 - tmp)
        else:
            q[1] = (hi + 1) * random.random()
    elif sample < 2 * p:
        q[0] = MODE_GRADATION
        q[1] = random.random() * hi
        q[2] = random.random() * hi
        q[3] = 1.0 if random.random() < 0.5 else 0.0
    else:
        q[0] = MODE_GAUSSIAN
        n = int(1 + random.random() * 20)
        q[1] = n
        for k in range(n):
            q[4 + 4 * k] = cropsize * random.random()            # mu along rows
            q[5 + 4 * k] = cropsize * random.random()            # mu along columns
            q[6 + 4 * k] = 2000 * random.random() + 1000
            q[7 + 4 * k] = 2000 * random.random() + 1000
        q[2] = 0.5 * random.random() + 0.5
    return top, left, flip, q


def draw_crop_flip(img_h, img_w, cropsize):
    """VimeoSepTuplet's training draws: torchvision RandomCrop.get_params (two torch.randint calls unless the image
    already has the crop size) then the temporal flip from `random` (dataset_vidseq.py:12-13, 82-85)."""
    if img_h < cropsize or img_w < cropsize:
        raise ValueError(f"Required crop size {(cropsize, cropsize)} is larger than input image size {(img_h, img_w)}")
    if img_h == cropsize and img_w == cropsize:
        top = left = 0
    else:
        top = int(torch.randint(0, img_h - cropsize + 1, size=(1,)).item())
        left = int(torch.randint(0, img_w - cropsize + 1, size=(1,)).item())
    return top, left, random.random() >= 0.5


def load_septuplet_u8(path):
    """[7, H, W, 3] uint8 from <path>/f001.png .. f007.png (PIL decode, RGB)."""
    from PIL import Image
    frames = []
    for i in range(1, 8):
        with Image.open(os.path.join(path, f"f00{i}.png")) as im:
            frames.append(np.asarray(im.convert("RGB"), dtype=np.uint8))
    return np.stack(frames)


# ----------------------------------------------------------------------------- datasets
class _VimeoBase:
    TRAIN_LIST = "sep_trainlist.txt"

    def __init__(self, data_root, is_training, cropsize):
        self.data_root = data_root
        self.image_root = os.path.join(data_root, "sequences")
        self.training = bool(is_training)
        self.cropsize = int(cropsize)
        with open(os.path.join(data_root, self.TRAIN_LIST)) as f:
            self.trainlist = f.read().splitlines()
        with open(os.path.join(data_root, "sep_testlist.txt")) as f:
            self.testlist = f.read().splitlines()

    def __len__(self):
        return len(self.trainlist if self.training else self.testlist)

    def _frames(self, index):
        name = (self.trainlist if self.training else self.testlist)[index]
        return load_septuplet_u8(os.path.join(self.image_root, name))


class VimeoSepTuplet(_VimeoBase):
    """stem/dataset_vidseq.py:24-96.  Item: {"frames": uint8 [7,H,W,3], "crop": (top, left, size), "flip": bool}; in
    test mode the whole frame is kept (size None)."""

    def make_item(self, frames):
        H, W = frames.shape[1:3]
        if self.training:
            top, left, flip = draw_crop_flip(H, W, self.cropsize)
            return {"frames": frames, "crop": (top, left, self.cropsize), "flip": flip}
        return {"frames": frames, "crop": (0, 0, None), "flip": False}

    def __getitem__(self, index):
        return self.make_item(self._frames(index))


class VimeoSepTuplet_QMap(_VimeoBase):
    """stem_roi/stem_roi_dataset.py:13-163 (same constructor; the frame size is read from the files instead of the
    hard-coded 448x256).  As upstream, the temporal flip is drawn in test mode as well (:100-101)."""
    TRAIN_LIST = "vimeo_sep_trainlist_all.txt"

    def __init__(self, data_root, is_training=True, cropsize=256, level_range=(0, 100), level=0):
        super().__init__(data_root, is_training, cropsize)
        self.level_range, self.level, self.p = level_range, level, 0.3

    def make_item(self, frames):
        H, W = frames.shape[1:3]
        top, left, flip, q = draw_crop_flip_qmap(H, W, self.cropsize, self.training, self.p, self.level_range, self.level)
        return {"frames": frames, "crop": (top, left, self.cropsize), "flip": flip, "qmap": q, "inv_range": 1.0 / self.level_range[1]}

    def __getitem__(self, index):
        return self.make_item(self._frames(index))


# ----------------------------------------------------------------------------- device: one launch per batch
def render_qmaps(params, cropsize, device, inv_range=0.01):
    """params: [B, QP] float64 (host) -> [B,1,c,c] float32 on `device` (stem_qmap_render)."""
    params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, QP)
    assert int(_lib.hip().stem_qmap_params_per_sample()) == QP
    pd = torch.from_numpy(params).to(device)
    out = torch.empty((params.shape[0], 1, cropsize, cropsize), device=device, dtype=torch.float32)
    F._chk(_lib.hip().stem_qmap_render(pd.data_ptr(), out.data_ptr(), params.shape[0], cropsize, float(np.float32(inv_range)), F._stream()))
    return out


def crop_frames(frames_u8, crops, flips, cropsize, device):
    """frames_u8: uint8 [B,7,H,W,3] (host or device); crops [(top,left)], flips [bool] -> list of 7 x [B,3,c,c] float32."""
    src = torch.as_tensor(frames_u8)
    if src.device.type != "cuda":
        src = src.pin_memory().to(device, non_blocking=True)
    B, T, H, W, _ = src.shape
    prm = torch.tensor([[int(t), int(l), int(bool(f))] for (t, l), f in zip(crops, flips)], dtype=torch.int32).to(device)
    out = torch.empty((T, B, 3, cropsize, cropsize), device=device, dtype=torch.float32)
    F._chk(_lib.hip().stem_crop_u8_to_f32(src.data_ptr(), out.data_ptr(), prm.data_ptr(), B, T, H, W, cropsize, F._stream()))
    return [out[t] for t in range(T)]


def device_batch(items, device):
    """Collate dataset items and render them on `device`: -> images (list of 7 x [B,3,c,c]) or (images, qmap [B,1,c,c])
    -- the objects the training loops iterate over (stem/trainSTEM.py:194, stem_roi/train_stem_roi.py:499-503)."""
    frames = np.stack([it["frames"] for it in items])
    size = items[0]["crop"][2]
    if size is None:                                   # test mode of VimeoSepTuplet: whole frames (must be square-croppable)
        H, W = frames.shape[2:4]
        if H != W:
            raise ValueError("full-frame batches need square frames; evaluate non-square frames one GOP at a time")
        size = H
    images = crop_frames(frames, [it["crop"][:2] for it in items], [it["flip"] for it in items], size, device)
    if "qmap" not in items[0]:
        return images
    qmap = render_qmaps(np.stack([it["qmap"] for it in items]), size, device, items[0]["inv_range"])
    return images, qmap


class DeviceLoader:
    """Batches of a dataset rendered on the device, with `num_workers` host threads decoding PNGs ahead (the only
    per-pixel host work left).  Sample order is shuffled with `random` like DataLoader(shuffle=True) would; draws for
    the augmentation happen in the consumer thread, in batch order, so a seeded run is reproducible."""

    def __init__(self, dataset, batch_size, shuffle, num_workers, device, prefetch=2):
        self.ds, self.bs, self.shuffle, self.nw, self.device, self.prefetch = dataset, batch_size, shuffle, max(1, num_workers), device, prefetch

    def __len__(self):
        return (len(self.ds) + self.bs - 1) // self.bs

    def __iter__(self):
        order = list(range(len(self.ds)))
        if self.shuffle:
            random.shuffle(order)
        batches = [order[i:i + self.bs] for i in range(0, len(order), self.bs)]
        q = queue.Queue(maxsize=self.prefetch)

        def produce():
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(self.nw) as pool:
                for b in batches:
                    q.put(list(pool.map(self.ds._frames, b)))
            q.put(None)

        threading.Thread(target=produce, daemon=True).start()
        for _ in batches:
            decoded = q.get()
            yield device_batch([self.ds.make_item(fr) for fr in decoded], self.device)


def get_loader(mode, data_root, batch_size, shuffle, num_workers, cropsize=256, device="cuda"):
    """stem/dataset_vidseq.py:98-105"""
    return DeviceLoader(VimeoSepTuplet(data_root, is_training=(mode == "train"), cropsize=cropsize), batch_size, shuffle, num_workers, device)


def get_loader_roi(mode, data_root, batch_size, shuffle, num_workers, cropsize=256, level=0, device="cuda"):
    """stem_roi/stem_roi_dataset.py:156-163"""
    ds = VimeoSepTuplet_QMap(data_root, is_training=(mode == "train"), cropsize=cropsize, level=level)
    return DeviceLoader(ds, batch_size, shuffle, num_workers, device)
