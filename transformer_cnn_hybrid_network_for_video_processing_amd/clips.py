"""Clip input pipeline: the step before the hot path once synthetic data is replaced.

The reference's clip loader exists only as bytecode (``__pycache__/dataset.cpython-38.pyc``: ``CSVDataset.__getitem__`` src
L106-113 turns one CSV row of T frame paths into a list of T ``[3,S,S]`` tensors, ``DataloaderSequential.load_images`` src
L124-127 batches them with ``shuffle=True``, i.e. a T-major list of ``[B,3,S,S]``); its source analogue for single images is
``Dataloader.py:8-46`` (pandas CSV -> ``PIL.Image.open`` -> ``Resize`` -> ``ToTensor``).  torchvision and cv2 are not part of
this image, so the decode/resize leg below uses PIL directly and is "parity unpinned" against those libraries' resamplers; what IS
pinned is ``ToTensor`` (uint8 HWC -> float CHW / 255), done here on the GPU after the PCIe copy.

What is MI355X-specific: frames cross PCIe as uint8 from pinned host memory (19 MB instead of 77 MB per config-2 batch: ~0.3 ms
instead of ~1.2 ms on a 63 GB/s link, against a ~1.6 ms training step), on a dedicated copy stream, two batches deep, and are
expanded to the fp32 ``[B,T,3,H,W]`` clip tensor by ``hyb_frames_u8hwc_to_f32chw`` on that stream -- the training stream only
waits on an event.  No CPU fallback for the device leg.
"""
import csv
import os

import numpy as np
import torch

from ._lib import lib


class ClipCSVDataset(torch.utils.data.Dataset):
    """One CSV row = the T frame paths of a clip (dataset.pyc src L86-113; column layout of Datasets/generateDataset.py:4-25: one
    path per cell, no header).  ``__getitem__`` -> (uint8 array [T, size, size, 3], label) -- decoded and resized on the host."""

    def __init__(self, csv_path, size=224, labels=None, root=None):
        with open(csv_path, newline="") as f:
            self.rows = [[c for c in row if c] for row in csv.reader(f) if row]
        if not self.rows:
            raise ValueError(f"{csv_path} holds no clips")
        t = {len(r) for r in self.rows}
        if len(t) != 1:
            raise ValueError(f"clips of different lengths in {csv_path}: {sorted(t)}")
        self.size, self.labels, self.root = int(size), labels, root

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        from PIL import Image
        frames = []
        for p in self.rows[i]:
            with Image.open(os.path.join(self.root, p) if self.root else p) as im:
                frames.append(np.asarray(im.convert("RGB").resize((self.size, self.size), Image.BILINEAR), dtype=np.uint8))
        return np.stack(frames), (0 if self.labels is None else int(self.labels[i]))


class SyntheticClipSource:
    """Endless seeded uint8 clip batches [B,T,H,W,3] + labels [B] (BASELINE's synthetic data, in the form a decoder produces)."""

    def __init__(self, batch, frames, size, num_classes=8, seed=0, distinct=4):
        g = np.random.default_rng(seed)
        self.batches = [(g.integers(0, 256, (batch, frames, size, size, 3), dtype=np.uint8), g.integers(0, num_classes, (batch,), dtype=np.int64))
                        for _ in range(distinct)]

    def __iter__(self):
        i = 0
        while True:
            yield self.batches[i % len(self.batches)]
            i += 1


def collate_clips(samples):
    """[(uint8 [T,H,W,3], label)] -> (uint8 [B,T,H,W,3], int64 [B]) for torch.utils.data.DataLoader(collate_fn=...)."""
    return np.stack([s[0] for s in samples]), np.asarray([s[1] for s in samples], dtype=np.int64)


def t_major(x):
    """The reference's collated layout: a list of T tensors [B,3,H,W] (views of the [B,T,3,H,W] clip tensor)."""
    return [x[:, t] for t in range(x.shape[1])]


class ClipPipeline:
    """Iterates over ``source`` (any iterable of (uint8 [B,T,H,W,3], labels [B]) host batches -- a DataLoader over ClipCSVDataset
    with ``collate_fn=collate_clips``, or SyntheticClipSource) and yields device tensors (clips fp32 [B,T,3,H,W] in [0,1], labels),
    ``depth`` batches ahead of the consumer: pinned staging buffers, async H2D on its own stream, ToTensor on the device."""

    def __init__(self, source, device="cuda", depth=2):
        self.source, self.depth = source, max(1, int(depth))
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ClipPipeline feeds the MI355X HIP path: device must be cuda (there is no CPU fallback)")
        self.stream = torch.cuda.Stream(device=self.device)
        self._slots = None

    def _make_slots(self, shape):
        B, T, H, W, C = shape
        self._slots = [dict(host=torch.empty(shape, dtype=torch.uint8).pin_memory(), host_y=torch.empty(B, dtype=torch.int64).pin_memory(),
                            dev_u8=torch.empty(shape, dtype=torch.uint8, device=self.device),
                            x=torch.empty(B, T, C, H, W, dtype=torch.float32, device=self.device),
                            y=torch.empty(B, dtype=torch.int64, device=self.device), ready=torch.cuda.Event(), free=torch.cuda.Event())
                       for _ in range(self.depth + 1)]
        for s in self._slots:
            s["free"].record(torch.cuda.current_stream(self.device))

    def _issue(self, slot, batch):
        frames, labels = batch
        frames = np.ascontiguousarray(frames)
        if frames.dtype != np.uint8 or frames.ndim != 5:
            raise ValueError("expected uint8 clips [B,T,H,W,C]")
        B, T, H, W, C = frames.shape
        s = self._slots[slot]
        s["free"].synchronize()                               # the consumer has finished with this slot's device tensors
        s["host"].numpy()[...] = frames                       # pageable -> pinned (the decoder could write here directly)
        s["host_y"].numpy()[...] = np.asarray(labels, dtype=np.int64)
        with torch.cuda.stream(self.stream):
            s["dev_u8"].copy_(s["host"], non_blocking=True)
            s["y"].copy_(s["host_y"], non_blocking=True)
            lib.call("hyb_frames_u8hwc_to_f32chw", s["dev_u8"].data_ptr(), s["x"].data_ptr(), B * T, H, W, C, self.stream.cuda_stream)
            s["ready"].record(self.stream)

    def __iter__(self):
        it = iter(self.source)
        pending = []                                          # slots in flight, oldest first
        nxt = 0
        exhausted = False
        while True:
            while not exhausted and len(pending) < self.depth:
                try:
                    batch = next(it)
                except StopIteration:
                    exhausted = True
                    break
                if self._slots is None:
                    self._make_slots(tuple(batch[0].shape))
                self._issue(nxt, batch)
                pending.append(nxt)
                nxt = (nxt + 1) % len(self._slots)
            if not pending:
                return
            slot = pending.pop(0)
            s = self._slots[slot]
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(s["ready"])                        # GPU-side wait only: the host does not block
            yield s["x"], s["y"]
            s["free"].record(torch.cuda.current_stream(self.device))      # everything the consumer enqueued on x / y so far
