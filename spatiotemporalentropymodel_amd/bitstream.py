"""Bitstream container of the reference tooling (compressai_examples/codec.py:63-220), byte-compatible:

    2 bytes  header: model id | (metric << 4 | quality-1)           get_header / parse_header  (:93-119)
    2 x u32  original (h, w)                                         big-endian                 (:181-182)
    3 x u32  (shape[0], shape[1], n_strings)                                                    (:183-184)
    n x      u32 length + raw bytes of each string (one batch element per string list)          (:185-187)

The STEM scripts keep strings in memory (stem/evalSTEM.py:115-119); `write_frame` / `read_frame` put one coded frame
(I or P, any of the models here) into that layout, and `write_sequence` / `read_sequence` simply concatenate frame
records behind a frame count so that a GOP can be exchanged as one file.  `pad` / `crop` are the centred multiple-of-64
padding helpers (:122-151).  Host-side byte formatting only: there is no device work here.
"""
import struct

import torch
import torch.nn.functional as F

# compressai/zoo/__init__.py:17-24 in declaration order (ids are positions in that dict)
MODEL_NAMES = ("bmshj2018-factorized", "bmshj2018-hyperprior", "mbt2018-mean", "mbt2018", "cheng2020-anchor", "cheng2020-attn")
model_ids = {k: i for i, k in enumerate(MODEL_NAMES)}
metric_ids = {"mse": 0}


def inverse_dict(d):
    assert len(d.keys()) == len(set(d.keys()))
    return {v: k for k, v in d.items()}


def write_uints(fd, values, fmt=">{:d}I"):
    fd.write(struct.pack(fmt.format(len(values)), *values))


def write_uchars(fd, values, fmt=">{:d}B"):
    fd.write(struct.pack(fmt.format(len(values)), *values))


def read_uints(fd, n, fmt=">{:d}I"):
    return struct.unpack(fmt.format(n), _read_exact(fd, n * struct.calcsize("I")))


def read_uchars(fd, n, fmt=">{:d}B"):
    return struct.unpack(fmt.format(n), _read_exact(fd, n * struct.calcsize("B")))


def write_bytes(fd, values, fmt=">{:d}s"):
    if len(values) == 0:
        return
    fd.write(struct.pack(fmt.format(len(values)), values))


def read_bytes(fd, n, fmt=">{:d}s"):
    return struct.unpack(fmt.format(n), _read_exact(fd, n * struct.calcsize("s")))[0]


def _read_exact(fd, n):
    buf = fd.read(n)
    if len(buf) != n:
        raise ValueError(f"truncated bitstream: wanted {n} bytes, got {len(buf)}")
    return buf


def get_header(model_name, metric, quality):
    """1 byte model id, 4 bits metric, 4 bits quality-1 (codec.py:93-102)."""
    if model_name not in model_ids:
        raise ValueError(f'unknown model "{model_name}"')
    if metric not in metric_ids:
        raise ValueError(f'unknown metric "{metric}"')
    code = (metric_ids[metric] << 4) | (quality - 1 & 0x0F)
    return model_ids[model_name], code


def parse_header(header):
    """codec.py:105-119"""
    model_id, code = header
    quality = (code & 0x0F) + 1
    metric = code >> 4
    return inverse_dict(model_ids)[model_id], inverse_dict(metric_ids)[metric], quality


def write_frame(fd, header, original_size, shape, strings):
    """One coded frame in the layout of codec._encode (:178-187).  `strings` is the model's list of per-latent string
    lists; like the reference tool the record holds batch element 0 of each."""
    write_uchars(fd, header)
    write_uints(fd, (int(original_size[0]), int(original_size[1])))
    write_uints(fd, (int(shape[0]), int(shape[1]), len(strings)))
    for s in strings:
        write_uints(fd, (len(s[0]),))
        write_bytes(fd, s[0])


def read_frame(fd):
    """-> (model, metric, quality), original_size, shape, strings  (codec._decode :200-209)"""
    header = parse_header(read_uchars(fd, 2))
    original_size = read_uints(fd, 2)
    shape = read_uints(fd, 2)
    n_strings = read_uints(fd, 1)[0]
    strings = []
    for _ in range(n_strings):
        n = read_uints(fd, 1)[0]
        strings.append([read_bytes(fd, n) if n else b""])
    return header, original_size, shape, strings


def write_sequence(fd, frames):
    """frames: iterable of (header, original_size, shape, strings); a u32 frame count, then the frame records."""
    frames = list(frames)
    write_uints(fd, (len(frames),))
    for fr in frames:
        write_frame(fd, *fr)


def read_sequence(fd):
    return [read_frame(fd) for _ in range(read_uints(fd, 1)[0])]


def pad(x, p=2 ** 6):
    """Zero-pad H, W up to multiples of p, centred (codec.py:122-136)."""
    h, w = x.size(2), x.size(3)
    H, W = (h + p - 1) // p * p, (w + p - 1) // p * p
    left, top = (W - w) // 2, (H - h) // 2
    return F.pad(x, (left, W - w - left, top, H - h - top), mode="constant", value=0)


def crop(x, size):
    """Inverse of pad (codec.py:139-151)."""
    H, W = x.size(2), x.size(3)
    h, w = size
    left, top = (W - w) // 2, (H - h) // 2
    return F.pad(x, (-left, -(W - w - left), -top, -(H - h - top)), mode="constant", value=0)
