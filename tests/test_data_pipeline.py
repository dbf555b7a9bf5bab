"""Data pipeline: the host draws of spatiotemporalentropymodel_amd.data replay the reference
dataset's RNG calls (fixture: outputs of the reference's VimeoSepTuplet_QMap per `random.seed`), the oracle renders the
same quality maps, and -- on the GPU -- the HIP kernels reproduce both."""
import os
import random
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, assert_close

sys.path.insert(0, os.path.join(REPO, "oracle"))
import data_oracle as dorc  # noqa: E402

from spatiotemporalentropymodel_amd import data as D  # noqa: E402


def coordinate_frames(H=256, W=448):
    """The coordinate-coded septuplet the fixture generator wrote as PNGs (make_golden.py:_write_coordinate_septuplets)."""
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    return np.stack([np.stack([xx & 255, yy & 255, (xx >> 8) | ((yy >> 8) << 2) | (fr << 4)], -1) for fr in range(1, 8)]).astype(np.uint8)


def _draws(g, key):
    c, H, W, level = (int(v) for v in g["cfg"])
    out = []
    for seed in g[f"{key}:seeds"]:
        random.seed(int(seed))
        out.append(D.draw_crop_flip_qmap(H, W, c, training=(key == "train"), level=level))
    return c, H, W, out


@pytest.mark.parametrize("key", ["train", "test"])
def test_host_draws_and_oracle_maps_match_reference(golden, key):
    g = golden("roi_dataset.npz")
    c, H, W, draws = _draws(g, key)
    for i, (top, left, flip, q) in enumerate(draws):
        assert (top, left) == tuple(g[f"{key}:top_left"][i]), (i, top, left)
        assert list(g[f"{key}:order"][i]) == (list(range(7, 0, -1)) if flip else list(range(1, 8)))
        ref = g[f"{key}:qmap"][i]
        got = dorc.render_qmap(q, c)
        if int(q[0]) == D.MODE_GAUSSIAN:
            assert_close(got, ref, 1e-6, what=f"gaussian map seed {g[f'{key}:seeds'][i]}", floor=0.1)
        else:
            np.testing.assert_array_equal(got, ref)
    tags = set(str(t) for t in g["train:tags"])
    assert tags == {"zero", "hi", "uni", "grad", "gradT", "gauss"}


def test_vimeo_septuplet_draw_order():
    """VimeoSepTuplet: two torch.randint draws (rows first) then the flip from `random` (dataset_vidseq.py:12-13,82-85);
    parity unpinned: torchvision, whose RandomCrop.get_params the reference calls, is not installed here."""
    torch.manual_seed(5)
    random.seed(5)
    top, left, flip = D.draw_crop_flip(256, 448, 64)
    torch.manual_seed(5)
    random.seed(5)
    assert top == int(torch.randint(0, 193, (1,))) and left == int(torch.randint(0, 385, (1,))) and flip == (random.random() >= 0.5)
    assert D.draw_crop_flip(64, 64, 64)[:2] == (0, 0)
    with pytest.raises(ValueError):
        D.draw_crop_flip(32, 448, 64)


@pytest.mark.gpu
def test_hip_quality_maps_match_reference(golden):
    g = golden("roi_dataset.npz")
    for key in ("train", "test"):
        c, H, W, draws = _draws(g, key)
        maps = D.render_qmaps(np.stack([d[3] for d in draws]), c, "cuda:0").cpu().numpy()
        assert maps.shape == (len(draws), 1, c, c)
        for i, (_, _, _, q) in enumerate(draws):
            ref = g[f"{key}:qmap"][i]
            if int(q[0]) == D.MODE_GAUSSIAN:
                assert_close(maps[i, 0], ref, 2e-5, what="gaussian map", floor=0.1)
            else:
                np.testing.assert_array_equal(maps[i, 0], ref)


@pytest.mark.gpu
def test_hip_crop_flip_is_exact(golden):
    g = golden("roi_dataset.npz")
    c, H, W, draws = _draws(g, "train")
    fr = coordinate_frames(H, W)
    B = len(draws)
    frames = np.stack([fr] * B)
    images = D.crop_frames(frames, [(d[0], d[1]) for d in draws], [d[2] for d in draws], c, "cuda:0")
    assert len(images) == 7 and tuple(images[0].shape) == (B, 3, c, c) and images[0].is_contiguous()
    for b, (top, left, flip, _) in enumerate(draws):
        ref = dorc.crop_flip(fr, top, left, c, flip)
        for t in range(7):
            np.testing.assert_array_equal(images[t][b].cpu().numpy(), ref[t])
        px = (images[0][b] * 255).round().long().cpu()
        assert int(px[0, 0, 0]) | ((int(px[2, 0, 0]) & 3) << 8) == g["train:top_left"][b][1]
        assert [int(images[t][b][2, 0, 0] * 255 + 0.5) >> 4 for t in range(7)] == list(g["train:order"][b])


@pytest.mark.gpu
def test_device_loader_end_to_end(tmp_path):
    from PIL import Image
    names = ["00001/0001", "00001/0002", "00002/0001"]
    fr = coordinate_frames()
    for n in names:
        d = tmp_path / "sequences" / n
        d.mkdir(parents=True)
        for i in range(7):
            Image.fromarray(fr[i]).save(d / f"f00{i + 1}.png")
    for lst in ("vimeo_sep_trainlist_all.txt", "sep_trainlist.txt", "sep_testlist.txt"):
        (tmp_path / lst).write_text("\n".join(names) + "\n")
    random.seed(3)
    torch.manual_seed(3)
    batches = list(D.get_loader_roi("train", str(tmp_path), 2, False, 2, cropsize=64, device="cuda:0"))
    assert len(batches) == 2
    images, qmap = batches[0]
    assert len(images) == 7 and tuple(images[0].shape) == (2, 3, 64, 64) and tuple(qmap.shape) == (2, 1, 64, 64)
    assert tuple(batches[1][0][0].shape) == (1, 3, 64, 64)
    assert float(qmap.min()) >= 0 and float(qmap.max()) <= 1.01 and images[0].device.type == "cuda"
    random.seed(3)                                   # same seed -> same first batch (draws happen in the consumer, in order)
    again = next(iter(D.get_loader_roi("train", str(tmp_path), 2, False, 2, cropsize=64, device="cuda:0")))
    assert torch.equal(again[1], qmap) and all(torch.equal(a, b) for a, b in zip(again[0], images))
    stem_batches = list(D.get_loader("train", str(tmp_path), 3, True, 1, cropsize=128, device="cuda:0"))
    assert len(stem_batches) == 1 and len(stem_batches[0]) == 7 and tuple(stem_batches[0][0].shape) == (3, 3, 128, 128)
