"""CPU restatement of the quality-map synthesis and crop/flip of the reference's ROI dataset
(stem_roi/stem_roi_dataset.py:62-72, 89-101, 106-152).  TEST INFRASTRUCTURE ONLY (tests/, smoke); pinned against
tests/golden/roi_dataset.npz, which holds outputs of the reference class itself.

The parameter block is the one spatiotemporalentropymodel_amd.data.draw_crop_flip_qmap produces
(include/stem_hip.h: stem_qmap_render)."""
import numpy as np
import torch
from torch.distributions.multivariate_normal import MultivariateNormal


def _grid(c):
    # stem_roi_dataset.py:62-72: grid[..., 0] = row index, grid[..., 1] = column index (int64)
    x1, x2 = torch.tensor(range(c)), torch.tensor(range(c))
    g1, g2 = torch.meshgrid(x1, x2, indexing="ij")
    return torch.cat([g1.view(c, c, 1), g2.view(c, c, 1)], dim=-1)


def render_qmap(q, c, level_range_hi=100):
    """q: float64 parameter block -> float32 [c, c] map in 0..1 (:106-152)."""
    mode = int(q[0])
    qmap = np.zeros((c, c), dtype=float)
    if mode == 0:
        qmap[:] = q[1]
    elif mode == 1:
        qmap = np.tile(np.linspace(q[1], q[2], c), (c, 1)).astype(float)
        if q[3] != 0.0:
            qmap = qmap.T
    else:
        grid = _grid(c)
        for k in range(int(q[1])):
            mu_x, mu_y, var_x, var_y = (float(v) for v in q[4 + 4 * k: 8 + 4 * k])
            m = MultivariateNormal(torch.tensor([mu_x, mu_y]), torch.tensor([[var_x, 0], [0, var_y]]))
            qmap += torch.exp(m.log_prob(grid)).numpy()
        qmap *= 100 / qmap.max() * q[2]
    out = torch.FloatTensor(np.ascontiguousarray(qmap))
    out *= 1 / level_range_hi
    return out.numpy()


def crop_flip(frames_u8, top, left, c, flip):
    """frames uint8 [7,H,W,3] -> float32 [7,3,c,c]: crop, ToTensor (byte / 255), reversed order when flip (:89-101)."""
    out = frames_u8[:, top:top + c, left:left + c, :].astype(np.float32) / np.float32(255)
    out = np.ascontiguousarray(out.transpose(0, 3, 1, 2))
    return out[::-1].copy() if flip else out
