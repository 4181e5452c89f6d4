"""Synthetic inputs shared by the golden generator and the tests."""
import torch

from _weights import seeded_input


def skill_inputs():
    pred = seeded_input("skp", (3, 6, 32, 32, 1), 20, kind="uniform")
    target = seeded_input("skt", (3, 6, 32, 32, 1), 21, kind="uniform")
    # exact threshold hits (k/255 is what uint8 VIL frames decode to), zeros (most of a real VIL frame) and NaNs
    target[0, :, :8] = torch.tensor([16, 74, 133, 160, 181, 219, 15, 220]).float().view(1, 8, 1, 1).expand(6, 8, 32, 1) / 255
    pred[0, :, 8:16] = torch.tensor([16, 74, 133, 160, 181, 219, 73, 255]).float().view(1, 8, 1, 1).expand(6, 8, 32, 1) / 255
    target[1, :, 16:] = 0.0
    pred[2, 3, 5, 5, 0] = float("nan")
    target[2, 4, 6, 6, 0] = float("nan")
    return pred, target
