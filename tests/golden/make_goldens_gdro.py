#!/usr/bin/env python3
"""Golden fixture for GroupDRO-weighted cell losses (`use_gdro=True`), produced by running the REFERENCE's
`losses.GroupDRO` through the live sequence of `LitVAE._cell_loss` (task.py:419-451) executed by hand -- task.py itself
needs pytorch_lightning, absent here.  Run ONCE in the build container (where /root/reference is mounted read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens_gdro.py

gdro_sequence.npz holds DATA only: per step the head outputs, targets, masks, donor groups and the training-step counter;
the expected cell loss of every step and both trackers' weights after it, in fp64."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")
sys.dont_write_bytecode = True
import losses as ref_losses  # noqa: E402

CELL_PROPERTIES = {"Dementia": {"discrete": True, "values": [0, 1]}, "BRAAK_AD": {"discrete": False, "values": [-1]},
                   "Sex": {"discrete": True, "values": [0, 1, 2]}}
GROUPS = [0, 1, 10, 11, 20, 21, 30, 31, 40, 41, 50, 51, 60, 61, 998, 999]
TRAIN_STEPS = [497, 498, 499, 500, 501, 502, 503]     # the weights switch on at 500
B = 160


def main():
    paired = [[0, 1, 60, 61], [10, 11, 50, 51], [20, 21, 40, 41], [30, 31]]
    gdro = {"Dementia": ref_losses.GroupDRO(balances=[[0, 1, 10, 11, 20, 21, 30, 31, 40, 41, 50, 51, 60, 61]]),
            "BRAAK_AD": ref_losses.GroupDRO(balances=paired)}
    ce = nn.CrossEntropyLoss(reduction="none", ignore_index=-100)
    mse = nn.MSELoss(reduction="none")
    gen = torch.Generator().manual_seed(1234)
    blob = {"train_steps": np.array(TRAIN_STEPS), "groups": np.array(GROUPS)}
    for t, train_step in enumerate(TRAIN_STEPS):
        pred = {"Dementia": torch.randn(B, 2, generator=gen, dtype=torch.float64) * 2,
                "BRAAK_AD": torch.randn(B, 1, generator=gen, dtype=torch.float64),
                "Sex": torch.randn(B, 3, generator=gen, dtype=torch.float64)}
        vals = torch.stack([torch.randint(0, 2, (B,), generator=gen).double(), torch.randn(B, generator=gen, dtype=torch.float64),
                            torch.randint(0, 3, (B,), generator=gen).double()], dim=1)
        vals[::17, 0] = -100                                   # unknown labels: ignore_index, masked out
        mask = (torch.rand(B, 3, generator=gen) > 0.15).double()
        mask[::17, 0] = 0
        group_idx = [GROUPS[i] for i in torch.randint(0, len(GROUPS), (B,), generator=gen).tolist()]
        total = 0.0
        for n, (k, prop) in enumerate(CELL_PROPERTIES.items()):                      # task.py:429-448
            weights = gdro[k].get_weights(group_idx) if (k in gdro and train_step >= 500) else 1.0
            if prop["discrete"]:
                loss = ce(pred[k], vals[:, n].to(torch.int64))
            else:
                loss = mse(torch.squeeze(pred[k]), vals[:, n])
            total = total + (weights * loss * mask[:, n]).mean()
            if k in gdro:
                gdro[k].update_batch(loss, mask[:, n], group_idx)
                gdro[k].update_weights()
        for k, v in pred.items():
            blob[f"s{t}/pred/{k}"] = v.numpy()
        blob[f"s{t}/vals"], blob[f"s{t}/mask"], blob[f"s{t}/group_idx"] = vals.numpy(), mask.numpy(), np.array(group_idx)
        blob[f"s{t}/cell_loss"] = np.float64(total)
        for k, g in gdro.items():
            blob[f"s{t}/weights/{k}"] = np.array([g.weights[q] for q in GROUPS], dtype=np.float64)
            blob[f"s{t}/running/{k}"] = np.array([g.running_loss[q] for q in GROUPS], dtype=np.float64)
        assert np.isfinite(blob[f"s{t}/cell_loss"]), t
    np.savez_compressed(os.path.join(HERE, "gdro_sequence.npz"), **blob)
    print("wrote gdro_sequence.npz:", {t: float(blob[f"s{t}/cell_loss"]) for t in range(len(TRAIN_STEPS))})


if __name__ == "__main__":
    main()
