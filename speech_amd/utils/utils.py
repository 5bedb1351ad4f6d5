"""Recipe helpers with the reference's names (src/utils/utils.py:10-34): seeding and the loss plot."""
import random

import numpy as np
import torch


def set_seed(seed):
    """Seeds Python's, numpy's and torch's generators, as the recipes do before building a model."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def draw_loss_curve(train_loss, valid_loss=None, save_path='./loss.png'):
    """Training (and validation) loss per epoch as a PNG.  matplotlib is imported here, not at module import, and
    with the file-only Agg backend: training nodes have no display."""
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    curves = [("train", train_loss)] + ([("valid", valid_loss)] if valid_loss is not None else [])
    figure = plt.figure()
    for label, values in curves:
        values = values.detach().cpu().numpy() if isinstance(values, torch.Tensor) else np.asarray(values)
        plt.plot(np.arange(1, len(values) + 1), values, label=label)
    plt.xlabel('Epochs')
    plt.ylabel('Loss')
    plt.legend()
    plt.savefig(save_path, bbox_inches='tight')
    plt.close(figure)
