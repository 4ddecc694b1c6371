#!/usr/bin/env python
"""Alias of main_retrieval.py: the reference's entry file is main.py while its README launches
main_retrieval.py (and training/trainer.py:192 does `from main import save_model`)."""
import os

import torch

from main_retrieval import get_args, main  # noqa: F401


def save_model(epoch, args, model, type_name=""):
    """Weights-only checkpoint, same file naming as main.py:161-187."""
    to_save = model.module if hasattr(model, "module") else model
    name = "pytorch_model.bin.{}{}".format("" if type_name == "" else type_name + ".", epoch)
    path = os.path.join(args.output_dir, name)
    torch.save(to_save.state_dict(), path)
    return path


if __name__ == "__main__":
    main()
