"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

from neighborretr_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def problem(seed, B, Nt, Nv, M, device="cpu", blank_video=None, ragged=True):
    p = synth.make_problem(seed, B, Nt, Nv, M, ragged=ragged)
    if blank_video is not None and blank_video >= 0:
        p["video_mask"][blank_video] = 0
    return {k: torch.from_numpy(v).to(device) for k, v in p.items()}


def params(seed=7, device="cpu"):
    return {k: torch.from_numpy(v).to(device) for k, v in synth.make_params(seed).items()}


def noise(seed, B, Nt, Nv, device="cpu"):
    return {k: torch.from_numpy(v).to(device) for k, v in synth.make_noise(seed, B, Nt, Nv).items()}


def maxdiff(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max())
