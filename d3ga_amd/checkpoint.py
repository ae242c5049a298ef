"""Checkpoint files in the reference's on-disk layout (models/trainer.py:145-209).

A checkpoint is ``torch.save(((model_state_dict, optimizer_state_dict, scheduler_state_dict), iteration), path)`` at
``<run_dir>/checkpoints/chkpntNNNNNN.pth`` (``trainer.py:194-209``); ``restore`` picks the last file, or the one whose
first number equals ``iteration`` (``trainer.py:145-178``).  The modules of this package keep the reference's parameter
and buffer names, so files written by either side load on the other.
"""
import glob
import os
import re
from pathlib import Path

import torch


def checkpoint_path(run_dir, iteration):
    return os.path.join(run_dir, "checkpoints", "chkpnt" + str(iteration).zfill(6) + ".pth")


def save_checkpoint(run_dir, iteration, model, optimizer=None, scheduler=None, name=None):
    """trainer.py:194-209.  `name` (e.g. "/model.pth") is appended to run_dir like the reference's final save."""
    path = run_dir + name if name is not None else checkpoint_path(run_dir, iteration)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    params = (model.state_dict(), optimizer.state_dict() if optimizer is not None else {},
              scheduler.state_dict() if scheduler is not None else {})
    torch.save((params, iteration), path)
    return path


def find_checkpoint(run_dir, iteration=None):
    """Path the reference's restore() would load (None if there is no checkpoint)."""
    files = sorted(glob.glob(os.path.join(run_dir, "checkpoints", "*.pth")))
    if not files:
        return None
    path = files[-1]
    if iteration is not None:
        for f in files:
            nums = re.findall(r"\d+\.?\d*", Path(f).stem)
            if nums and int(float(nums[0])) == int(iteration):
                path = f
                break
    return path


def load_checkpoint(run_dir, model, optimizer=None, scheduler=None, iteration=None, strict=True, map_location=None):
    """trainer.py:145-178.  Returns the stored iteration (0 when nothing was found, like the reference)."""
    path = find_checkpoint(run_dir, iteration)
    if path is None:
        return 0
    (model_sd, optim_sd, sched_sd), first_iter = torch.load(path, weights_only=False, map_location=map_location)
    model.load_state_dict(model_sd, strict=strict)
    if optimizer is not None and optim_sd:
        optimizer.load_state_dict(optim_sd)
    if scheduler is not None and sched_sd:
        scheduler.load_state_dict(sched_sd)
    return first_iter
