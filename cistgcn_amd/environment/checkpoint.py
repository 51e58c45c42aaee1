"""Checkpoint I/O with the reference's on-disk contract.

Reference behaviour restated (no code shared): `environment/utils.py:60-66` writes every state as `<name>_last.pth.tar`,
additionally `<name>_best.pth.tar` when `is_best`, and `<name>_epoch_%05d.pth.tar` when `save_all`; the state is the dict
of `train.py:186-191` {epoch, lr, err, metric_used_to_save, state_dict, optimizer}.  `environment/model_loader.py:7-35`
resumes from it: for class name `CISTGCN` it restores epoch / err / lr, loads the optimizer state when an optimizer is
given (and then reports lr None: the optimizer carries it) and loads the model by key.  A missing file prints a message
and returns None, exactly as the reference does.

Both optimizers are interchangeable on disk: `runtime.FlatAdam.state_dict()` writes `torch.optim.Adam`'s layout and
`FlatAdam.load_state_dict()` reads it, so checkpoints move between the reference and this implementation in both directions.
"""
from pathlib import Path

import torch


def make_checkpoint(epoch, model, optimizer, err, metric_used_to_save="mpjpe"):
    """The state dict of train.py:186-191."""
    return {"epoch": epoch, "lr": optimizer.param_groups[0]["lr"], "err": err, "metric_used_to_save": metric_used_to_save,
            "state_dict": model.state_dict(), "optimizer": optimizer.state_dict()}


def save_ckpt(state, is_best=True, save_all=False, file_name="ckpt.pth.tar"):
    """environment/utils.py:60-66: `_last` always, `_best` when `is_best`, `_epoch_%05d` when `save_all`."""
    name = str(file_name)
    written = [name.replace(".pth.", "_last.pth.")]
    if is_best:
        print("Saving a new BEST model")
        written.append(name.replace(".pth.", "_best.pth."))
    if save_all:
        written.append(name.replace(".pth.", "_epoch_%05d.pth." % state["epoch"]))
    for path in written:
        torch.save(state, path)
    return written


def load_params_from_model_path(model_path, model, optimizer=None, map_location=None):
    """environment/model_loader.py:7-35 for the CISTGCN class name (the only one on this path)."""
    model_file = Path(model_path)
    if not (model_file.exists() and model_file.is_file()):
        print("model file in general_config is not a file or does not exist")
        return None
    print("Loading model from %s" % model_file)
    if map_location is None:
        map_location = next(model.parameters()).device
    ckpt = torch.load(model_file, map_location=map_location, weights_only=False)
    if model.__class__.__name__ != "CISTGCN" or "state_dict" not in ckpt:
        model.load_state_dict(ckpt)                      # bare state dict (the reference's fall-through branch)
        return {"epoch": None, "lr": None, "err": {"mpjpe": None}, "model": model, "optimizer": optimizer}
    start_epoch, err_best, lr_now = ckpt["epoch"], ckpt["err"], ckpt["lr"]
    if optimizer is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
        lr_now = None
    print("model loaded at epoch %s with test error of %s" % (start_epoch, err_best["mpjpe"] if isinstance(err_best, dict) else err_best))
    # copy_ into the existing tensors: parameters re-homed by FlatAdam / baked into a captured step keep their addresses
    model.load_state_dict(ckpt["state_dict"])
    return {"epoch": start_epoch, "lr": lr_now, "err": err_best, "model": model, "optimizer": optimizer}
