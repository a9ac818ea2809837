"""Checkpoint wire format of the reference's drivers, so that a `.pth` written
by the reference loads here and vice versa.

Writer: trainval_net_voc.py:488-500 -- torch.save of
    {'session', 'epoch' (= finished epoch + 1), 'model': state_dict, 'optimizer': state_dict,
     'pooling_mode': cfg.POOLING_MODE, 'class_agnostic'}
with the model's keys taken from `fasterRCNN.module` under DataParallel (so never 'module.'-
prefixed; a prefixed file from a hand-rolled save is still accepted on load).
Readers: resume trainval_net_voc.py:307-319, evaluation test_net_coco.py:275-279.
File name: '{dataset}_{backbone}_fasterRCNN_session-{s}_epoch-{e}_step-{k}.pth' (:489-493).
"""
import os

import torch

from .config import cfg


def checkpoint_name(output_dir, dataset, backbone, session, epoch, step):
    return os.path.join(output_dir, "{}_{}_fasterRCNN_session-{}_epoch-{}_step-{}.pth".format(
        dataset, backbone, session, epoch, step))


def _unwrap(model):
    return model.module if hasattr(model, "module") else model      # DataParallel / DDP


def save_checkpoint(path, model, optimizer, session, epoch, class_agnostic=True):
    """`epoch` is the epoch just finished; the file stores epoch + 1 like the reference."""
    torch.save({
        "session": session,
        "epoch": epoch + 1,
        "model": _unwrap(model).state_dict(),
        "optimizer": optimizer.state_dict(),
        "pooling_mode": cfg.POOLING_MODE,
        "class_agnostic": class_agnostic,
    }, path)


def load_checkpoint(path, model, optimizer=None, map_location="cpu", strict=True):
    """Restores model (and optimizer when given).  Returns (session, start_epoch, lr or None).
    Like the reference, a stored 'pooling_mode' overrides cfg.POOLING_MODE."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    sd = ckpt["model"]
    if sd and all(k.startswith("module.") for k in sd):
        sd = {k[len("module."):]: v for k, v in sd.items()}
    _unwrap(model).load_state_dict(sd, strict=strict)
    lr = None
    if optimizer is not None and "optimizer" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
        lr = optimizer.param_groups[0]["lr"]
    if "pooling_mode" in ckpt:
        cfg.POOLING_MODE = ckpt["pooling_mode"]
    return ckpt.get("session"), ckpt.get("epoch"), lr
