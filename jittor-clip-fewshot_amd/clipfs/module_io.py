"""Whole-module checkpoints in the layout Jittor's ``Module.save`` / ``Module.load`` use -- the stage-2 files of the
reference (slow_pace.py:1709-1713: ``channel_lp.save('test_pkl/channel.pkl')``, ``prompt_learner.save(...)``,
``clip_model.save(...)``; read back by test.py:1818-1821).

``Module.save(path)`` pickles ``{dotted parameter name: numpy array}`` of ``state_dict()``; ``Module.load(path)``
copies every entry whose name and shape match and reports the rest.  (Jittor 1.3.8.5 is not installable here, so
this layout is "parity unpinned": it restates jittor/__init__.py ``Module.save`` / ``load_parameters`` from memory; the
names are the attribute paths of the reference's module classes, which this package mirrors.)  Files are WRITTEN with
``pickle`` (plain dict of numpy arrays) and READ with the inert reader ``clipfs.safe_pkl`` -- nothing in a checkpoint
is ever executed.
"""
from __future__ import annotations

import os
import pickle
import sys
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

from . import safe_pkl


def module_state(module: torch.nn.Module, extra: Iterable[Tuple[str, torch.Tensor]] = ()) -> Dict[str, np.ndarray]:
    out = {k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}
    for k, v in extra:
        out[k] = v.detach().cpu().numpy().copy()
    return out


def save_module(module: torch.nn.Module, path: str, extra: Iterable[Tuple[str, torch.Tensor]] = ()) -> None:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump(module_state(module, extra), f, protocol=4)


def load_module(module: torch.nn.Module, path: str, ignore: Iterable[str] = ()) -> Tuple[list, list]:
    """Copy the checkpoint's tensors into ``module`` (in place: views and flat buffers stay bound).  Returns
    (missing names, unexpected names); like Jittor's ``load_parameters`` a mismatch is reported on stderr, not raised
    -- except a SHAPE mismatch of a matching name, which raises ValueError."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"File {path} does not exist.")
    data = safe_pkl.load(path)
    if not isinstance(data, dict):
        raise ValueError(f"{path}: expected a dict of arrays, found {type(data).__name__}")
    own = dict(module.state_dict())
    ignore = set(ignore)
    unexpected = []
    with torch.no_grad():
        for k, v in data.items():
            if k in ignore:
                continue
            if k not in own:
                unexpected.append(k)
                continue
            arr = np.ascontiguousarray(np.asarray(v))
            if arr.size == 1 and own[k].numel() == 1:
                arr = arr.reshape(tuple(own[k].shape))  # scalars (logit_scale): () and (1,) are the same parameter
            if tuple(arr.shape) != tuple(own[k].shape):
                raise ValueError(f"{path}: shape mismatch for {k}: expected {tuple(own[k].shape)}, found {tuple(arr.shape)}")
            own[k].copy_(torch.from_numpy(arr).to(own[k].dtype))
    missing = [k for k in own if k not in data]
    for kind, names in (("missing", missing), ("unexpected", unexpected)):
        if names:
            print(f"load {path}: {len(names)} {kind} key(s), e.g. {names[:3]}", file=sys.stderr)
    return missing, unexpected
