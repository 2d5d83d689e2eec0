"""The ONE high-priority side stream of a device, shared by everything in the process that overlaps work with the caller's
stream (the text tower of LoRATrainer, the view / MTA stages of ood.score_stream).

HIP maps streams onto a few hardware queues per priority class.  Every additional stream is another draw of that mapping:
measured on MI355X, two more high-priority streams created between two phases of one process put the f16 GEMM's internal
side stream (csrc/gemm_f16.hip) on the queue of the text tower and the cfg-5 step went from 93.9 to 142.7 ms without any
error.  So: one stream per device, created once, reused by everyone."""
from __future__ import annotations

import os
from typing import Dict

import torch

_SIDE: Dict[str, "torch.cuda.Stream"] = {}


def side_stream(device) -> "torch.cuda.Stream":
    dev = torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    key = f"cuda:{index}"
    if key not in _SIDE:
        prio = int(os.environ.get("CLIPFS_SIDE_STREAM_PRIORITY", "-1"))  # -1 = high (A/B aid: 0 = normal)
        _SIDE[key] = torch.cuda.Stream(device=torch.device("cuda", index), priority=prio)
    return _SIDE[key]
