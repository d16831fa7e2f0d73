"""Autocast fence for the public forwards.

The reference's trainer wraps `model(**inputs)` in `torch.cuda.amp.autocast(enabled=mixed_precision)` (`experiments/trainer.py:449`,
`recipes/default.yaml:89` sets it).  On this path the precision of every product is fixed by the HIP kernels (bf16 operands, fp32
accumulation, fp32 statistics / losses / master weights), so autocast has nothing to decide -- but left on it would recast the ATen
glue between the kernels (`softmax`, `sum`, `cat` promotions ...) and hand a kernel a dtype it does not take.  Every public forward
therefore runs with autocast switched off; an enabled `GradScaler` around it keeps working (the backward is linear in the scaled
loss and bf16 / fp32 have the exponent range for a 2^16 factor)."""
import functools

import torch


def no_autocast(fn):
    """Decorator: run `fn` with CUDA autocast disabled.  One flag read when autocast is off already (nested fences cost nothing)."""

    @functools.wraps(fn)
    def fenced(*args, **kwargs):
        if torch.is_autocast_enabled("cuda"):
            with torch.autocast("cuda", enabled=False):
                return fn(*args, **kwargs)
        return fn(*args, **kwargs)

    return fenced
