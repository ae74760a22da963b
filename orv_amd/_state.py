"""Process-wide bookkeeping shared by the host modules."""

# Bumped whenever parameter memory is rewritten behind torch's back (the fused optimizer kernels write parameters in place
# without touching tensor._version): host-side caches of derived weights (e.g. the packed QKV matrix) key on it.
weights_epoch = [0]


def bump_weights_epoch() -> None:
    weights_epoch[0] += 1


# Gradient destinations registered by the fused optimizer (id(parameter) -> its segment of the flat bf16 gradient buffer, viewed in
# the parameter's shape): the hand-written backward writes a weight gradient straight into its segment instead of into a
# temporary the optimizer copies later (3.4 GB less traffic and memory per 2B step).
import weakref

import os

_grad_views = {}
_INPLACE_GRADS = os.environ.get("ORV_INPLACE_GRADS", "1") != "0"      # A/B switch


def register_grad_views(params, views) -> None:
    for p, v in zip(params, views):
        _grad_views[id(p)] = (weakref.ref(p), v)


def grad_view(param):
    """A FRESH view object on the parameter's gradient segment (autograd keeps an incoming gradient without copying only if
    nobody else holds the tensor object), or None when no optimizer registered one or the parameter already holds a gradient
    (gradient accumulation: the segment IS the accumulated gradient then and must not be overwritten)."""
    if not _INPLACE_GRADS:
        return None
    hit = _grad_views.get(id(param))
    if hit is None or hit[0]() is not param or param.grad is not None:
        return None
    return hit[1].view(param.shape)
