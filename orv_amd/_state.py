"""Process-wide bookkeeping shared by the host modules."""

# Bumped whenever parameter memory is rewritten behind torch's back (the fused optimizer kernels write parameters in place
# without touching tensor._version): host-side caches of derived weights (e.g. the packed QKV matrix) key on it.
weights_epoch = [0]


def bump_weights_epoch() -> None:
    weights_epoch[0] += 1


# Bumped whenever ANY module registers a parameter (construction, ``m.weight = nn.Parameter(...)``, ``load_state_dict(assign=True)``:
# all go through ``Module.register_parameter``).  Caches that hold Parameter OBJECTS (GraphedTransformer's parameter list) re-collect
# when it moves - a replaced Parameter keeps neither the identity nor the storage of the old one, and the old object's ``_version`` /
# ``data_ptr`` never change again.
param_epoch = [0]
_param_hook = []


def watch_parameter_registration() -> None:
    """Install (once per process) torch's global parameter- AND module-registration hooks that bump ``param_epoch``:
    ``m.weight = nn.Parameter(...)`` goes through ``register_parameter``; attaching a pre-built submodule (``blk.ff = other``,
    replacing ``transformer_blocks``) goes through ``register_module`` / ``Module.__setattr__``.  Edits that bypass both
    (``del m.weight``, writes into ``_parameters`` / ``_modules``, ``ModuleList`` truncation via ``del``) are caught by the structural
    fingerprint ``GraphedTransformer._weights_version`` keeps next to the epoch."""
    if _param_hook:
        return
    from torch.nn.modules.module import (register_module_module_registration_hook,
                                         register_module_parameter_registration_hook)

    def _bump(module, name, value):
        param_epoch[0] += 1
        return None

    _param_hook.append(register_module_parameter_registration_hook(_bump))
    _param_hook.append(register_module_module_registration_hook(_bump))


# Gradient destinations registered by the fused optimizer (id(parameter) -> its segment of the flat bf16 gradient buffer, viewed in
# the parameter's shape): the hand-written backward writes a weight gradient straight into its segment instead of into a
# temporary the optimizer copies later (3.4 GB less traffic and memory per 2B step).
import os
import weakref

# id(parameter) -> (weakref(parameter), weakref(owning optimizer), segment index).  Only weak references: a deleted optimizer
# (resume, re-creation, tests) releases its flat buffers, and its entries die with it.
_grad_views = {}
_INPLACE_GRADS = os.environ.get("ORV_INPLACE_GRADS", "1") != "0"      # A/B switch


def register_grad_views(params, owner) -> None:
    """``owner`` (the fused optimizer) exposes ``_grad_segment(i)`` (the i-th view of its flat gradient buffer) and a set
    ``_handed`` of segment indices handed out for in-place writing since its last ``step()`` / ``zero_grad()``."""
    for k in [k for k, (pr, orf, _) in _grad_views.items() if pr() is None or orf() is None]:
        del _grad_views[k]                               # dead parameters / optimizers
    oref = weakref.ref(owner)
    for i, p in enumerate(params):
        _grad_views[id(p)] = (weakref.ref(p), oref, i)


def grad_view(param):
    """A FRESH view object on the parameter's gradient segment (autograd keeps an incoming gradient without copying only if
    nobody else holds the tensor object), or None - the caller then allocates a temporary and autograd accumulates - when
      * no live optimizer registered one,
      * the parameter already holds a gradient (gradient accumulation: the segment IS the accumulated gradient), or
      * the segment was already handed out since the optimizer's last ``step()`` / ``zero_grad()``: a second backward node in
        the same graph (the model called twice, or ``torch.autograd.grad`` followed by ``backward``) would overwrite the first
        node's gradient and autograd would then sum two aliases of one tensor (2 g_B instead of g_A + g_B)."""
    if not _INPLACE_GRADS:
        return None
    hit = _grad_views.get(id(param))
    if hit is None or hit[0]() is not param or param.grad is not None:
        return None
    owner = hit[1]()
    if owner is None:
        del _grad_views[id(param)]
        return None
    if hit[2] in owner._handed:
        return None
    owner._handed.add(hit[2])
    return owner._grad_segment(hit[2]).view(param.shape)
