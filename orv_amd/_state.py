"""Process-wide bookkeeping shared by the host modules."""

# Bumped whenever parameter memory is rewritten behind torch's back (the fused optimizer kernels write parameters in place
# without touching tensor._version): host-side caches of derived weights (e.g. the packed QKV matrix) key on it.
weights_epoch = [0]


def bump_weights_epoch() -> None:
    weights_epoch[0] += 1
