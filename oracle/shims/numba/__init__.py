"""Test-infrastructure stub (NOT product code): stands in for `numba` so the
reference's @njit bodies execute as plain NumPy/Python when oracle/gen_golden.py
imports the reference in the build container. Written for this repo."""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(f):
        return f
    return deco


jit = njit
