"""Backward of fused_chain._ChainPool (placeholder until the kernels land)."""


def backward(ctx, gout):
    raise NotImplementedError("recompute-chain backward")
