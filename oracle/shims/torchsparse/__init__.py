"""Test-infrastructure stub for the absent `torchsparse`."""


class SparseTensor:
    pass
