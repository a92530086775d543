"""Test-infrastructure stub for the absent `hydra`."""


def initialize(*a, **k):
    raise NotImplementedError


def compose(*a, **k):
    raise NotImplementedError
