"""Test-infrastructure stub for the absent `omegaconf`."""


class OmegaConf:
    pass


class ListConfig(list):
    pass


class DictConfig(dict):
    pass
