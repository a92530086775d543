from . import DictConfig  # noqa
