from . import ListConfig  # noqa
