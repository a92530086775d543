class Data(object):
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k in self.__dict__.keys() if not k.startswith('_')]


class Batch(Data):
    pass
