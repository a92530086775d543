"""Stand-in for torch_geometric.data (v1.6.3 is not in the image): attribute bag with the item access the
reference's transforms use (``data[key]``, ``data[key] = v``, ``key in data``, ``data.keys``, ``num_nodes``)."""


class Data:
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in self.__dict__

    @property
    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None and not k.startswith('_')]

    @property
    def num_nodes(self):
        if '_num_nodes' in self.__dict__:
            return self.__dict__['_num_nodes']
        pos = self.__dict__.get('pos')
        return None if pos is None else pos.shape[0]

    @num_nodes.setter
    def num_nodes(self, n):
        self.__dict__['_num_nodes'] = n


class Batch(Data):
    pass
