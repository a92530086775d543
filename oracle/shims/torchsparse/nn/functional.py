def sphash(*a, **k):
    raise NotImplementedError


def sphashquery(*a, **k):
    raise NotImplementedError
