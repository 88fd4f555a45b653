registry = {}


def register(id, entry_point, **kw):
    registry[id] = entry_point
