class Discrete(object):
    def __init__(self, n):
        self.n = n


class Box(object):
    def __init__(self, low, high, shape=None):
        self.low, self.high, self.shape = low, high, shape
