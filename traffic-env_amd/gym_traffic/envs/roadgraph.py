"""GridRoad: the m x n Manhattan grid of one-way road segments.

Host-side topology tables with the reference's attribute names (roadgraph.py:25-64): `len m n
roads train_roads intersections phases dest nexts locs entrypoints`.  Road numbering: four
direction blocks of m*n train roads (block d, cell (row, col) -> d*m*n + row*n + col; d = 0
eastbound, 1 westbound, 2 towards higher rows, 3 towards lower rows), then 2n + 2m exit roads.
The tables are built with array arithmetic over whole direction blocks; the device library builds
the same tables itself (csrc/tfx_hip.hip build_tables) and tests check the two against each other
and against tables captured from the reference.
"""
import numpy as np


class GridRoad(object):
    def __init__(self, m, n, l):
        self.m, self.n = int(m), int(n)
        self.len = np.float32(l)
        v = self.m * self.n
        self.intersections = v
        self.train_roads = 4 * v
        self.roads = self.train_roads + 2 * self.n + 2 * self.m
        R, r, m, n = self.roads, self.train_roads, self.m, self.n

        road = np.arange(R)
        self.phases = (road < 2 * v).astype(np.int32)           # E-W blocks are phase 1
        self.dest = np.where(road < r, road % v, -1).astype(np.int32)

        cell = np.arange(v)
        row, col = cell // n, cell % n
        nexts = np.full(R, -1, np.int64)
        nexts[0 * v:1 * v] = np.where(col < n - 1, cell + 1, r + n + row)
        nexts[1 * v:2 * v] = np.where(col > 0, v + cell - 1, r + 2 * n + m + row)
        nexts[2 * v:3 * v] = np.where(row < m - 1, 2 * v + cell + n, r + n + m + col)
        nexts[3 * v:4 * v] = np.where(row > 0, 3 * v + cell - n, r + col)
        self.nexts = nexts.astype(np.int32)
        self.locs = np.float32(l) * self._unit_segments(0.02)
        self.entrypoints = None

    def generate_entrypoints(self, choices):
        """Entry roads per side; bit k of `choices` set = side k closed (0 west, 1 east, 2 row 0,
        3 last row).  Order of the concatenation matters: spawns index into it."""
        m, n, v = self.m, self.n, self.m * self.n
        sides = (n * np.arange(m),
                 v + n * np.arange(1, m + 1) - 1,
                 2 * v + np.arange(n),
                 3 * v + n * (m - 1) + np.arange(n))
        keep = [s for k, s in enumerate(sides) if not (int(choices) >> k) & 1]
        self.entrypoints = (np.concatenate(keep) if keep else np.empty(0)).astype(np.int32)
        return self.entrypoints

    def _unit_segments(self, eps):
        """[roads, 2, 2] float32 start/end points in units of the road length (render only)."""
        m, n, v, R = self.m, self.n, self.m * self.n, self.roads
        seg = np.zeros((R, 2, 2), np.float32)
        cell = np.arange(v)
        row, col = (cell // n).astype(np.float32), (cell % n).astype(np.float32)
        z = np.float32(eps)

        def put(sl, x0, y0, x1, y1):
            seg[sl, 0, 0], seg[sl, 0, 1], seg[sl, 1, 0], seg[sl, 1, 1] = x0, y0, x1, y1
        put(slice(0, v), col - 1, row - z, col, row - z)
        put(slice(v, 2 * v), col + 1, row + z, col, row + z)
        put(slice(2 * v, 3 * v), col + z, row - 1, col + z, row)
        put(slice(3 * v, 4 * v), col - z, row + 1, col - z, row)
        b = 4 * v
        j = np.arange(n, dtype=np.float32)
        i = np.arange(m, dtype=np.float32)
        put(slice(b, b + n), j - z, 0, j - z, -1)
        put(slice(b + n, b + n + m), n - 1, i - z, n, i - z)
        put(slice(b + n + m, b + 2 * n + m), j + z, m - 1, j + z, m)
        put(slice(b + 2 * n + m, R), 0, i + z, -1, i + z)
        return seg
