"""render(): top view of one env, drawn with matplotlib (pyglet/gym's classic-control viewer, which
the reference uses at traffic_env.py:285-359, are not installed on the MI355X image).

Same picture as the reference: every road a line coloured by its light (green / yellow / red,
update_colors :335-346), every car a short bar from x back to x - l along its road (update_locs
:348-359, road geometry from GridRoad.locs).
"""
import numpy as np


def light_colours(graph, current_phase, elapsed, yellow_ticks):
    """[train_roads, 3] RGB per road (traffic_env.py:335-346)."""
    r = graph.train_roads
    dst = graph.dest[:r]
    red_phase = graph.phases[:r] == np.asarray(current_phase)[dst]
    fresh = np.asarray(elapsed)[dst] < yellow_ticks
    col = np.zeros((r, 3), np.float32)
    col[red_phase & fresh] = (1, 1, 0)
    col[red_phase & ~fresh] = (1, 0, 0)
    col[~red_phase & fresh] = (1, 0, 0)
    col[~red_phase & ~fresh] = (0, 1, 0)
    return col


def car_segments(graph, state, leading, lastcar, car_len):
    """[(x_front, y_front, x_back, y_back)] world coordinates of every live car."""
    C = state.shape[-1]
    segs = []
    for e in range(graph.roads):
        ld, lc = int(leading[e]), int(lastcar[e])
        if ld == lc:
            continue
        slots = list(range(ld + 1, lc + 1)) if ld < lc else list(range(ld + 1, C)) + list(range(1, lc + 1))
        xs = state[e, 0, slots]
        p0, p1 = graph.locs[e, 0], graph.locs[e, 1]
        u = (p1 - p0) / float(graph.len)
        for x in xs:
            a = p0 + u * float(x)
            b = p0 + u * float(x - car_len)
            segs.append((a[0], a[1], b[0], b[1]))
    return np.asarray(segs, np.float32).reshape(-1, 4)


class MatplotlibViewer(object):
    def __init__(self, graph):
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        self.plt = plt
        self.fig, self.ax = plt.subplots(figsize=(8, 8))

    def draw(self, graph, state, leading, lastcar, current_phase, elapsed, yellow_ticks, car_len, mode):
        ax = self.ax
        ax.clear()
        cols = light_colours(graph, current_phase, elapsed, yellow_ticks)
        for e in range(graph.roads):
            c = cols[e] if e < graph.train_roads else (0, 1, 0)
            ax.plot(graph.locs[e, :, 0], graph.locs[e, :, 1], color=tuple(c), linewidth=1)
        for x0, y0, x1, y1 in car_segments(graph, state, leading, lastcar, car_len):
            ax.plot([x0, x1], [y0, y1], color=(0, 0, 1), linewidth=5)
        ax.set_aspect('equal')
        self.fig.canvas.draw()
        if mode == 'rgb_array':
            buf = np.asarray(self.fig.canvas.buffer_rgba())
            return buf[..., :3].copy()
        return True

    def close(self):
        self.plt.close(self.fig)
