"""Shared input cases for the parity tests (synthetic fields, adversarial shapes)."""
import numpy as np

import oracle_lib as ol


def field(h, w, seed):
    return ol.random_field(h, w, seed)


def smooth_field(h, w, seed, octaves=4):
    """Band-limited field: long plateaus and valleys (many rings per level), unlike iid noise."""
    rng = np.random.default_rng(seed)
    acc = np.zeros((h, w))
    for o in range(octaves):
        step = 2 ** (octaves - o + 1)
        gh, gw = h // step + 2, w // step + 2
        g = rng.random((gh, gw))
        ys = np.linspace(0, gh - 1.001, h)
        xs = np.linspace(0, gw - 1.001, w)
        y0, x0 = ys.astype(int), xs.astype(int)
        fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
        a = g[y0][:, x0] * (1 - fy) * (1 - fx) + g[y0 + 1][:, x0] * fy * (1 - fx) \
            + g[y0][:, x0 + 1] * (1 - fy) * fx + g[y0 + 1][:, x0 + 1] * fy * fx
        acc += a / (o + 1)
    acc -= acc.min()
    acc /= max(acc.max(), 1e-9)
    return (acc * 253).astype(np.uint8)


def adversarial_cases():
    """(name, image, seeds) -- the edge cases SURVEY 8c lists."""
    out = []
    rng = np.random.default_rng(7)
    # plateau: every pixel equal -> no strict maxima; flood from hand-placed seeds takes many rings
    plateau = np.full((24, 31), 9, dtype=np.uint8)
    out.append(("plateau_one_seed", plateau, [(12, 15)]))
    out.append(("plateau_two_seeds", plateau, [(3, 4), (20, 27)]))
    out.append(("plateau_no_seeds", plateau, []))
    # seeds on the border and corners: coloured but never grown into; still act as neighbours
    img = rng.integers(0, 254, (20, 20), dtype=np.uint8)
    out.append(("border_seeds", img, [(0, 0), (0, 7), (19, 19), (10, 0), (5, 19), (10, 10)]))
    # duplicate seeds: later entry overwrites the colour (lib.rs:1675-1677)
    out.append(("duplicate_seeds", img, [(5, 5), (9, 12), (5, 5), (9, 12), (15, 3)]))
    # adjacent seeds
    out.append(("adjacent_seeds", img, [(5, 5), (5, 6), (6, 5), (12, 12), (13, 12)]))
    # NEVER_FILL walls split the image; ALWAYS_FILL floor
    walls = rng.integers(0, 254, (32, 40), dtype=np.uint8)
    walls[:, 13] = 255
    walls[17, :] = 255
    walls[17, 20] = 0
    walls[5:9, 20:30] = 0
    out.append(("never_fill_walls", walls, [(4, 4), (25, 6), (8, 30), (28, 33), (2, 25)]))
    # degenerate shapes: no interior
    out.append(("h2", rng.integers(0, 254, (2, 9), dtype=np.uint8), [(0, 3), (1, 5)]))
    out.append(("w1", rng.integers(0, 254, (9, 1), dtype=np.uint8), [(4, 0)]))
    out.append(("one_px", np.array([[3]], dtype=np.uint8), [(0, 0)]))
    out.append(("3x3", np.array([[1, 2, 3], [4, 0, 6], [7, 8, 9]], dtype=np.uint8), [(0, 1)]))
    # ragged / odd sizes around tile boundaries
    for (h, w) in ((63, 65), (64, 64), (65, 129), (130, 67), (7, 200), (200, 5)):
        f = rng.integers(0, 254, (h, w), dtype=np.uint8)
        out.append((f"odd_{h}x{w}", f, None))
    # a spiral corridor: one long path, arrival rings in the hundreds
    sp = np.full((33, 33), 255, dtype=np.uint8)
    r, c, dr, dc = 1, 1, 0, 1
    seen = set()
    for _ in range(33 * 33):
        sp[r, c] = 7
        seen.add((r, c))
        nr, nc = r + dr, c + dc
        ahead2 = (nr + dr, nc + dc)
        if not (1 <= nr < 32 and 1 <= nc < 32) or (nr, nc) in seen or ahead2 in seen:
            dr, dc = dc, -dr
            nr, nc = r + dr, c + dc
            if not (1 <= nr < 32 and 1 <= nc < 32) or (nr, nc) in seen or (nr + dr, nc + dc) in seen:
                break
        r, c = nr, nc
    out.append(("spiral", sp, [(1, 1)]))
    # monotone ramp: every level opens one more column
    ramp = np.tile(np.arange(50, dtype=np.uint8) * 5, (20, 1))
    out.append(("ramp", ramp, [(10, 1), (3, 40)]))
    return out


def seeds_or_maxima(img, seeds):
    if seeds is None:
        return [tuple(int(v) for v in rc) for rc in ol.find_local_minima(img)]
    return list(seeds)
