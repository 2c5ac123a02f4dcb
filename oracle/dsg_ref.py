"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference's object-connectivity step.

PINNING STATUS: **pinned.**  ``add_object_connectivity`` and its predicates (``src/hydra_gnn/preprocess_dsgs.py:89-225``) are
importable in the build container; ``tests/golden/make_dsg_fixture.py`` ran them on the reference's own test graph and
committed the result (``tests/golden/dsg_x8F5xyUWy9e_expected.npz``: 62 kept objects, 178 object edges).
``tests/test_dsg_reader.py`` checks this restatement against that file edge for edge, in order.

Only ``tests/`` may import this module.
"""
import numpy as np


def _is_on(p1, p2, s1, s2, max_on):  # preprocess_dsgs.py:89-110
    xy_dist = np.abs(p1[0:2] - p2[0:2])
    z_dist = np.abs(p1[2] - p2[2])
    n1_above_n2 = p1[2] > p2[2]
    new_thresh = max_on + (s1[2] + s2[2]) / 2
    if all(xy_dist <= s2[0:2] / 2) and n1_above_n2 and z_dist <= new_thresh:
        return True
    if all(xy_dist <= s1[0:2] / 2) and not n1_above_n2 and z_dist <= new_thresh:
        return True
    return False


def _is_under(p1, p2, s1, s2):  # :139-158
    xy_dist = np.abs(p1[0:2] - p2[0:2])
    if all(xy_dist <= s1[0:2] / 2) or all(xy_dist <= s2[0:2] / 2):
        return bool(p1[2] < p2[2] or p2[2] < p1[2])
    return False


def _is_near(p1, p2, s1, s2, threshold_near, max_near):  # :161-180
    avg_size = (s1 + s2) / 2.0
    dist = np.abs(p1 - p2)
    return bool(all(dist <= avg_size * threshold_near) and all(dist - avg_size <= max_near * np.ones(3)))


def object_edges(pos, size, room, threshold_near=2.0, max_near=2.0, max_on=0.2):
    """:191-225 -- objects in visiting order; every object against the earlier objects of its room; int64 [2, E] (i, j)."""
    pos, size = np.asarray(pos, dtype=np.float64), np.asarray(size, dtype=np.float64)
    by_room, out = {}, []
    for i in range(len(room)):
        r = int(room[i])
        if r < 0:
            continue
        for j in by_room.setdefault(r, []):
            if (_is_on(pos[i], pos[j], size[i], size[j], max_on) or _is_under(pos[i], pos[j], size[i], size[j])
                    or _is_near(pos[i], pos[j], size[i], size[j], threshold_near, max_near)):
                out.append((i, j))
        by_room[r].append(i)
    return np.array(out, dtype=np.int64).reshape(-1, 2).T
