"""CPU: known-answer tests of the stickman oracle (oracle/stickman_oracle.c), hand-derived from the published
OpenCV 4.1.2 line / fillPoly algorithms (cv2 itself is absent: parity against OpenCV is unpinned)."""
import numpy as np

from oracle import stickman as S


def _line(a, b, h=16, w=16, color=255):
    kps = np.array([[a, b]], dtype=np.float32)
    return S.raster(kps, [], [(1, 0, 1, 0, color)], h, w)[0, 0]


def _pts(img):
    ys, xs = np.nonzero(img)
    return sorted(zip(xs.tolist(), ys.tolist()))


def test_axis_aligned_and_diagonal_lines():
    assert _pts(_line((2, 3), (6, 3))) == [(x, 3) for x in range(2, 7)]
    assert _pts(_line((5, 1), (5, 4))) == [(5, y) for y in range(1, 5)]
    assert _pts(_line((0, 0), (4, 4))) == [(i, i) for i in range(5)]
    assert _pts(_line((4, 0), (0, 4))) == sorted((4 - i, i) for i in range(5))
    assert _pts(_line((3, 3), (3, 3))) == [(3, 3)]


def test_bresenham_steps_and_direction_independence():
    # dx 4, dy 2: err = dx - 2dy = 0 -> (0,0) (1,0) (2,1) (3,1) (4,2)   (left-to-right iterator)
    want = [(0, 0), (1, 0), (2, 1), (3, 1), (4, 2)]
    assert _pts(_line((0, 0), (4, 2))) == want
    assert _pts(_line((4, 2), (0, 0))) == want          # the iterator always starts from the left end point
    # steep: dy 4, dx 2 -> x advances on the masked steps
    assert _pts(_line((0, 0), (2, 4))) == [(0, 0), (0, 1), (1, 2), (1, 3), (2, 4)]


def test_lines_are_clipped_to_the_image():
    assert _pts(_line((12, 5), (40, 5))) == [(x, 5) for x in range(12, 16)]
    assert _pts(_line((20, 20), (30, 30))) == []        # completely outside
    img = _line((0, 8), (31, 8), w=16)
    assert _pts(img) == [(x, 8) for x in range(16)]


def test_invalid_joints_draw_nothing_and_coordinates_truncate():
    assert _pts(_line((-1, 3), (6, 3))) == []            # any negative coordinate invalidates the line
    assert _pts(_line((2.9, 3.7), (6.2, 3.1))) == [(x, 3) for x in range(2, 7)]   # np.int_ truncation


def _poly(vs, h=16, w=16):
    kps = np.array([vs], dtype=np.float32)
    return S.raster(kps, list(range(len(vs))), [(0, 0, 0, 2, 255)], h, w)[0, 2]


def test_fill_rectangle_and_triangle():
    img = _poly([(2, 2), (6, 2), (6, 5), (2, 5)])
    assert _pts(img) == sorted((x, y) for y in range(2, 6) for x in range(2, 7))
    tri = _poly([(0, 0), (8, 0), (0, 8)])
    assert _pts(tri) == sorted((x, y) for y in range(9) for x in range(0, 9 - y))
    assert int((tri > 0).sum()) == 45


def test_polygon_needs_three_valid_points_and_clips():
    assert _pts(_poly([(2, 2), (6, 2), (-1, 5)])) == []      # only two valid vertices -> nothing (lib/utils.py:348)
    img = _poly([(10, 10), (30, 10), (30, 30), (10, 30)])    # extends past the 16x16 frame
    assert _pts(img) == sorted((x, y) for y in range(10, 16) for x in range(10, 16))


def test_draw_order_is_overwrite_order():
    kps = np.array([[(1, 4), (9, 4), (5, 1), (5, 8)]], dtype=np.float32)
    cmds = [(1, 0, 1, 0, 255), (1, 2, 3, 0, 127)]          # horizontal 255, then vertical 127 on the same plane
    img = S.raster(kps, [], cmds, 12, 12)[0, 0]
    assert img[4, 5] == 127 and img[4, 4] == 255 and img[2, 5] == 127


def test_h36m_draw_list_planes():
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from behavior_driven_video_synthesis_amd.lib.utils import H36M_JOINT_MODEL, stickman_draw_list
    cmds = stickman_draw_list(H36M_JOINT_MODEL)
    assert len(cmds) == 3 + 5 + 5 + 4
    rng = np.random.default_rng(0)
    kps = (rng.normal(32, 12, size=(3, 17, 2))).astype(np.float32)
    out = S.raster(kps, H36M_JOINT_MODEL.body, cmds, 64, 64)
    assert set(np.unique(out[:, 2])) <= {0, 255}            # plane 2: body polygon only
    assert set(np.unique(out[:, 0])) <= {0, 127, 255}       # plane 0: left limbs 255, head 127 (body colour 0)
    assert set(np.unique(out[:, 1])) <= {0, 127, 255}
