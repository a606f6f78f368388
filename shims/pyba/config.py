"""`from pyba.config import df3d_bones, df3d_colors` (reference df3d/core.py:110,311): plotting constants of the
38-joint fly skeleton -- five-joint chains for the six legs, the antennae, the three stripe points per side; one RGB
colour per joint (limb-wise)."""

_SIDE_BONES = [[0, 1], [1, 2], [2, 3], [3, 4], [5, 6], [6, 7], [7, 8], [8, 9], [10, 11], [11, 12], [12, 13], [13, 14], [16, 17], [17, 18]]
df3d_bones = _SIDE_BONES + [[a + 19, b + 19] for a, b in _SIDE_BONES]

_LIMB_COLORS = [(255, 0, 0), (0, 0, 255), (0, 255, 0), (150, 200, 200), (255, 165, 0)]  # 3 legs, antenna, stripes
_SIDE_COLORS = [_LIMB_COLORS[0]] * 5 + [_LIMB_COLORS[1]] * 5 + [_LIMB_COLORS[2]] * 5 + [_LIMB_COLORS[3]] + [_LIMB_COLORS[4]] * 3
df3d_colors = _SIDE_COLORS + [tuple(int(0.6 * c) for c in col) for col in _SIDE_COLORS]
