"""Offset tables of the affinity stencil.

Same names and results as the reference's scripts_cvppp/utils/affinity_ours.py:4-15
(gen_offsets / multi_offset) and the 3D tables: the norm5 shift list
(scripts_ac3ac4/loss/loss_embedding_mse.py:176, offsets at scripts_ac3ac4/inference.py:190-193).
"""

NORM5_SHIFTS = (1, 1, 1, 2, 3, 3, 3, 9, 9, 4, 27, 27)


def gen_offsets(shift, neighbor=4):
    assert neighbor == 4 or neighbor == 8, 'neigbor must be 4 or 8!'
    axis = [[-shift, 0], [0, -shift]]
    diag = [[-shift, -shift], [-shift, shift]]
    return axis if neighbor == 4 else axis + diag


def multi_offset(shifts, neighbor=4):
    out = []
    for s in shifts:
        out.extend(gen_offsets(s, neighbor=neighbor))
    return out


def axis_offsets_3d(shifts):
    """channel i looks `shifts[i]` voxels back along axis i % 3 of (z, y, x)."""
    out = []
    for i, s in enumerate(shifts):
        o = [0, 0, 0]
        o[i % 3] = -int(s)
        out.append(o)
    return out
