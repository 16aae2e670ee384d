"""The sub-boxes of the splined assembly (solver_kernels.hpp SPL_MAXSUB / SPL_SUB_MAX; assemble_splined_kernel):
the tiling's invariants, restated in Python and checked for every box a 31 x 24 grid can hold. CPU only: the
kernel's own results are held to JtJ by tests/test_solver_parity.py::test_normal_equations_splined_closeups"""
import re, os
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HPP  = open(os.path.join(HERE, "..", "mrcal_amd", "csrc", "solver_kernels.hpp")).read()
HIP  = open(os.path.join(HERE, "..", "mrcal_amd", "csrc", "assembly_splined.hip")).read()
SPL_MAXSUB  = int(re.search(r"#define SPL_MAXSUB\s+(\d+)",  HPP).group(1))
SPL_SUB_MAX = int(re.search(r"#define SPL_SUB_MAX\s+(\d+)", HPP).group(1))
SPL_TW      = int(re.search(r"#define SPL_TW\s+(\d+)",      HIP).group(1))
SPL_NEXTRA  = 12 + 6 + 1


def tiling(owx, owy, order):
    """what assemble_splined_kernel does with a box of owx x owy control points: (nsx, nsy, T), the sub-boxes
    [(x0, y0, w, h)] relative to the box, and the owner of a patch starting at (px, py)"""
    T = SPL_SUB_MAX - order
    nsx = nsy = 1
    if owx*owy + SPL_NEXTRA > SPL_TW:
        nsx = max(1, (owx - order + T - 1)//T)
        nsy = max(1, (owy - order + T - 1)//T)
    nsub = nsx*nsy
    boxes = []
    for isub in range(nsub):
        gx, gy = isub % nsx, isub // nsx
        x0, y0 = (gx*T, gy*T) if nsub > 1 else (0, 0)
        w = min(T + order, owx - x0) if nsub > 1 else owx
        h = min(T + order, owy - y0) if nsub > 1 else owy
        boxes.append((x0, y0, w, h))
    owner = lambda px, py: (py//T)*nsx + (px//T) if nsub > 1 else 0
    return nsx, nsy, T, boxes, owner


@pytest.mark.parametrize("order", (2, 3))
def test_every_patch_lies_whole_in_the_sub_box_that_owns_it(order):
    n1 = order + 1
    biggest = 0
    for owx in range(n1, 32):
        for owy in range(n1, 25):
            nsx, nsy, T, boxes, owner = tiling(owx, owy, order)
            if nsx*nsy > SPL_MAXSUB: continue          # (row by row, with atomics: the documented remainder)
            biggest = max(biggest, owx*owy)
            for (x0, y0, w, h) in boxes:
                assert w >= n1 and h >= n1
                assert w*h + SPL_NEXTRA <= SPL_TW, (owx, owy, w, h)      # fits the local tile
            # every patch that can occur in the box
            for px in range(owx - order):
                for py in range(owy - order):
                    i = owner(px, py)
                    assert 0 <= i < len(boxes)
                    x0, y0, w, h = boxes[i]
                    assert x0 <= px and px + order < x0 + w and y0 <= py and py + order < y0 + h, (owx, owy, px, py, boxes[i])
    # the whole of a 30 x 20 grid is served, and a good deal more
    nsx, nsy, *_ = tiling(30, 20, order)
    assert nsx*nsy <= SPL_MAXSUB
    assert biggest >= 30*20


def test_gather_window_holds_every_pair_of_control_points_that_can_meet():
    """assemble_splined_gather_kernel keeps a control point's row of A as its last 2 (order Nx + order) + 1 columns:
    two control points of one surface meet in a Gram only inside some corner's (order+1)^2 patch"""
    for order, Nx, Ny in ((3, 30, 20), (2, 24, 18), (3, 11, 8)):
        window = 2*(order*Nx + order)
        worst = 0
        for ix in range(Nx):
            for iy in range(Ny):
                r = 2*(iy*Nx + ix)
                for dx in range(-order, order + 1):
                    for dy in range(-order, 1):
                        jx, jy = ix + dx, iy + dy
                        if not (0 <= jx < Nx and 0 <= jy < Ny): continue
                        c = 2*(jy*Nx + jx)
                        if c <= r: worst = max(worst, r - c)
        assert worst <= window, (order, Nx, worst, window)


def test_reciprocal_division_of_local_columns():
    """spl_magic(): c // wx == (c * ceil(2^16 / wx)) >> 16 for the local columns of any box the tile can hold"""
    for wx in range(1, 129):
        magic = (65536 + wx - 1)//wx
        c = np.arange(0, 256)
        assert np.array_equal((c*magic) >> 16, c//wx), wx
