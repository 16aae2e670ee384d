"""lchol_plan() - the one function by which the host sizes the grids of the launch-per-panel Cholesky's launches and every
kernel of it finds its role (cholesky_large.hip) - checked on the CPU through its dev export: no GPU needed.
Round 5 gave it `own` (the nested-dissection chains factor their own panels and stop in front of their border) and the
host sizes the dissection's launches for the LARGEST plan it provides for, trusting that no smaller plan needs more
workgroups in any launch; both are held here."""
import ctypes as C
import os
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB = 64


@pytest.fixture(scope="module")
def plan():
    lib = C.CDLL(os.path.join(ROOT, "mrcal_amd", "libmrcal_amd.so"))
    f = lib.mrcal_amd_debug_lchol_plan
    f.restype, f.argtypes = None, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    names = ("npanels", "has_next", "incl00", "ntiles", "ntrsm", "nchain", "ntile", "nblocks", "pprev", "npad")
    def call(n, l, own=-1, with_inverse=True):
        out = (C.c_int*10)()
        f(n, l, 1 if with_inverse else 0, own, out)
        return dict(zip(names, [int(v) for v in out]))
    return call


def blocks_below(n, p):
    """64-row blocks below panel p of an n x n matrix with its rhs row, as the trailing update sees them (without the rhs
    row) and as the panel solve does (with it)"""
    m0 = min(n, (p + 1)*NB)
    return (n - m0 + NB - 1)//NB, (n + 1 - m0 + NB - 1)//NB


@pytest.mark.parametrize("n", (1, 63, 64, 65, 200, 652, 1206, 4096))
def test_the_whole_matrix(plan, n):
    """own = -1: every panel is factored. Over the launches 0 .. npanels: every trailing tile of every panel once, the
    next diagonal block by workgroup 0 instead of a tile; every panel's rows below solved once; every (p, q < p) block of
    L^-1 by one chain workgroup, every (p, q, k), q <= k <= p - 2, by one tile workgroup; nothing behind the closing launch"""
    P = (n + NB - 1)//NB
    assert plan(n, 0)["npanels"] == P and plan(n, 0)["npad"] == NB*P
    tiles = trsm = chain = ytile = 0
    for l in range(P + 1):
        q = plan(n, l)
        assert q["nblocks"] == (1 + q["ntiles"] + q["ntrsm"] + q["nchain"] + q["ntile"] if (q["has_next"] or q["ntiles"] + q["ntrsm"] + q["nchain"] + q["ntile"]) else 0)
        assert q["incl00"] == 0                       # (a panel with rows of the matrix below it has a next one)
        tiles += q["ntiles"]; trsm += q["ntrsm"]; chain += q["nchain"]; ytile += q["ntile"]
        if l < P:
            nbt, _ = blocks_below(n, l)
            assert q["has_next"] == (1 if l + 1 < P else 0)
            assert q["ntiles"] == (nbt*(nbt + 1)//2 - 1 + nbt if nbt > 0 else 0)
    assert plan(n, P + 1)["nblocks"] == 0 and plan(n, P + 5)["nblocks"] == 0
    assert trsm == sum(blocks_below(n, p)[1] for p in range(P))
    assert chain == P*(P - 1)//2
    assert ytile == sum(k - q + 1 >= 1 for p in range(P) for q in range(P) for k in range(q, p - 1))


@pytest.mark.parametrize("own,border", ((1, 1), (1, 226), (3, 40), (4, 226), (4, 290), (16, 1000)))
def test_a_chain_that_stops_in_front_of_its_border(plan, own, border):
    """own panels of a matrix of 64 own + border: launches 0 .. own. The last own panel has no next block of its own but a
    border: the tile behind it is a tile like the others (incl00); the panel solves reach through the border and the rhs
    row; L^-1 is made for the own panels alone, in a workspace sized by them"""
    n = NB*own + border
    tiles = trsm = chain = ytile = 0
    for l in range(own + 1):
        q = plan(n, l, own)
        assert q["npanels"] == own and q["npad"] == NB*own
        tiles += q["ntiles"]; trsm += q["ntrsm"]; chain += q["nchain"]; ytile += q["ntile"]
        if l < own:
            nbt, _ = blocks_below(n, l)
            last = l + 1 == own
            assert q["has_next"] == (0 if last else 1) and q["incl00"] == (1 if last else 0)
            assert q["ntiles"] == nbt*(nbt + 1)//2 - (0 if last else 1) + nbt
    assert plan(n, own + 1, own)["nblocks"] == 0
    assert trsm == sum(blocks_below(n, p)[1] for p in range(own))
    assert chain == own*(own - 1)//2
    assert ytile == sum(1 for p in range(own) for q in range(own) for k in range(q, p - 1))
    # every tile of the border is updated by every own panel, and the own panels' trailing tiles as in a matrix of their own
    nb_border = (border + NB - 1)//NB
    expect = 0
    for p in range(own):
        nbt = (own - 1 - p) + nb_border
        expect += nbt*(nbt + 1)//2 - (1 if p + 1 < own else 0) + nbt
    assert tiles == expect


def test_the_host_sizes_the_dissections_launches_for_the_largest_plan(plan):
    """launch_cholesky_large(nds): the grid of round l is sized by the plan of (64 R + ns_max, own = R); a side of own' <= R
    panels with a border of nS <= ns_max must not need more workgroups in ANY launch - its panels' launches, its closing
    launch at l = own' (where the grid was sized for a panel's launch), nothing behind it"""
    for R, ns_max in ((1, 64), (3, 286), (4, 290), (6, 500)):
        for l in range(R + 1):
            host = plan(NB*R + ns_max, l, R)["nblocks"]
            for own in range(1, R + 1):
                for nS in (1, 63, 64, 65, ns_max - 64, ns_max - 1, ns_max):
                    if nS < 1 or nS > ns_max: continue
                    q = plan(NB*own + nS, l, own)
                    assert q["nblocks"] <= host, (R, ns_max, l, own, nS, q, host)
