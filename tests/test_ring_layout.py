"""Host-side restatement of the lane <-> cell maps of the LDS-ring backward kernel for K < 16 (csrc/vmp_svae_ring.h, round 5):
the tile's RPT = 64 // K data rows are laid out as rows 0..3 in columns 0..K-1 of the four 16-lane DPP rows plus the remaining
rows in the 16 - K spare lanes.  The kernel's correctness against the oracle is covered on the GPU (tests/test_svae_gpu.py,
tests/test_fullsize_gpu.py); this file pins the arithmetic of the maps themselves - every tile cell is held by exactly one lane,
the pull / rotation sources are the lanes the reductions assume - for every K the launcher accepts, without a GPU.
Replaces nothing in the reference (TF's autodiff of models/svae.py:14-119 has no such layout); it documents ours."""
import pytest


def linear_map(K):
    """run-time-K form (KS = 0): lane -> tile cell index or -1."""
    RPT, E, XR = 64 // K, 16 - K, 64 // K - 4
    out = []
    for lane in range(64):
        d, c = lane >> 4, lane & 15
        if c < K:
            out.append(d * K + c)
        else:
            x = d * E + (c - K)
            out.append(4 * K + x if x < XR * K else -1)
    return out


def rotation_map(K):
    """compile-time-K form (SvRingRot<K>): lane -> tile cell index or -1, and the layout constants."""
    E, XR = 16 - K, 64 // K - 4
    CPR = 4 // XR
    U = (K + CPR - 1) // CPR
    assert U <= E
    out = []
    for lane in range(64):
        d, c = lane >> 4, lane & 15
        if c < K:
            out.append(d * K + c)
        else:
            j = c - K
            comp = (d % CPR) * U + j
            out.append((4 + d // CPR) * K + comp if (j < U and comp < K) else -1)
    return out, E, CPR, U


@pytest.mark.parametrize('K', range(8, 16))
def test_linear_spare_lane_map_is_a_bijection_and_pull_sources_match(K):
    RPT, E, XR = 64 // K, 16 - K, 64 // K - 4
    m = linear_map(K)
    cells = sorted(c for c in m if c >= 0)
    assert cells == list(range(RPT * K))                       # every cell of the tile on exactly one lane
    for d in range(4):                                         # pull: main lane (d, col) reads the lane that holds cell (4 + d, col)
        for col in range(K):
            if d < XR:
                xq = d * K + col
                src = (xq // E) * 16 + K + xq % E
                assert m[src] == (4 + d) * K + col
    # a spare lane of tile row r >= 4 finds its row sum in DPP row r - 4
    for lane, c in enumerate(m):
        if c >= 4 * K:
            assert 0 <= c // K - 4 < 4


@pytest.mark.parametrize('K', [8, 10, 11, 12])
def test_rotation_spare_lane_map_lands_on_the_components_main_lanes(K):
    m, E, CPR, U = rotation_map(K)
    RPT = 64 // K
    assert sorted(c for c in m if c >= 0) == list(range(RPT * K))
    for lane, c in enumerate(m):
        d, col = lane >> 4, lane & 15
        if col >= K and c >= 0:
            chunk = d % CPR
            n = E + chunk * U                                  # row_ror:n moves the value n columns to the right inside the DPP row
            assert 1 <= n <= 15
            assert (col + n) % 16 == c % K                     # ... onto the main lane of the cell's component
            assert c // K == 4 + d // CPR                      # rows_total(): the CPR DPP rows d // CPR * CPR .. share this tile row
    # the launcher gives K = 10 (L = 8) this form; 4 // XR must be integral for it to exist at all
    assert (64 // K - 4) in (1, 2, 4)


def test_no_rotation_layout_for_k9_and_no_spare_rows_from_k13():
    assert 64 // 9 - 4 == 3                                    # three extra rows do not divide the four DPP rows: K = 9 stays on the pull form
    for K in (13, 14, 15):
        assert 64 // K == 4 and all(c < 4 * K for c in linear_map(K))
