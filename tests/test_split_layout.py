"""CPU checks of the LDS layout claims of split16_dma_fwd_kernel (graphsage_amd/csrc/gs_split16.hip), restated in Python:
the A planes are filled by LDS-DMA (lane-linear destination: wave-uniform base + 16 lane) with the 8-k chunk permutation applied to
the SOURCE address, and read back through the same permutation; every 16-lane group of a ds_read_b128 must touch all 64 banks once
(MI355X_MICROARCH.md, LDS table: groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63}; bank of byte
address a = (a / 4) mod 64)."""
import numpy as np

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
A_PLANE = 128 * 64


def _banks(addr):
    """the four consecutive banks a 16-byte access at byte address `addr` touches"""
    return [((addr // 4) + j) % 64 for j in range(4)]


def test_dma_destination_is_lane_linear_and_source_permutation_is_an_involution():
    # thread tid of the 512: row tid >> 2, position tid & 3; its wave's DMA instruction writes base + 16 * lane
    img = {}
    for tid in range(512):
        wave, lane = tid >> 6, tid & 63
        arow, pos = tid >> 2, tid & 3
        chunk = pos ^ ((arow >> 2) & 3)                       # akc: the 8-k chunk this thread FETCHES
        dst = wave * 1024 + 16 * lane                          # where the hardware puts it
        assert dst == arow * 64 + pos * 16                     # = row-major [128 rows][4 positions] without padding
        img[(arow, pos)] = chunk
    for arow in range(128):
        assert sorted(img[(arow, p)] for p in range(4)) == [0, 1, 2, 3]
        for c in range(4):                                     # reading chunk c of row r at position c ^ swz(r) finds it
            assert img[(arow, c ^ ((arow >> 2) & 3))] == c


def test_a_fragment_reads_are_bank_conflict_free():
    for wm in range(2):
        for i in range(2):
            for q in range(2):
                for p in range(2):
                    for grp in GROUPS:
                        used = []
                        for lane in grp:
                            l31, lh = lane & 31, lane >> 5
                            sw = (l31 >> 2) & 3
                            addr = p * A_PLANE + (64 * wm + 32 * i + l31) * 64 + (((2 * q + lh) ^ sw) * 16)
                            used += _banks(addr)
                        assert len(used) == 64 and len(set(used)) == 64


def test_b_fragment_reads_are_bank_conflict_free():
    A_BYTES = 2 * A_PLANE
    for wn in range(4):
        for j in range(2):
            for q in range(2):
                for p in range(2):
                    for grp in GROUPS:
                        used = []
                        for lane in grp:
                            l31, lh = lane & 31, lane >> 5
                            addr = A_BYTES + (64 * wn + l31) * 16 + lh * 2 * 4096 + (4 * q + p) * 4096 + j * (32 * 16)
                            used += _banks(addr)
                        assert len(set(used)) == 64


def test_b_tile_chunks_land_where_the_fragments_read_them():
    # DMA: chunk c = (k-group of the stage) * 2 + piece of column tid & 255 goes to b_dst + c * 4096 + 16 * lane, b_dst = (wave & 3) * 1024
    for tid in range(512):
        wave, lane = tid >> 6, tid & 63
        col = tid & 255
        for j in range(4):
            c = (wave >> 2) + 2 * j
            dst = (wave & 3) * 1024 + c * 4096 + 16 * lane
            assert dst == c * 4096 + col * 16                  # [8 chunks][256 columns][16 bytes]
    # fragment (q, lh, piece p, column n): k-group 2 q + lh of the stage, chunk (2 q + lh) * 2 + p
    for q in range(2):
        for lh in range(2):
            for p in range(2):
                c = (2 * q + lh) * 2 + p
                assert lh * 2 * 4096 + (4 * q + p) * 4096 == c * 4096


def test_three_slot_ring_never_overwrites_a_slot_that_is_still_read():
    """Stage s is read from slot s % 3 (first half before the barrier of stage s, second-half look-ahead after the barrier of stage
    s - 1); stage s + 2 is requested at the top of stage s into slot (s + 2) % 3 = (s - 1) % 3, whose last reads were retired by the
    lgkmcnt(0) + barrier in the middle of stage s - 1."""
    for s in range(2, 40):
        writes = (s + 2) % 3
        assert writes == (s - 1) % 3
        assert writes != s % 3 and writes != (s + 1) % 3       # neither the stage being read nor the one read next
