"""The identities the cell decoder (redux_amd/csrc/redux_decode_cells.hpp) rests on, replayed on the CPU with plain integers
against the Python restatement of AdaptiveTreeModel (adaptive_tree.rs:36-136):

  * the Fenwick tree cut into cells of four levels -- group g holds the nodes (16 c + j) << 4g, j = 1 .. 15 -- with a PARTIAL
    topmost cell when the symbol width is not a multiple of four;
  * get_symbol as a carry-form descent over full node values: q = ~rem (mod 2^32), q2 = q + t, "went right" = the top bit of
    q2, the new q = max(q, q2), cum(s + 1) - v - 1 = min over the levels of q2, seeded with the virtual root probe against
    tree[2^bits] = count - 1 (whose top bit is the EOF test);
  * update(s + 1) as the write-back of the HALF of each cell that the path lies in: a level's node is incremented exactly
    where the descent went left, and having gone left at a cell's top level puts the rest of the path into the top node's half.
"""
import random

import pytest

from oracle import redux_ref as rr

M32 = 0xFFFFFFFF


class Cells:
    """The kernel's data structure: per group a list of cells, a cell = 16 slots (slot j = node j, slot 0 unused)."""

    def __init__(self, sb):
        self.sb = sb
        self.groups = (sb + 3) // 4
        self.top_levels = sb - 4 * (self.groups - 1)
        self.cell = []
        for g in range(self.groups):
            n = 1 if g == self.groups - 1 else 1 << (sb - 4 * (g + 1))
            # every node = its lowbit (all frequencies 1): (lowbit of j) << 4g
            self.cell.append([[0] + [(j & -j) << (4 * g) for j in range(1, 16)] for _ in range(n)])

    def get_symbol(self, v, count, update):
        """(symbol or None for EOF, lo, hi) for the code value v; the model is updated when `update`."""
        q = (~v) & M32
        hq = (q + count - 1) & M32
        if hq >> 31:  # v >= count - 1: the EOF symbol (adaptive_tree.rs:116)
            return None, count - 1, count
        bits = 0
        for g in range(self.groups - 1, -1, -1):
            levels = self.top_levels if g == self.groups - 1 else 4
            c = self.cell[g][0 if g == self.groups - 1 else bits]
            went = []  # went right at in-cell level 3, 2, 1, 0 (absent levels: left)
            j = 0      # in-cell prefix of the path
            for lv in range(levels - 1, -1, -1):
                node = j | (1 << lv)
                q2 = (q + c[node]) & M32
                right = q2 >> 31
                bits = (bits << 1) | right
                q = max(q, q2)
                hq = min(hq, q2)
                went.append((lv, node, right))
                if right:
                    j = node
            if update:
                # what the kernel adds: +1 on the path's node of every level where the descent went left -- and those nodes all
                # lie in the half (nodes 1..8 / 9..15) the path ends in, which is what makes the update one write-back
                half_right = any(lv == 3 and right for lv, _, right in went)
                for lv, node, right in went:
                    if not right:
                        assert (node >= 9) == half_right or node == 8 and not half_right, (node, half_right)
                        c[node] += 1
        return bits, (v + q + 1) & M32, (v + hq + 1) & M32

    def node(self, e):
        """tree[e] of the reference."""
        tz = (e & -e).bit_length() - 1
        g = tz // 4
        return self.cell[g][e >> (4 * g + 4)][(e >> (4 * g)) & 15]


@pytest.mark.parametrize("sb,fb", [(1, 3), (2, 8), (3, 12), (4, 10), (4, 20), (5, 9), (6, 20), (7, 12), (9, 14), (10, 20), (11, 13), (12, 14), (12, 20)])
def test_cell_descent_and_write_back_equal_the_reference_tree(sb, fb):
    rnd = random.Random(sb * 100 + fb)
    p = rr.Parameters(sb, fb, min(fb + 2, 32) if fb + 2 <= 32 else fb + 2)
    ref = rr.AdaptiveTreeModel(p)
    cells = Cells(sb)
    nfreeze = ((1 << fb) - 1) - ((1 << sb) + 1)
    steps = 6000 if sb <= 7 else 3000
    skew = rnd.choice([1, 2, 4])
    for step in range(steps):
        count = ref.total_frequency()
        assert count == (1 << sb) + 1 + min(step, nfreeze)
        # code values: uniform, skewed towards small symbols, and the boundaries of the table
        r = rnd.random()
        v = rnd.randrange(count - 1) if r < 0.5 else int((count - 1) * rnd.random() ** skew) if r < 0.95 else rnd.choice([0, count - 2])
        v = min(v, count - 2)
        s_ref, lo_ref, hi_ref = ref.get_symbol(v)  # (updates the reference model, unless frozen)
        s, lo, hi = cells.get_symbol(v, count, step < nfreeze)
        assert (s, lo, hi) == (s_ref, lo_ref, hi_ref), (sb, step, v)
    # the EOF probe, and every node of the tree after all those updates
    count = ref.total_frequency()
    assert cells.get_symbol(count - 1, count, False) == (None, count - 1, count)
    for e in range(1, 1 << sb):
        assert cells.node(e) == ref.tree[e], (sb, e)
