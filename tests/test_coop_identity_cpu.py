"""The identity the small-launch encoder (redux_amd/csrc/redux_coop.hpp) rests on, replayed on the CPU with plain integers:
what AdaptiveTreeModel::get_frequency_range returns for symbol i (adaptive_tree.rs:63-92) depends only on the counts of the
symbols before i (capped at the freeze point, adaptive_tree.rs:84), so 64 lanes that each start from the counts of
everything before their segment produce, together, exactly the (low, high) sequence of the serial model.  Also the
in-place Fenwick construction k_coop_model uses (node i covers symbols i - lowbit(i) .. i - 1)."""
import numpy as np
import pytest


def serial_pairs(data, nfreeze):
    """(low, high, total) per symbol of a model that starts with every frequency at 1 (256 symbols + EOF) and counts the
    first nfreeze symbols."""
    freq = np.ones(257, dtype=np.int64)
    out = []
    for i, s in enumerate(data):
        cum = np.concatenate(([0], np.cumsum(freq)))
        out.append((int(cum[s]), int(cum[s + 1]), int(cum[257])))
        if i < nfreeze:
            freq[s] += 1
    return out


def coop_pairs(data, nfreeze, lanes=64):
    n = len(data)
    seg = (n + lanes - 1) // lanes
    out = [None] * n
    # per-segment histograms of the symbols that are still counted, then an exclusive scan over the segments
    hist = np.zeros((lanes, 256), dtype=np.int64)
    for j in range(lanes):
        for i in range(min(j * seg, n), min((j + 1) * seg, n)):
            if i < nfreeze:
                hist[j, data[i]] += 1
    start = np.cumsum(hist, axis=0) - hist
    for j in range(lanes):
        b0, b1 = min(j * seg, n), min((j + 1) * seg, n)
        # Fenwick increments in place, as the kernel builds them: a[i] = count of symbol i - 1, then a[i + lowbit(i)] += a[i]
        a = np.zeros(257, dtype=np.int64)
        a[1:257] = start[j]
        for i in range(1, 256):
            k = i + (i & -i)
            if k < 256:
                a[k] += a[i]
        d = a  # d[e], e = 1 .. 255: increments of node e; node 256 is derived from the number of updates

        def cum(s, nup):  # sum of the frequencies of symbols < s: s (the initial ones) + the increments on s's root path
            if s == 257:
                return 257 + nup
            if s == 256:
                return 256 + nup
            t, e = s, s
            while e > 0:
                t += d[e]
                e -= e & -e
            return t

        for i in range(b0, b1):
            s = int(data[i])
            nup = min(i, nfreeze)
            out[i] = (int(cum(s, nup)), int(cum(s + 1, nup)), 257 + nup)
            if i < nfreeze:
                e = s + 1
                while e < 256:
                    d[e] += 1
                    e += e & -e
    return out


@pytest.mark.parametrize("n,nfreeze", [(1, 10), (63, 1000), (64, 1000), (65, 1000), (1000, 10 ** 9), (1500, 700), (4096, 1300)])
def test_segments_started_from_prefix_counts_reproduce_the_serial_model(n, nfreeze):
    rng = np.random.default_rng(n * 7 + nfreeze % 13)
    data = (rng.integers(0, 256, n) >> rng.integers(0, 6)).astype(np.int64)
    data[rng.integers(0, n, max(1, n // 50))] = 255  # the symbol whose upper end is the derived node 256
    assert coop_pairs(data, nfreeze) == serial_pairs(data, nfreeze)


def test_wave_decoder_table_search_and_update():
    """k_decode_wave's model (redux_amd/csrc/redux_decode_wave.hpp) replayed with numpy: four planes of inclusive prefix
    sums across 64 lanes, the plane picked by three compares against the plane tops, the lane by counting the entries not
    above the code value, cum(s) / cum(s + 1) read from the neighbouring lanes, the update as +1 on every entry above s."""
    rng = np.random.default_rng(11)
    lane = np.arange(64)
    C = [lane + 1 + 64 * q for q in range(4)]  # cum(l + 64 q + 1), all frequencies 1
    T = [64, 128, 192]
    freq = np.ones(256, dtype=np.int64)
    for step in range(3000):
        total_data = int(freq.sum())  # = cum(256) = count - 1
        v = int(rng.integers(0, total_data))  # a code value below the EOF range
        pq = (v >= T[0]) + (v >= T[1]) + (v >= T[2])
        X = C[pq]
        b = int((X <= v).sum())
        assert b < 64
        s = 64 * pq + b
        hi = int(X[b])
        lo = int(X[b - 1]) if b else (0 if pq == 0 else T[pq - 1])
        cum = np.concatenate(([0], np.cumsum(freq)))
        assert (lo, hi) == (int(cum[s]), int(cum[s + 1])) and lo <= v < hi, (step, v, s)
        if step < 2500:  # (then frozen)
            freq[s] += 1
            for q in range(4):
                C[q] = C[q] + (lane + 64 * q >= s)
            T = [T[0] + (s < 64), T[1] + (s < 128), T[2] + (s < 192)]
        for q in range(3):
            assert T[q] == int(C[q][63])
