"""Host-side tables of a kernel map (csrc/hostprep.hip: tile descriptors of the gathered GEMM, work items of the weight
gradient) against their numpy formulation -- the launch ORDER is part of the contract (it decides which feature rows
are still in L2), so every array must match exactly."""
import numpy as np
import pytest

TILE_ROWS = 128


def _tiles_numpy(k_off_host, skip_k=-1):
    k_off = np.asarray(k_off_host, dtype=np.int64)
    cnt = np.diff(k_off)
    if skip_k >= 0:
        cnt = cnt.copy()
        cnt[skip_k] = 0
    nt = (cnt + TILE_ROWS - 1) // TILE_ROWS
    total = int(nt.sum())
    if total == 0:
        return np.zeros((3, 0), dtype=np.int32), 0
    tile_k = np.repeat(np.arange(len(cnt), dtype=np.int64), nt)
    first = np.repeat(np.cumsum(nt) - nt, nt)
    within = np.arange(total, dtype=np.int64) - first
    row0 = k_off[tile_k] + within * TILE_ROWS
    rows = np.minimum(TILE_ROWS, k_off[tile_k + 1] - row0)
    order = np.argsort((within + 0.5) / nt[tile_k], kind="stable")
    return np.stack([tile_k[order], row0[order], rows[order]]).astype(np.int32), total


def _items_numpy(k_off_host, chunk, mode=2, group=32):
    k_off = np.asarray(k_off_host, dtype=np.int64)
    cnt = np.diff(k_off)
    n_k = (cnt + chunk - 1) // chunk
    total = int(n_k.sum())
    item_k = np.repeat(np.arange(len(cnt), dtype=np.int64), n_k)
    first = np.repeat(np.cumsum(n_k) - n_k, n_k)
    within = np.arange(total, dtype=np.int64) - first
    p0 = k_off[item_k] + within * chunk
    p1 = np.minimum(p0 + chunk, k_off[item_k + 1])
    order = np.arange(total, dtype=np.int64)
    if mode >= 1 and total:
        order = np.argsort((within + 0.5) / n_k[item_k], kind="stable")
        if mode >= 2:
            i = np.arange(total, dtype=np.int64)
            g, j = i // group, i % group
            order = order[np.argsort(8 * ((g // 8) * group + j) + g % 8, kind="stable")]
    return (np.stack([item_k, p0, p1, order]).astype(np.int32), total,
            np.concatenate([[0], np.cumsum(n_k)]).astype(np.int32))


def _rule_books():
    rng = np.random.default_rng(7)
    for K in (1, 8, 27, 125):
        yield K, np.zeros(K + 1, dtype=np.int64).tolist()                   # no pairs at all
        yield K, np.arange(K + 1, dtype=np.int64).tolist()                  # one pair per offset
        yield K, (np.arange(K + 1, dtype=np.int64) * TILE_ROWS).tolist()    # exactly one full tile each (all keys tie)
        for _ in range(12):
            cnt = rng.integers(0, 6000, K) * (rng.random(K) > 0.25)
            yield K, np.concatenate([[0], np.cumsum(cnt)]).tolist()
    # the shape of a stride-1 3^3 map of the bench workload: one dominant centre offset
    cnt = rng.integers(8000, 130000, 27)
    cnt[13] = 352468
    yield 27, np.concatenate([[0], np.cumsum(cnt)]).tolist()


def test_tile_descriptors_match_numpy():
    from lidog_amd import me
    for K, k_off in _rule_books():
        for skip in (-1, K // 2):
            want, n_want = _tiles_numpy(k_off, skip)
            got, n_got = me._tiles_host(k_off, skip)
            assert n_got == n_want and got.shape == want.shape and got.dtype == np.int32
            assert np.array_equal(got, want), (K, skip)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_weight_gradient_items_match_numpy(mode, monkeypatch):
    from lidog_amd import me
    monkeypatch.setattr(me, "_WGRAD_ORDER", mode)
    for K, k_off in _rule_books():
        for chunk in (128, 1024, 4096):
            want = _items_numpy(k_off, chunk, mode, me._WGRAD_GROUP)
            got = me._wgrad_items_host(k_off, chunk)
            assert got[1] == want[1]
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]), (K, chunk)
            if got[1]:   # row 3 is a permutation of the items
                assert np.array_equal(np.sort(got[0][3]), np.arange(got[1]))
