// Host-side tables of a kernel map: tile descriptors of the gathered GEMM and work items of the weight gradient.
// Pure host code (no device access): what lidog_amd/me.py used to compute with numpy per map and batch (~3 ms of
// python per training step over the 15 maps and ~25 (map, channels) shapes of MinkUNet34); same values, same order.
#include <algorithm>
#include <numeric>
#include <vector>

#include "common.h"

namespace {
// order[] = stable argsort of key(k, w) = (w + 0.5) / n[k] over the items enumerated offset by offset (k ascending, w
// ascending inside k) -- without a comparison sort over all of them: inside one offset the keys ascend in steps of
// 1 / n[k] >= 1 / B with B = max n[k], so a bucket floor(key * B) holds at most one item per offset; items are dropped
// into their buckets in enumeration order (k ascending) and every bucket (a handful of items) is insertion-sorted by
// key, which keeps equal keys in enumeration order.  O(total + K) instead of O(total log total).
void order_by_position(const std::vector<int64_t> &n, int64_t total, std::vector<int64_t> &order) {
    const int K = (int)n.size();
    int64_t B = 1;
    for (int k = 0; k < K; ++k) B = n[k] > B ? n[k] : B;
    std::vector<double> key(total);
    std::vector<int64_t> bucket(total), start(B + 1, 0);
    int64_t t = 0;
    for (int k = 0; k < K; ++k)
        for (int64_t w = 0; w < n[k]; ++w, ++t) {
            key[t] = ((double)w + 0.5) / (double)n[k];
            int64_t b = (int64_t)(key[t] * (double)B);
            bucket[t] = b < B ? b : B - 1;
            ++start[bucket[t] + 1];
        }
    for (int64_t b = 0; b < B; ++b) start[b + 1] += start[b];
    order.assign(total, 0);
    std::vector<int64_t> fill(start.begin(), start.end() - 1);
    for (t = 0; t < total; ++t) order[fill[bucket[t]]++] = t;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t i = start[b] + 1; i < start[b + 1]; ++i) {
            int64_t v = order[i], j = i;
            while (j > start[b] && key[order[j - 1]] > key[v]) {
                order[j] = order[j - 1];
                --j;
            }
            order[j] = v;
        }
}
}  // namespace

// Tile descriptors (tile_k, tile_row0, tile_rows) of 128-row tiles that never straddle an offset, in launch order:
// tiles at the same relative position of their offset segment run together (pairs are sorted by output row inside a
// segment, so those tiles gather nearly the same feature rows while they are still in L2).
// k_off_host [K+1]; skip_k >= 0: that offset is left out; out [3][cap] int32 -> rows of n_tiles entries are written at
// out, out + n_tiles, out + 2 n_tiles (a contiguous [3][n_tiles] array).  Returns n_tiles, or -1 if cap is too small.
extern "C" int64_t lidog_tiles_host(const int64_t *k_off_host, int32_t K, int32_t skip_k, int32_t tile_rows,
                                    int32_t *out, int64_t cap) {
    std::vector<int64_t> nt(K);
    int64_t total = 0;
    for (int k = 0; k < K; ++k) {
        int64_t cnt = (k == skip_k) ? 0 : k_off_host[k + 1] - k_off_host[k];
        nt[k] = (cnt + tile_rows - 1) / tile_rows;
        total += nt[k];
    }
    if (total > cap) return -1;
    if (total == 0) return 0;
    std::vector<int32_t> tk(total), row0(total), rows(total);
    int64_t t = 0;
    for (int k = 0; k < K; ++k)
        for (int64_t w = 0; w < nt[k]; ++w, ++t) {
            int64_t r0 = k_off_host[k] + w * tile_rows;
            int64_t left = k_off_host[k + 1] - r0;
            tk[t] = k;
            row0[t] = (int32_t)r0;
            rows[t] = (int32_t)(left < tile_rows ? left : tile_rows);
        }
    std::vector<int64_t> order;
    order_by_position(nt, total, order);
    for (int64_t i = 0; i < total; ++i) {
        out[i] = tk[order[i]];
        out[total + i] = row0[order[i]];
        out[2 * total + i] = rows[order[i]];
    }
    return total;
}

// Work items of the weight gradient (include/lidog_amd.h: lidog_sconv_wgrad): the rule book cut into pair ranges of
// `chunk` pairs that never straddle an offset.  items [4][n] = (k, first pair, end pair, launch order): rows 0-2 are
// ordered by k; row 3 says which item workgroup x runs -- order_mode >= 1: items at the same relative position of
// their offsets next to each other, >= 2: additionally in groups of `group` that land on the same XCD (workgroups go to
// the 8 XCDs round-robin).  item_off [K+1] = first item of every offset.  Returns n, or -1 if cap is too small;
// items is written as a contiguous [4][n] array.
extern "C" int64_t lidog_wgrad_items_host(const int64_t *k_off_host, int32_t K, int64_t chunk, int32_t order_mode,
                                          int32_t group, int32_t *items, int32_t *item_off, int64_t cap) {
    if (chunk <= 0 || group <= 0) return -1;
    std::vector<int64_t> nk(K);
    int64_t total = 0;
    item_off[0] = 0;
    for (int k = 0; k < K; ++k) {
        int64_t cnt = k_off_host[k + 1] - k_off_host[k];
        nk[k] = (cnt + chunk - 1) / chunk;
        total += nk[k];
        item_off[k + 1] = (int32_t)total;
    }
    if (total > cap) return -1;
    if (total == 0) return 0;
    int64_t t = 0;
    for (int k = 0; k < K; ++k)
        for (int64_t w = 0; w < nk[k]; ++w, ++t) {
            int64_t p0 = k_off_host[k] + w * chunk;
            int64_t p1 = p0 + chunk < k_off_host[k + 1] ? p0 + chunk : k_off_host[k + 1];
            items[t] = k;
            items[total + t] = (int32_t)p0;
            items[2 * total + t] = (int32_t)p1;
        }
    std::vector<int64_t> order(total);
    std::iota(order.begin(), order.end(), (int64_t)0);
    if (order_mode >= 1) {
        order_by_position(nk, total, order);
        if (order_mode >= 2) {
            // item i of that sequence is launched at position 8 * ((g / 8) * group + j) + g % 8 (g = i / group,
            // j = i % group): positions are distinct, so ordering by them is a scatter followed by a compaction
            const int64_t groups = (total + group - 1) / group;
            const int64_t span = 8 * ((groups + 7) / 8) * group;
            std::vector<int64_t> at(span, -1);
            for (int64_t i = 0; i < total; ++i) {
                int64_t g = i / group, j = i % group;
                at[8 * ((g / 8) * group + j) + g % 8] = order[i];
            }
            int64_t w = 0;
            for (int64_t q = 0; q < span; ++q)
                if (at[q] >= 0) order[w++] = at[q];
        }
    }
    for (int64_t i = 0; i < total; ++i) items[3 * total + i] = (int32_t)order[i];
    return total;
}
