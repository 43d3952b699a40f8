/*
 * oracle/me_oracle.c -- CPU restatement of the MinkowskiEngine v0.5.4 operator
 * semantics that LiDOG's hot path relies on.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product path (lidog_amd/) never links, imports or calls it.
 *
 * PARITY STATUS: "parity unpinned" at the MinkowskiEngine boundary.  The
 * reference (saltoricristiano/lidog) delegates all sparse arithmetic to the
 * third-party pip package MinkowskiEngine==0.5.4 (README.md:29 of the
 * reference), whose source is not vendored and which is not installed here.
 * The reference has no tests and no golden vectors for this path.  What IS
 * pinned (see tests/ and tests/golden/make_golden.py):
 *   - the convolution arithmetic, against torch.nn.functional.conv3d /
 *     conv_transpose3d on densified inputs (dense equivalence);
 *   - gradients, against torch.autograd.gradcheck in fp64 on the python side;
 *   - everything that is the reference's OWN python (model wiring of
 *     utils/models/minkunet_bev.py, sparse2super, Encoder2D, the DICE losses),
 *     by importing that code in the build container on top of this oracle.
 * The ME conventions that nothing in the reference pins (row order of strided
 * maps = first occurrence, kernel offset index x-fastest) follow the published
 * ME 0.5.4 CPU algorithm as summarised in SURVEY.md section 8(b).
 *
 * Semantics restated (reference call sites in utils/models/minkunet_bev.py):
 *   orc_unique_first   ME.SparseTensor(coordinates, features)      trainer_lighting_2d.py:151
 *   orc_stride         coordinate map stride (conv k2 s2)          minkunet_bev.py:62,69,76,83
 *   orc_kernel_map     kernel map for MinkowskiConvolution         minkunet_bev.py:57-123
 *   orc_conv_fwd/...   gather -> GEMM -> scatter-add per offset    (ME CPU algorithm)
 *   orc_sparse_quantize ME.utils.sparse_quantize                   semantickitti_bev.py:232-238
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int64_t cap;      /* power of two */
    int32_t *rows;    /* -1 = empty, else index into coords */
    const int32_t *coords;
} orc_table;

static inline uint64_t orc_hash4(const int32_t *c) {
    uint64_t h = 1469598103934665603ULL;
    for (int d = 0; d < 4; ++d) {
        h ^= (uint32_t)c[d];
        h *= 1099511628211ULL;
        h ^= h >> 29;
    }
    return h;
}

static int orc_table_init(orc_table *t, int64_t n, const int32_t *coords) {
    int64_t cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    t->cap = cap;
    t->coords = coords;
    t->rows = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    if (!t->rows) return -1;
    memset(t->rows, 0xff, sizeof(int32_t) * (size_t)cap);
    return 0;
}

static void orc_table_free(orc_table *t) { free(t->rows); t->rows = NULL; }

/* returns the row stored for key (inserting `row` if absent) */
static inline int32_t orc_table_insert(orc_table *t, const int32_t *key, int32_t row) {
    uint64_t slot = orc_hash4(key) & (uint64_t)(t->cap - 1);
    for (;;) {
        int32_t r = t->rows[slot];
        if (r < 0) { t->rows[slot] = row; return row; }
        if (memcmp(t->coords + 4 * (int64_t)r, key, 16) == 0) return r;
        slot = (slot + 1) & (uint64_t)(t->cap - 1);
    }
}

static inline int32_t orc_table_find(const orc_table *t, const int32_t *key) {
    uint64_t slot = orc_hash4(key) & (uint64_t)(t->cap - 1);
    for (;;) {
        int32_t r = t->rows[slot];
        if (r < 0) return -1;
        if (memcmp(t->coords + 4 * (int64_t)r, key, 16) == 0) return r;
        slot = (slot + 1) & (uint64_t)(t->cap - 1);
    }
}

/* ME.SparseTensor construction: rows with equal (b,x,y,z) collapse onto the
 * first occurrence; output rows keep first-occurrence order.
 * unique_rows[j] = input row that became output row j; inverse[i] = output row
 * of input row i.  Returns the number of output rows (== n when all unique). */
int64_t orc_unique_first(const int32_t *coords, int64_t n, int32_t *unique_rows, int32_t *inverse) {
    orc_table t;
    if (orc_table_init(&t, n, coords)) return -1;
    int32_t *first2out = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int64_t n_out = 0;
    for (int64_t i = 0; i < n; ++i) {
        int32_t r = orc_table_insert(&t, coords + 4 * i, (int32_t)i);
        if (r == (int32_t)i) {
            first2out[i] = (int32_t)n_out;
            unique_rows[n_out++] = (int32_t)i;
        }
        inverse[i] = first2out[r];
    }
    free(first2out);
    orc_table_free(&t);
    return n_out;
}

static inline int32_t orc_floor_to(int32_t c, int32_t s) {
    /* floor(c / s) * s with floor toward -inf (negative LiDAR coordinates) */
    int32_t q = c / s;
    if ((c % s != 0) && ((c < 0) != (s < 0))) --q;
    return q * s;
}

/* Strided coordinate map: out coordinate = floor(c / s) * s per spatial dim
 * (batch column untouched), duplicates collapse, rows in first-occurrence
 * order of the parent map.  parent2child[i] = output row of parent row i. */
int64_t orc_stride(const int32_t *coords, int64_t n, int32_t new_stride, int32_t *out_coords,
                   int32_t *parent2child) {
    orc_table t;
    if (orc_table_init(&t, n, out_coords)) return -1;
    int64_t n_out = 0;
    for (int64_t i = 0; i < n; ++i) {
        int32_t key[4];
        key[0] = coords[4 * i];
        for (int d = 1; d < 4; ++d) key[d] = orc_floor_to(coords[4 * i + d], new_stride);
        /* tentatively append so the table can compare against it */
        memcpy(out_coords + 4 * n_out, key, 16);
        int32_t r = orc_table_insert(&t, key, (int32_t)n_out);
        if (r == (int32_t)n_out) ++n_out;
        parent2child[i] = r;
    }
    orc_table_free(&t);
    return n_out;
}

/* Kernel map in neighbour-table form: nbr[o*K + k] = row of the input map at
 * coordinate out[o] + offsets[k] (same batch), or -1.  The pair list of ME
 * (in_maps[k], out_maps[k]) is nbr read k-major, out row ascending.
 * Returns the number of pairs. */
int64_t orc_kernel_map(const int32_t *in_coords, int64_t n_in, const int32_t *out_coords, int64_t n_out,
                       const int32_t *offsets, int32_t K, int32_t *nbr) {
    orc_table t;
    if (orc_table_init(&t, n_in, in_coords)) return -1;
    for (int64_t i = 0; i < n_in; ++i) orc_table_insert(&t, in_coords + 4 * i, (int32_t)i);
    int64_t pairs = 0;
#pragma omp parallel for reduction(+ : pairs) schedule(static)
    for (int64_t o = 0; o < n_out; ++o) {
        for (int32_t k = 0; k < K; ++k) {
            int32_t key[4];
            key[0] = out_coords[4 * o];
            for (int d = 0; d < 3; ++d) key[d + 1] = out_coords[4 * o + d + 1] + offsets[3 * k + d];
            int32_t r = orc_table_find(&t, key);
            nbr[o * K + k] = r;
            pairs += (r >= 0);
        }
    }
    orc_table_free(&t);
    return pairs;
}

/* ME rule book from the neighbour table: for each k, pairs (in,out) in
 * ascending out-row order.  k_off has K+1 entries. */
void orc_pairs_from_nbr(const int32_t *nbr, int64_t n_out, int32_t K, int64_t *k_off, int32_t *pair_in,
                        int32_t *pair_out) {
    int64_t p = 0;
    for (int32_t k = 0; k < K; ++k) {
        k_off[k] = p;
        for (int64_t o = 0; o < n_out; ++o) {
            int32_t r = nbr[o * K + k];
            if (r >= 0) { pair_in[p] = r; pair_out[p] = (int32_t)o; ++p; }
        }
    }
    k_off[K] = p;
}

/* Sparse convolution forward, ME CPU algorithm: for every kernel offset k in
 * ascending order, gather the input rows of its pairs, multiply by W[k]
 * ([Cin,Cout] row-major slice of the [K,Cin,Cout] kernel) and add the product
 * rows into the output rows.  The product of one pair is an fmaf chain over
 * ci ascending starting from 0; it is then ADDED to the output (separate
 * rounding), which is the gather->GEMM->scatter-add order.  `out` must be
 * zero-filled (or hold the bias) by the caller.
 * A transposed convolution is the same call with pair_in / pair_out swapped. */
void orc_conv_fwd(const float *in, const float *W, const int64_t *k_off, const int32_t *pair_in,
                  const int32_t *pair_out, int32_t K, int32_t Cin, int32_t Cout, float *out) {
    for (int32_t k = 0; k < K; ++k) {
        const float *Wk = W + (int64_t)k * Cin * Cout;
#pragma omp parallel for schedule(static)
        for (int64_t p = k_off[k]; p < k_off[k + 1]; ++p) {
            const float *x = in + (int64_t)pair_in[p] * Cin;
            float *y = out + (int64_t)pair_out[p] * Cout;
            for (int32_t c = 0; c < Cout; ++c) {
                float t = 0.0f;
                for (int32_t ci = 0; ci < Cin; ++ci) t = fmaf(x[ci], Wk[(int64_t)ci * Cout + c], t);
                y[c] += t;
            }
        }
    }
}

/* dL/d(in): gin[i] += gout[o] . W[k]^T, same pair order; chain over c ascending. */
void orc_conv_bwd_data(const float *gout, const float *W, const int64_t *k_off, const int32_t *pair_in,
                       const int32_t *pair_out, int32_t K, int32_t Cin, int32_t Cout, float *gin) {
    for (int32_t k = 0; k < K; ++k) {
        const float *Wk = W + (int64_t)k * Cin * Cout;
#pragma omp parallel for schedule(static)
        for (int64_t p = k_off[k]; p < k_off[k + 1]; ++p) {
            const float *g = gout + (int64_t)pair_out[p] * Cout;
            float *y = gin + (int64_t)pair_in[p] * Cin;
            for (int32_t ci = 0; ci < Cin; ++ci) {
                float t = 0.0f;
                for (int32_t c = 0; c < Cout; ++c) t = fmaf(g[c], Wk[(int64_t)ci * Cout + c], t);
                y[ci] += t;
            }
        }
    }
}

/* dL/dW[k] = sum over pairs of in[i]^T . gout[o]; accumulated in double so the
 * oracle is a tight reference for the GPU's tree reductions. */
void orc_conv_bwd_weight(const float *in, const float *gout, const int64_t *k_off, const int32_t *pair_in,
                         const int32_t *pair_out, int32_t K, int32_t Cin, int32_t Cout, float *gW) {
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int32_t k = 0; k < K; ++k) {
        for (int32_t ci = 0; ci < Cin; ++ci) {
            double *acc = (double *)calloc((size_t)Cout, sizeof(double));
            for (int64_t p = k_off[k]; p < k_off[k + 1]; ++p) {
                double x = in[(int64_t)pair_in[p] * Cin + ci];
                const float *g = gout + (int64_t)pair_out[p] * Cout;
                for (int32_t c = 0; c < Cout; ++c) acc[c] += x * (double)g[c];
            }
            float *dst = gW + ((int64_t)k * Cin + ci) * Cout;
            for (int32_t c = 0; c < Cout; ++c) dst[c] = (float)acc[c];
            free(acc);
        }
    }
}

/* ME.utils.sparse_quantize core (semantickitti_bev.py:232-238 call site):
 * integer voxel coordinates [n,3] (already floor(p/q)) -> first point of each
 * distinct voxel, in order of that first point.  labels may be NULL; a voxel
 * whose points disagree on the label gets ignore_label.
 * index[j] = first point of voxel j; inverse[i] = voxel of point i. */
int64_t orc_sparse_quantize(const int32_t *vox, int64_t n, const int32_t *labels, int32_t ignore_label,
                            int32_t *index, int32_t *inverse, int32_t *voxel_labels) {
    int32_t *c4 = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) {
        c4[4 * i] = 0;
        c4[4 * i + 1] = vox[3 * i];
        c4[4 * i + 2] = vox[3 * i + 1];
        c4[4 * i + 3] = vox[3 * i + 2];
    }
    int64_t m = orc_unique_first(c4, n, index, inverse);
    free(c4);
    if (m < 0) return m;
    if (labels && voxel_labels) {
        for (int64_t j = 0; j < m; ++j) voxel_labels[j] = labels[index[j]];
        for (int64_t i = 0; i < n; ++i)
            if (labels[i] != voxel_labels[inverse[i]]) voxel_labels[inverse[i]] = ignore_label;
    }
    return m;
}

/* Helpers of the "blas" mode of oracle/me_cpu (the timed CPU baseline): ME's CPU convolution gathers the input
 * rows of one kernel offset into a dense buffer, calls a BLAS GEMM and scatter-adds the result rows.  Within one
 * offset every output row occurs at most once, so the scatter-add is race-free across threads. */
void orc_gather_rows(const float *src, const int32_t *idx, int64_t n, int32_t C, float *dst) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) memcpy(dst + i * C, src + (int64_t)idx[i] * C, sizeof(float) * (size_t)C);
}

void orc_scatter_add_rows(const float *src, const int32_t *idx, int64_t n, int32_t C, float *dst) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float *d = dst + (int64_t)idx[i] * C;
        const float *s = src + i * C;
        for (int32_t c = 0; c < C; ++c) d[c] += s[c];
    }
}

#ifdef _OPENMP
#include <omp.h>
void orc_set_threads(int32_t n) { omp_set_num_threads(n); }
#else
void orc_set_threads(int32_t n) { (void)n; }
#endif
