// Data-path kernels next to the hot path: voxelisation helpers for
// ME.utils.sparse_quantize (utils/datasets/semantickitti_bev.py:232-238) and the BEV label rasteriser
// PC2ImgConverter.getBEVImageNew (utils/datasets/semantickitti_bev.py:433-464).  The unique / first-occurrence /
// inverse-map part of sparse_quantize reuses lidog_coords_insert + lidog_coords_compact (coords.hip).
#include "common.h"

// rows[i] = (batch, floor(x/q), floor(y/q), floor(z/q)); float32 division and floor exactly as
// np.floor(points / quantization_size) on a float32 array
__global__ __launch_bounds__(256) void k_voxel_floor(const float *__restrict__ pts, int64_t n, float qx, float qy,
                                                     float qz, int batch, int4 *__restrict__ rows) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    rows[i] = make_int4(batch, (int)floorf(x / qx), (int)floorf(y / qy), (int)floorf(z / qz));
}

extern "C" int lidog_voxel_floor(const float *points, int64_t n, float qx, float qy, float qz, int32_t batch,
                                 int32_t *rows, void *stream) {
    if (n == 0) return 0;
    k_voxel_floor<<<(unsigned)cdiv64(n, 256), 256, 0, (hipStream_t)stream>>>(points, n, qx, qy, qz, batch,
                                                                           (int4 *)rows);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// voxel label = label of its first point, or ignore_label when any point of the voxel disagrees
__global__ __launch_bounds__(256) void k_label_init(const int32_t *__restrict__ labels,
                                                    const int32_t *__restrict__ unique_rows, int64_t m,
                                                    int32_t *__restrict__ voxel_labels) {
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < m) voxel_labels[j] = labels[unique_rows[j]];
}
__global__ __launch_bounds__(256) void k_label_vote(const int32_t *__restrict__ labels,
                                                    const int32_t *__restrict__ unique_rows,
                                                    const int32_t *__restrict__ inverse, int64_t n, int ignore_label,
                                                    int32_t *voxel_labels) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int v = inverse[i];
    // compare against the FIRST point's label (immutable input), so concurrent writers all store the same value
    if (labels[i] != labels[unique_rows[v]]) voxel_labels[v] = ignore_label;
}

extern "C" int lidog_label_vote(const int32_t *labels, const int32_t *unique_rows, const int32_t *inverse, int64_t n,
                                int64_t m, int32_t ignore_label, int32_t *voxel_labels, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (m) k_label_init<<<(unsigned)cdiv64(m, 256), 256, 0, st>>>(labels, unique_rows, m, voxel_labels);
    if (n) k_label_vote<<<(unsigned)cdiv64(n, 256), 256, 0, st>>>(labels, unique_rows, inverse, n, ignore_label,
                                                                  voxel_labels);
    LIDOG_LAUNCH_CHECK();
    return 0;
}

// BEV label image: the LAST valid in-bounds point of a pixel wins (numpy fancy assignment order).
// point_idx [B,S,S] must be pre-filled with -1; img_labels is derived from it afterwards.
__global__ __launch_bounds__(256) void k_bev_label_winner(const int4 *__restrict__ coords,
                                                          const int32_t *__restrict__ labels, int64_t n,
                                                          const int32_t *__restrict__ lut_x,
                                                          const int32_t *__restrict__ lut_y,
                                                          const int32_t *__restrict__ lut_z, int lut_lo, int lut_n,
                                                          int S, const int64_t *__restrict__ batch_start,
                                                          int32_t *point_idx) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (labels[i] == -1) return;
    int4 c = coords[i];
    int ix = c.y - lut_lo, iy = c.z - lut_lo, iz = c.w - lut_lo;
    if (ix < 0 || ix >= lut_n || iy < 0 || iy >= lut_n || iz < 0 || iz >= lut_n) return;
    int px = lut_x[ix], py = lut_y[iy];
    if (px < 0 || py < 0 || lut_z[iz] == 0) return;
    // index of the point inside its own scan, as the reference's imgPointsIdx stores it
    atomicMax(&point_idx[((int64_t)c.x * S + py) * S + px], (int32_t)(i - batch_start[c.x]));
}
__global__ __launch_bounds__(256) void k_bev_label_fill(const int32_t *__restrict__ labels,
                                                        const int32_t *__restrict__ point_idx, int64_t total, int S,
                                                        const int64_t *__restrict__ batch_start,
                                                        int64_t *__restrict__ img_labels) {
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    int b = (int)(e / ((int64_t)S * S));
    int p = point_idx[e];
    img_labels[e] = (p >= 0) ? (int64_t)labels[batch_start[b] + p] : -1;
}

extern "C" int lidog_bev_label_raster(const int32_t *coords, const int32_t *labels, int64_t n, const int32_t *lut_x,
                                      const int32_t *lut_y, const int32_t *lut_z, int32_t lut_lo, int32_t lut_n,
                                      int32_t B, int32_t S, const int64_t *batch_start, int32_t *point_idx,
                                      int64_t *img_labels, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    int64_t total = (int64_t)B * S * S;
    if (n)
        k_bev_label_winner<<<(unsigned)cdiv64(n, 256), 256, 0, st>>>((const int4 *)coords, labels, n, lut_x, lut_y,
                                                                     lut_z, lut_lo, lut_n, S, batch_start, point_idx);
    if (total)
        k_bev_label_fill<<<(unsigned)cdiv64(total, 256), 256, 0, st>>>(labels, point_idx, total, S, batch_start,
                                                                       img_labels);
    LIDOG_LAUNCH_CHECK();
    return 0;
}
