/* Oracle (TEST INFRASTRUCTURE ONLY): plain-C restatement of the depth un-projection and the integer
 * grid-cell indexing of the reference.  Never linked into the product.
 *
 *   a1  ProjectorUtils.pixel_to_world_mapping + compute_scaling_params
 *       (Detic/SMNet/projector/core.py:68-114, 116-149, 177-225)
 *   a2  grid-cell index (Detic/SMNet/build_memory_data.py:135-144; robot ordering
 *       Detic/robot_demo.py:526-534)
 *
 * Operation order (all IEEE fp32, no FMA contraction: build with -ffp-contract=off):
 *   xs = ((u + 0.5) - cx) / fx          ys = ((v + 0.5) - cy) / fy
 *   z  = d ;  x = z * xs ;  y = z * ys
 *   w_i = ((T[i][0]*x + T[i][1]*y) + T[i][2]*z) + T[i][3]      (row i of the 4x4 bmm, k ascending)
 *   w_i -= proj_shift[i]   (ProjectorUtils.world_shift_origin, core.py:220)
 *   w_i -= map_shift[i]    (map_world_shift, build_memory_data.py:135)
 *   ix = clip(rint(w_0 / cell), 0, map_w-1) ;  iz = clip(rint(w_2 / cell), 0, map_h-1)
 *   idx = iz*map_w + ix  (order 0, offline data)   |   ix*map_h + iz  (order 1, robot demo)
 * rint = round-half-to-even = torch.round.  The bmm accumulation order of the reference's BLAS is not
 * specified; k-ascending without FMA is this build's definition (pinned against the reference projector
 * on CPU in tests/test_oracle_golden.py, where the index agrees on all pixels of the fixture).
 */
#include <math.h>
#include <stdint.h>

#pragma STDC FP_CONTRACT OFF

void oracle_unproject_world(const float *depth, int H, int W, const float *T /*16*/,
                            float fx, float fy, float cx, float cy,
                            const float *proj_shift /*3*/, float *xyz /*H*W*3*/)
{
    for (int v = 0; v < H; ++v) {
        for (int u = 0; u < W; ++u) {
            float xs = (((float)u + 0.5f) - cx) / fx;
            float ys = (((float)v + 0.5f) - cy) / fy;
            float z = depth[(long)v * W + u];
            float x = z * xs;
            float y = z * ys;
            for (int i = 0; i < 3; ++i) {
                float a = T[i * 4 + 0] * x;
                float b = T[i * 4 + 1] * y;
                float c = T[i * 4 + 2] * z;
                float w = ((a + b) + c) + T[i * 4 + 3];
                w = w - proj_shift[i];
                xyz[((long)v * W + u) * 3 + i] = w;
            }
        }
    }
}

void oracle_grid_index(const float *xyz /*P*3*/, long P, const float *map_shift /*3*/, float cell,
                       int map_w, int map_h, int order, int32_t *idx /*P*/)
{
    for (long p = 0; p < P; ++p) {
        float wx = xyz[p * 3 + 0] - map_shift[0];
        float wz = xyz[p * 3 + 2] - map_shift[2];
        float qx = rintf(wx / cell);
        float qz = rintf(wz / cell);
        /* .long() of a float: values are clipped right after, so saturate instead of UB on huge values */
        long ix = (qx != qx) ? 0 : (qx < -1e9f ? -1000000000L : (qx > 1e9f ? 1000000000L : (long)qx));
        long iz = (qz != qz) ? 0 : (qz < -1e9f ? -1000000000L : (qz > 1e9f ? 1000000000L : (long)qz));
        if (ix < 0) ix = 0;
        if (ix > map_w - 1) ix = map_w - 1;
        if (iz < 0) iz = 0;
        if (iz > map_h - 1) iz = map_h - 1;
        idx[p] = (int32_t)(order == 0 ? iz * map_w + ix : ix * map_h + iz);
    }
}
