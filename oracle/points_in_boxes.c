/* oracle/points_in_boxes.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C restatement of the reference's point-in-rotated-box test:
 *   /root/reference/lidargen/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:121-168
 *     (lidar_to_local_coords_cpu, check_pt_in_box3d_cpu MARGIN 1e-2, points_in_boxes_cpu)
 *   /root/reference/lidargen/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-36,313-336
 *     (check_pt_in_box3d MARGIN 1e-5, points_in_boxes_kernel: first containing box or -1)
 * Arithmetic contract shared with the HIP kernel (csrc/geometry.hip pt_in_box): rotation in
 * float with cos/sin of -heading correctly rounded to float (evaluated in double), the three
 * comparisons in double exactly as the reference's `dz / 2.0`, `dx / 2.0 + MARGIN` promote.
 */
#include <math.h>

static int pt_in_box(const float* pt, const float* b, float margin) {
    const float x = pt[0], y = pt[1], z = pt[2];
    const float cx = b[0], cy = b[1], cz = b[2], dx = b[3], dy = b[4], dz = b[5], rz = b[6];
    if ((double)fabsf(z - cz) > (double)dz / 2.0) return 0;
    const float cosa = (float)cos((double)(-rz)), sina = (float)sin((double)(-rz));
    const float sx = x - cx, sy = y - cy;
    const float lx = sx * cosa + sy * (-sina);
    const float ly = sx * sina + sy * cosa;
    return ((double)fabsf(lx) < (double)dx / 2.0 + (double)margin) &
           ((double)fabsf(ly) < (double)dy / 2.0 + (double)margin);
}

/* out[N_box, M] = 0/1 */
void oracle_points_in_boxes_mask(const float* boxes, int nb, const float* pts, int np,
                                 float margin, int* out) {
    for (int i = 0; i < nb; ++i)
        for (int j = 0; j < np; ++j) out[(long)i * np + j] = pt_in_box(pts + 3L * j, boxes + 7 * i, margin);
}

/* out[B, M] = index of the first containing box, -1 if none */
void oracle_points_in_boxes_index(const float* boxes, const float* pts, int B, int nb, int np,
                                  float margin, int* out) {
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < np; ++j) {
            int idx = -1;
            for (int k = 0; k < nb; ++k)
                if (pt_in_box(pts + ((long)b * np + j) * 3, boxes + ((long)b * nb + k) * 7, margin)) {
                    idx = k;
                    break;
                }
            out[(long)b * np + j] = idx;
        }
}
