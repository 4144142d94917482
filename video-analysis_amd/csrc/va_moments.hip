// va_moments.hip -- A9: moments of a closed contour, cv2.moments(contour)
//
// replaces  regionprops(contour=...):  cv2.moments(contour),                video/analysis/image.py:355
//           Polygon.moments:           cv2.moments(np.asarray(contour, np.float32)),
//                                                                            video/analysis/shapes.py:527-533
//
// Green's theorem over the polygon: ten float64 accumulators that OpenCV's contourMoments
// updates point by point.  The sums are NOT exact in float64 for large frames (a 1080p term of
// a30 reaches 1e17 > 2^53), so the result depends on the order of the additions; to reproduce
// OpenCV's values bit for bit the walk over one contour stays sequential, in OpenCV's order, on a
// single lane (built with -ffp-contract=off).  Contours are a few hundred points long and the
// batch supplies the parallelism: one wave per frame, so the frames spread over the CUs.
// The central / normalised moments are completed on the host from these ten values, like the
// raster moments (video/analysis/image.py: moments_from_spatial).
#include "va_common.h"

namespace va {
namespace {

template <bool IS_FLOAT>
__global__ __launch_bounds__(64) void contour_moments_kernel(const void *__restrict__ points,
                                                             const int32_t *__restrict__ npoints,
                                                             int max_points, double *__restrict__ out)
{
    if (threadIdx.x != 0)
        return;
    const int f = blockIdx.x;
    int n = npoints ? npoints[f] : max_points;
    n = n < 0 ? 0 : (n > max_points ? max_points : n);
    const int32_t *pi = (const int32_t *)points + (size_t)f * max_points * 2;
    const float *pf = (const float *)points + (size_t)f * max_points * 2;
    double *o = out + (size_t)f * 10;
    for (int k = 0; k < 10; k++)
        o[k] = 0.0;
    if (n == 0)
        return;
    auto px = [&](int i) { return IS_FLOAT ? (double)pf[2 * i] : (double)pi[2 * i]; };
    auto py = [&](int i) { return IS_FLOAT ? (double)pf[2 * i + 1] : (double)pi[2 * i + 1]; };
    double a00 = 0, a10 = 0, a01 = 0, a20 = 0, a11 = 0, a02 = 0, a30 = 0, a21 = 0, a12 = 0, a03 = 0;
    double xi_1 = px(n - 1), yi_1 = py(n - 1);
    double xi_12 = xi_1 * xi_1, yi_12 = yi_1 * yi_1;
    for (int i = 0; i < n; i++) {
        const double xi = px(i), yi = py(i);
        const double xi2 = xi * xi, yi2 = yi * yi;
        const double dxy = xi_1 * yi - xi * yi_1;
        const double xii_1 = xi_1 + xi, yii_1 = yi_1 + yi;
        a00 += dxy;
        a10 += dxy * xii_1;
        a01 += dxy * yii_1;
        a20 += dxy * (xi_1 * xii_1 + xi2);
        a11 += dxy * (xi_1 * (yii_1 + yi_1) + xi * (yii_1 + yi));
        a02 += dxy * (yi_1 * yii_1 + yi2);
        a30 += dxy * xii_1 * (xi_12 + xi2);
        a03 += dxy * yii_1 * (yi_12 + yi2);
        a21 += dxy * (xi_12 * (3 * yi_1 + yi) + 2 * xi * xi_1 * yii_1 + xi2 * (yi_1 + 3 * yi));
        a12 += dxy * (yi_12 * (3 * xi_1 + xi) + 2 * yi * yi_1 * xii_1 + yi2 * (xi_1 + 3 * xi));
        xi_1 = xi;
        yi_1 = yi;
        xi_12 = xi2;
        yi_12 = yi2;
    }
    if (fabs(a00) > 1.1920928955078125e-07) {       // FLT_EPSILON
        const double s = a00 > 0 ? 1.0 : -1.0;     // m00 >= 0 for either orientation
        o[0] = a00 * (s * 0.5);
        o[1] = a10 * (s * 0.16666666666666666666666666666667);
        o[2] = a01 * (s * 0.16666666666666666666666666666667);
        o[3] = a20 * (s * 0.083333333333333333333333333333333);
        o[4] = a11 * (s * 0.041666666666666666666666666666667);
        o[5] = a02 * (s * 0.083333333333333333333333333333333);
        o[6] = a30 * (s * 0.05);
        o[7] = a21 * (s * 0.016666666666666666666666666666667);
        o[8] = a12 * (s * 0.016666666666666666666666666666667);
        o[9] = a03 * (s * 0.05);
    }
}

}  // namespace

int launch_contour_moments(const void *points, const int32_t *npoints, int n, int max_points,
                           int is_float, double *out, hipStream_t st)
{
    VA_REQUIRE(points && out && n >= 0 && max_points > 0, "contour_moments: bad argument");
    if (n == 0)
        return VA_OK;
    if (is_float)
        contour_moments_kernel<true><<<n, 64, 0, st>>>(points, npoints, max_points, out);
    else
        contour_moments_kernel<false><<<n, 64, 0, st>>>(points, npoints, max_points, out);
    VA_LAUNCH_CHECK("contour_moments_kernel");
    return VA_OK;
}

}  // namespace va
