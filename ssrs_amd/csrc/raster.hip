// K1 -- updraft raster kernels for gfx950 (MI355X).
//
// Reference semantics (paths relative to /root/reference):
//   ssrs/layers.py:11-22    compute_orographic_updraft
//   ssrs/layers.py:63-128   compute_slope_degrees / compute_aspect_degrees
//   ssrs/layers.py:171-185  get_above_threshold_speed
//   ssrs/simulator.py:189-243  orographic -> np.save(f32) -> load -> threshold
//
// All arithmetic is f64 like the reference; the kernels are HBM-streaming:
//   k_orographic       12 B/cell (f32 slope+aspect in, f32 out), +8 B if the
//                      thresholded f64 raster is also written; batched over B
//                      wind cases so the terrain is read once ((8+12B)/B B)
//   k_updraft_from_dem 8..12 B/cell: DEM tile (+1-cell halo) staged in LDS,
//                      Horn gradients -> updraft with no trig at all
// Built with -ffp-contract=off (see build.py).
#include "common.h"

namespace ssrs {

constexpr double kPi = 3.141592653589793;  // np.pi

// ----------------------------------------------------------------------------
// XCD-aware tile order: blocks are dealt round-robin over the 8 XCDs, so give
// block b the tile (b % 8) * ceil(n/8) + b / 8 (bijective form): each XCD then
// walks one contiguous band of tiles and finds its halo rows in its own L2.
__device__ __forceinline__ int xcd_tile(int b, int n)
{
    const int q = n / 8, r = n % 8, x = b % 8, j = b / 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// layers.py:171-185 on one value already rounded to f32 and widened again
__device__ __forceinline__ double usable_updraft(double v, double thr, double em1)
{
    double f = 0.0;
    if (v > 1e-02) {
        if (v > thr) {
            f = v;
        } else {
            const double y = v / thr;
            const double y2 = y * y;
            const double y5 = (y2 * y2) * y;       // (v/thr)**5
            f = thr * (exp(y5) - 1.0) / em1;       // em1 = e - 1 (host libm)
        }
    }
    return f;
}

// ---------------------------------------------------------------------------
// DEM tile staging: TW x TH outputs per block, (TW+2) x (TH+2) f64 in LDS.
constexpr int TW = 64;          // one wave spans a tile row: 512 B coalesced
constexpr int TH = 32;
constexpr int LW = TW + 2;
constexpr int LH = TH + 2;

template <typename T>
__device__ __forceinline__ void stage_dem_tile(const T *__restrict__ dem, int rows,
                                               int cols, int r0, int c0,
                                               double *__restrict__ tile)
{
    for (int i = threadIdx.x; i < LW * LH; i += kBlock) {
        const int lr = i / LW, lc = i - lr * LW;
        int gr = r0 - 1 + lr, gc = c0 - 1 + lc;
        gr = gr < 0 ? 0 : (gr >= rows ? rows - 1 : gr);   // clamped cells are only
        gc = gc < 0 ? 0 : (gc >= cols ? cols - 1 : gc);   // read by border outputs
        tile[i] = static_cast<double>(dem[static_cast<size_t>(gr) * cols + gc]);
    }
    __syncthreads();
}

// Horn gradients exactly in the reference's operand order (layers.py:78-90);
// "x" is the ROW axis and "y" the COLUMN axis there.
__device__ __forceinline__ void horn(const double *__restrict__ t, int lr, int lc,
                                     double res, double &dzdx, double &dzdy)
{
    const double *m = t + (lr - 1) * LW + lc;
    const double *z = t + lr * LW + lc;
    const double *p = t + (lr + 1) * LW + lc;
    const double z1 = m[1], z2 = z[1], z3 = p[1];
    const double z4 = m[0], z6 = p[0];
    const double z7 = m[-1], z8 = z[-1], z9 = p[-1];
    const double d = 8 * res;
    dzdx = ((z3 + 2 * z6 + z9) - (z1 + 2 * z4 + z7)) / d;
    dzdy = ((z1 + 2 * z2 + z3) - (z7 + 2 * z8 + z9)) / d;
}

template <typename Tin, typename Tout>
__global__ __launch_bounds__(kBlock) void k_slope_aspect(
    const Tin *__restrict__ dem, double res, Tout *__restrict__ slope,
    Tout *__restrict__ aspect, int rows, int cols, int tiles_x, int ntiles)
{
    __shared__ double tile[LW * LH];
    const int t = xcd_tile(blockIdx.x, ntiles);
    const int r0 = (t / tiles_x) * TH, c0 = (t % tiles_x) * TW;
    stage_dem_tile(dem, rows, cols, r0, c0, tile);
    const int lc = threadIdx.x % TW + 1;
    const int c = c0 + lc - 1;
    if (c >= cols) return;
    const double r2d = 180.0 / kPi;
    for (int lr = threadIdx.x / TW + 1; lr <= TH; lr += kBlock / TW) {
        const int r = r0 + lr - 1;
        if (r >= rows) break;
        double s = 0.0, a = 0.0;
        if (r > 0 && c > 0 && r < rows - 1 && c < cols - 1) {
            double dzdx, dzdy;
            horn(tile, lr, lc, res, dzdx, dzdy);
            s = atan(sqrt(dzdx * dzdx + dzdy * dzdy)) * r2d;   // np.degrees
            const double dx = dzdx == 0.0 ? 1e-10 : dzdx;
            const double ang = atan(dzdy / dx) * r2d;
            a = 180.0 - ang + 90.0 * (dx / fabs(dx));
        }
        const size_t i = static_cast<size_t>(r) * cols + c;
        if (slope) slope[i] = static_cast<Tout>(s);
        if (aspect) aspect[i] = static_cast<Tout>(a);
    }
}

// Thresholded updraft with the two divisions folded into host constants
// (v / thr -> v * inv_thr, . / (e - 1) -> . * scale) and exp(x) - 1 on x = (v/thr)^5 in (0, 1] as
// its Taylor polynomial to x^17 (truncation 1.6e-16 relative; libm's exp(x) - 1 carries 1.1e-16
// ABSOLUTE, so the two agree to 1e-16 absolute and the polynomial is the more accurate one for
// small x): 17 fused multiply-adds instead of ocml's exp (range reduction, ldexp and three
// branches).  Inside the rtol 1e-12 / atol 1e-15 the tests state.
__device__ __forceinline__ double expm1_unit(double x)
{
    constexpr double c[17] = {
        1.00000000000000000e+00,
        5.00000000000000000e-01,
        1.66666666666666657e-01,
        4.16666666666666644e-02,
        8.33333333333333322e-03,
        1.38888888888888894e-03,
        1.98412698412698413e-04,
        2.48015873015873016e-05,
        2.75573192239858925e-06,
        2.75573192239858883e-07,
        2.50521083854417202e-08,
        2.08767569878681002e-09,
        1.60590438368216133e-10,
        1.14707455977297245e-11,
        7.64716373181981641e-13,
        4.77947733238738525e-14,
        2.81145725434552060e-15};
    double p = c[16];
#pragma unroll
    for (int k = 15; k >= 0; --k) p = __builtin_fma(p, x, c[k]);
    return x * p;
}

__device__ __forceinline__ double usable_updraft_fast(double v, double thr, double inv_thr,
                                                      double scale)
{
    double f = 0.0;
    if (v > 1e-02) {
        if (v > thr) {
            f = v;
        } else {
            const double y = v * inv_thr;
            const double y2 = y * y;
            f = scale * expm1_unit((y2 * y2) * y);          // thr (exp((v/thr)^5) - 1) / (e - 1)
        }
    }
    return f;
}

// 1 / sqrt(s) for a normal positive s: the hardware estimate and two Newton steps (ocml's rsqrt
// adds scaling for denormals and special cases these sums of squares never need)
__device__ __forceinline__ double rsqrt_pos(double s)
{
    double r = __builtin_amdgcn_rsq(s);
    const double h = 0.5 * s;
#pragma unroll
    for (int k = 0; k < 2; ++k) r = r * __builtin_fma(-h, r * r, 1.5);
    return r;
}

struct FusedArgs {
    double d, d2;             // 8 res and its square
    double wspeed, cos_w, sin_w, min_val;
    double thr, inv_thr, scale;
};

template <typename Tin>
__global__ __launch_bounds__(kBlock) void k_updraft_from_dem(
    const Tin *__restrict__ dem, FusedArgs fa, float *__restrict__ orograph,
    double *__restrict__ usable, int rows, int cols, int tiles_x, int ntiles)
{
    __shared__ double tile[LW * LH];
    const int t = xcd_tile(blockIdx.x, ntiles);
    const int r0 = (t / tiles_x) * TH, c0 = (t % tiles_x) * TW;
    stage_dem_tile(dem, rows, cols, r0, c0, tile);
    const int lc = threadIdx.x % TW + 1;
    const int c = c0 + lc - 1;
    if (c >= cols) return;
    // each wave owns TH/4 consecutive rows of the tile and slides a 3 x 3 window
    // down its column: 3 new LDS reads per cell instead of 8
    constexpr int kRowsPerWave = TH / (kBlock / TW);
    const int lr0 = (threadIdx.x / TW) * kRowsPerWave + 1;
    const double *q = tile + (lr0 - 1) * LW + lc;
    double m_l = q[-1], m_c = q[0], m_r = q[1];               // row lr - 1
    double z_l = q[LW - 1], z_c = q[LW], z_r = q[LW + 1];     // row lr
    for (int lr = lr0; lr < lr0 + kRowsPerWave; ++lr) {
        const int r = r0 + lr - 1;
        if (r >= rows) break;
        const double *p = tile + (lr + 1) * LW + lc;
        const double p_l = p[-1], p_c = p[0], p_r = p[1];     // row lr + 1
        double w = 0.0;
        if (r > 0 && c > 0 && r < rows - 1 && c < cols - 1) {
            // un-normalised Horn sums X = 8 res dz_dx, Y = 8 res dz_dy in the
            // reference's operand order (layers.py:78-90; "x" = row axis)
            const double X = (p_r + 2 * p_c + p_l) - (m_r + 2 * m_c + m_l);
            const double Y = (m_r + 2 * z_r + p_r) - (m_l + 2 * z_l + p_l);
            if (X != 0.0) {
                // sin(slope) cos(aspect - wdirn) = -(dz_dy cos w + dz_dx sin w) / sqrt(1 + g^2)
                //                                 = -(Y cos w + X sin w) / sqrt(d^2 + X^2 + Y^2)
                // (sin(atan g) = g / sqrt(1 + g^2); cos(aspect - w) from aspect =
                //  180 - atan(dz_dy/dz_dx) + 90 sign(dz_dx)): no trig, no division.
                const double P = -(Y * fa.cos_w + X * fa.sin_w);
                if (P > 0.0) w = fa.wspeed * (P * rsqrt_pos(fa.d2 + (X * X + Y * Y)));
            } else {
                // dz_dx == 0: the reference substitutes 1e-10 for the aspect only
                // (layers.py:124) -- rare, evaluated in the explicit form
                const double dzdy = Y / fa.d, dx = 1e-10;
                const double g2 = dzdy * dzdy, gp2 = dx * dx + g2;
                const double proj = -(dzdy * fa.cos_w + dx * fa.sin_w);
                if (proj > 0.0 && g2 > 0.0) w = fa.wspeed * (proj * sqrt(g2 / (gp2 * (1.0 + g2))));
            }
        }
        w = w > fa.min_val ? w : fa.min_val;
        const size_t i = static_cast<size_t>(r) * cols + c;
        const float w32 = static_cast<float>(w);
        if (orograph) orograph[i] = w32;
        if (usable)
            usable[i] = usable_updraft_fast(static_cast<double>(w32), fa.thr, fa.inv_thr, fa.scale);
        m_l = z_l; m_c = z_c; m_r = z_r;
        z_l = p_l; z_c = p_c; z_r = p_r;
    }
}

// sin / cos of an angle given in DEGREES.  The reference converts to radians
// first ((a - w) * pi / 180, two roundings) and calls libm; ocml's f64 sin/cos
// carry a Payne-Hanek path these bounded arguments never need and made
// k_orographic ALU-bound (170-180 us at C2).  Here the quadrant is removed
// exactly in degrees (x - 90 k is exact for |x| < 2^52), the remainder
// |r| <= 45 deg goes through the classic minimax kernels on [-pi/4, pi/4]
// (fdlibm k_sin / k_cos coefficients, < 1-2 ulp).  Result within ~1e-15 of the
// reference's value, far inside the 1-f32-ulp tolerance of the orograph.
__device__ __forceinline__ void sincos_deg(double x, double &sn, double &cs)
{
    const double kq = rint(x * (1.0 / 90.0));
    const double r = x - 90.0 * kq;                    // exact
    const double t = r * (kPi / 180.0);
    const double z = t * t;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double s0 = t + (t * z) * ps;
    const double c0 = 1.0 - (0.5 * z - (z * z) * pc);
    const int q = static_cast<int>(kq) & 3;
    sn = (q == 0) ? s0 : (q == 1) ? c0 : (q == 2) ? -s0 : -c0;
    cs = (q == 0) ? c0 : (q == 1) ? -s0 : (q == 2) ? -c0 : s0;
}

// ---------------------------------------------------------------------------
// Elementwise orographic updraft, VEC consecutive cells per thread (16-byte
// accesses for f32 rasters), batched over wind cases.
constexpr int kMaxUniformBatch = 16;
struct UniformWind {
    double wspeed[kMaxUniformBatch];
    double wdirn[kMaxUniformBatch];
};

template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Pack {
    T v[VEC];
};

template <typename Tin, typename Tw, bool UNIFORM, int VEC>
__global__ __launch_bounds__(kBlock) void k_orographic(
    const Tin *__restrict__ slope, const Tin *__restrict__ aspect,
    const Tw *__restrict__ wspeed, const Tw *__restrict__ wdirn, UniformWind uni,
    double min_val, float *__restrict__ orograph, double thr, double em1,
    double *__restrict__ usable, size_t ncells, int batch)
{
    const size_t nvec = ncells / VEC;   // host guarantees ncells % VEC == 0
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < nvec;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const Pack<Tin, VEC> s = reinterpret_cast<const Pack<Tin, VEC> *>(slope)[i];
        const Pack<Tin, VEC> a = reinterpret_cast<const Pack<Tin, VEC> *>(aspect)[i];
        double sin_s[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j)
        {
            double unused;
            sincos_deg(static_cast<double>(s.v[j]), sin_s[j], unused);
        }
        for (int b = 0; b < batch; ++b) {
            const size_t o = static_cast<size_t>(b) * nvec + i;
            Pack<Tw, VEC> ws, wd;
            if (!UNIFORM) {
                ws = reinterpret_cast<const Pack<Tw, VEC> *>(wspeed)[o];
                wd = reinterpret_cast<const Pack<Tw, VEC> *>(wdirn)[o];
            }
            Pack<float, VEC> out;
            Pack<double, VEC> use;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const double spd = UNIFORM ? uni.wspeed[b] : static_cast<double>(ws.v[j]);
                const double dir = UNIFORM ? uni.wdirn[b] : static_cast<double>(wd.v[j]);
                double ad, unused;
                sincos_deg(static_cast<double>(a.v[j]) - dir, unused, ad);
                ad = ad > 0.0 ? ad : 0.0;                      // np.maximum(0., .)
                double w = spd * (sin_s[j] * ad);
                w = w > min_val ? w : min_val;
                out.v[j] = static_cast<float>(w);
                if (usable)
                    use.v[j] = usable_updraft(static_cast<double>(out.v[j]), thr, em1);
            }
            if (orograph) reinterpret_cast<Pack<float, VEC> *>(orograph)[o] = out;
            if (usable) reinterpret_cast<Pack<double, VEC> *>(usable)[o] = use;
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_threshold(const float *__restrict__ in,
                                                      double thr, double em1,
                                                      double *__restrict__ out, size_t n)
{
    const size_t nvec = n / VEC;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < nvec;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const Pack<float, VEC> v = reinterpret_cast<const Pack<float, VEC> *>(in)[i];
        Pack<double, VEC> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            o.v[j] = usable_updraft(static_cast<double>(v.v[j]), thr, em1);
        reinterpret_cast<Pack<double, VEC> *>(out)[i] = o;
    }
}

static inline int stream_grid(size_t nthreads_needed)
{
    size_t b = (nthreads_needed + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > static_cast<size_t>(kMaxStreamBlocks)) b = kMaxStreamBlocks;
    return static_cast<int>(b);
}

static inline bool aligned16(const void *p)
{   // NULL counts as aligned (optional arrays); 32 B covers Pack<double, 4>
    return (reinterpret_cast<uintptr_t>(p) & 31u) == 0;
}

template <typename Tin, typename Tw, bool UNIFORM>
static int launch_orographic(const void *slope, const void *aspect, const void *wspeed,
                             const void *wdirn, const UniformWind &uni, double min_val,
                             float *orograph, double thr, double *usable, size_t ncells,
                             int batch, hipStream_t st)
{
    const double em1 = exp(1.0) - 1.0;
    const bool vec = (ncells % 4 == 0) && aligned16(slope) && aligned16(aspect) &&
                     aligned16(wspeed) && aligned16(wdirn) && aligned16(orograph) &&
                     aligned16(usable);
    auto s = static_cast<const Tin *>(slope);
    auto a = static_cast<const Tin *>(aspect);
    auto ws = static_cast<const Tw *>(wspeed);
    auto wd = static_cast<const Tw *>(wdirn);
    if (vec) {
        hipLaunchKernelGGL((k_orographic<Tin, Tw, UNIFORM, 4>), dim3(stream_grid(ncells / 4)),
                           dim3(kBlock), 0, st, s, a, ws, wd, uni, min_val, orograph, thr,
                           em1, usable, ncells, batch);
    } else {
        hipLaunchKernelGGL((k_orographic<Tin, Tw, UNIFORM, 1>), dim3(stream_grid(ncells)),
                           dim3(kBlock), 0, st, s, a, ws, wd, uni, min_val, orograph, thr,
                           em1, usable, ncells, batch);
    }
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}

}  // namespace ssrs

using namespace ssrs;

extern "C" int ssrs_slope_aspect(const void *dem, int dem_type, double res, void *slope,
                                 void *aspect, int out_type, int rows, int cols,
                                 void *stream)
{
    SSRS_REQUIRE(dem != nullptr, "ssrs_slope_aspect: dem is NULL");
    SSRS_REQUIRE(rows >= 3 && cols >= 3, "ssrs_slope_aspect: need rows, cols >= 3 (got %d x %d)",
                 rows, cols);
    SSRS_REQUIRE(res > 0.0, "ssrs_slope_aspect: res must be > 0");
    SSRS_REQUIRE((dem_type == SSRS_F32 || dem_type == SSRS_F64) &&
                     (out_type == SSRS_F32 || out_type == SSRS_F64),
                 "ssrs_slope_aspect: bad element type");
    if (!slope && !aspect) return SSRS_OK;
    const int tx = (cols + TW - 1) / TW, ty = (rows + TH - 1) / TH, nt = tx * ty;
    hipStream_t st = as_stream(stream);
#define SA_LAUNCH(TI, TO)                                                              \
    hipLaunchKernelGGL((k_slope_aspect<TI, TO>), dim3(nt), dim3(kBlock), 0, st,        \
                       static_cast<const TI *>(dem), res, static_cast<TO *>(slope),    \
                       static_cast<TO *>(aspect), rows, cols, tx, nt)
    if (dem_type == SSRS_F64 && out_type == SSRS_F64) SA_LAUNCH(double, double);
    else if (dem_type == SSRS_F64) SA_LAUNCH(double, float);
    else if (out_type == SSRS_F64) SA_LAUNCH(float, double);
    else SA_LAUNCH(float, float);
#undef SA_LAUNCH
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}

extern "C" int ssrs_updraft_from_dem(const void *dem, int dem_type, double res,
                                     double wspeed, double wdirn, double min_val,
                                     float *orograph, double threshold, double *usable,
                                     int rows, int cols, void *stream)
{
    SSRS_REQUIRE(dem != nullptr, "ssrs_updraft_from_dem: dem is NULL");
    SSRS_REQUIRE(rows >= 3 && cols >= 3, "ssrs_updraft_from_dem: need rows, cols >= 3");
    SSRS_REQUIRE(res > 0.0, "ssrs_updraft_from_dem: res must be > 0");
    SSRS_REQUIRE(dem_type == SSRS_F32 || dem_type == SSRS_F64,
                 "ssrs_updraft_from_dem: bad element type");
    SSRS_REQUIRE(!(usable && !(threshold > 0.0)),
                 "ssrs_updraft_from_dem: usable requested without a positive threshold");
    if (!orograph && !usable) return SSRS_OK;
    const int tx = (cols + TW - 1) / TW, ty = (rows + TH - 1) / TH, nt = tx * ty;
    const double w = wdirn * kPi / 180.0;
    FusedArgs fa;
    fa.d = 8 * res;
    fa.d2 = fa.d * fa.d;
    fa.wspeed = wspeed;
    fa.cos_w = cos(w);
    fa.sin_w = sin(w);
    fa.min_val = min_val;
    fa.thr = threshold;
    fa.inv_thr = threshold > 0.0 ? 1.0 / threshold : 0.0;
    fa.scale = threshold > 0.0 ? threshold / (exp(1.0) - 1.0) : 0.0;
    hipStream_t st = as_stream(stream);
    if (dem_type == SSRS_F64)
        hipLaunchKernelGGL((k_updraft_from_dem<double>), dim3(nt), dim3(kBlock), 0, st,
                           static_cast<const double *>(dem), fa, orograph, usable, rows, cols, tx, nt);
    else
        hipLaunchKernelGGL((k_updraft_from_dem<float>), dim3(nt), dim3(kBlock), 0, st,
                           static_cast<const float *>(dem), fa, orograph, usable, rows, cols, tx, nt);
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}

extern "C" int ssrs_orographic_updraft(const void *slope, const void *aspect, int in_type,
                                       const void *wspeed, const void *wdirn, int wind_type,
                                       const double *wspeed0, const double *wdirn0,
                                       double min_val, float *orograph, double threshold,
                                       double *usable, int rows, int cols, int batch,
                                       void *stream)
{
    SSRS_REQUIRE(slope && aspect, "ssrs_orographic_updraft: slope/aspect is NULL");
    SSRS_REQUIRE(rows > 0 && cols > 0 && batch > 0,
                 "ssrs_orographic_updraft: rows, cols, batch must be > 0");
    SSRS_REQUIRE((in_type == SSRS_F32 || in_type == SSRS_F64) &&
                     (wind_type == SSRS_F32 || wind_type == SSRS_F64),
                 "ssrs_orographic_updraft: bad element type");
    SSRS_REQUIRE((wspeed == nullptr) == (wdirn == nullptr),
                 "ssrs_orographic_updraft: give both wind rasters or neither");
    SSRS_REQUIRE(!(usable && threshold < 0.0),
                 "ssrs_orographic_updraft: usable requested with threshold < 0");
    if (!orograph && !usable) return SSRS_OK;
    const size_t ncells = static_cast<size_t>(rows) * cols;
    hipStream_t st = as_stream(stream);
    UniformWind uni = {};
    if (wspeed) {
        if (in_type == SSRS_F32 && wind_type == SSRS_F32)
            return launch_orographic<float, float, false>(slope, aspect, wspeed, wdirn, uni,
                                                          min_val, orograph, threshold,
                                                          usable, ncells, batch, st);
        if (in_type == SSRS_F32)
            return launch_orographic<float, double, false>(slope, aspect, wspeed, wdirn, uni,
                                                           min_val, orograph, threshold,
                                                           usable, ncells, batch, st);
        if (wind_type == SSRS_F32)
            return launch_orographic<double, float, false>(slope, aspect, wspeed, wdirn, uni,
                                                           min_val, orograph, threshold,
                                                           usable, ncells, batch, st);
        return launch_orographic<double, double, false>(slope, aspect, wspeed, wdirn, uni,
                                                        min_val, orograph, threshold, usable,
                                                        ncells, batch, st);
    }
    SSRS_REQUIRE(wspeed0 && wdirn0, "ssrs_orographic_updraft: uniform mode needs wspeed0/wdirn0");
    for (int b0 = 0; b0 < batch; b0 += kMaxUniformBatch) {
        const int nb = batch - b0 < kMaxUniformBatch ? batch - b0 : kMaxUniformBatch;
        for (int j = 0; j < nb; ++j) {
            uni.wspeed[j] = wspeed0[b0 + j];
            uni.wdirn[j] = wdirn0[b0 + j];
        }
        float *o = orograph ? orograph + static_cast<size_t>(b0) * ncells : nullptr;
        double *u = usable ? usable + static_cast<size_t>(b0) * ncells : nullptr;
        int rc = in_type == SSRS_F32
                     ? launch_orographic<float, float, true>(slope, aspect, nullptr, nullptr,
                                                             uni, min_val, o, threshold, u,
                                                             ncells, nb, st)
                     : launch_orographic<double, float, true>(slope, aspect, nullptr, nullptr,
                                                              uni, min_val, o, threshold, u,
                                                              ncells, nb, st);
        if (rc != SSRS_OK) return rc;
    }
    return SSRS_OK;
}

extern "C" int ssrs_threshold_updraft(const float *in, double threshold, double *out,
                                      size_t n, void *stream)
{
    SSRS_REQUIRE(in && out, "ssrs_threshold_updraft: NULL pointer");
    SSRS_REQUIRE(threshold > 0.0, "ssrs_threshold_updraft: threshold must be > 0");
    if (n == 0) return SSRS_OK;
    const double em1 = exp(1.0) - 1.0;
    hipStream_t st = as_stream(stream);
    if (n % 4 == 0 && aligned16(in) && aligned16(out))
        hipLaunchKernelGGL((k_threshold<4>), dim3(stream_grid(n / 4)), dim3(kBlock), 0, st, in,
                           threshold, em1, out, n);
    else
        hipLaunchKernelGGL((k_threshold<1>), dim3(stream_grid(n)), dim3(kBlock), 0, st, in,
                           threshold, em1, out, n);
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}

// ---------------------------------------------------------------------------
// K6 -- wind field preparation for snapshot / seasonal modes
// (ssrs/simulator.py:778-792): speed/direction samples -> easterly/northerly
// components -> interpolation to the terrain grid -> speed and direction
// (degrees in [0, 360)).  The reference interpolates scattered WTK points with
// scipy griddata (Qhull Delaunay, 'linear'); on a regular lattice -- the shape
// of the 2-km WTK grid and of the synthetic configs -- piecewise-bilinear
// interpolation is the natural equivalent and needs no triangulation.  The
// lattice (<= a few thousand points) is read through L2; one thread per cell.
namespace ssrs {

__global__ __launch_bounds__(kBlock) void k_wind_lattice(
    const double *__restrict__ lat_speed, const double *__restrict__ lat_dirn, int nx, int ny,
    double x0, double y0, double dx, double dy, double cell, double *__restrict__ wspeed,
    double *__restrict__ wdirn, int rows, int cols, int batch)
{
    const size_t ncell = static_cast<size_t>(rows) * cols;
    const size_t total = ncell * batch;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const int b = static_cast<int>(i / ncell);
        const size_t c0 = i - static_cast<size_t>(b) * ncell;
        const int r = static_cast<int>(c0 / cols), c = static_cast<int>(c0 % cols);
        // cell centre in lattice units, clamped to the lattice hull
        double fx = (c * cell - x0) / dx, fy = (r * cell - y0) / dy;
        fx = fx < 0.0 ? 0.0 : (fx > nx - 1.0 ? nx - 1.0 : fx);
        fy = fy < 0.0 ? 0.0 : (fy > ny - 1.0 ? ny - 1.0 : fy);
        int ix = static_cast<int>(fx), iy = static_cast<int>(fy);
        ix = ix > nx - 2 ? (nx > 1 ? nx - 2 : 0) : ix;
        iy = iy > ny - 2 ? (ny > 1 ? ny - 2 : 0) : iy;
        const double tx = nx > 1 ? fx - ix : 0.0, ty = ny > 1 ? fy - iy : 0.0;
        const double *ls = lat_speed + static_cast<size_t>(b) * nx * ny;
        const double *ld = lat_dirn + static_cast<size_t>(b) * nx * ny;
        double east = 0.0, north = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jx = ix + (k & 1 && nx > 1 ? 1 : 0), jy = iy + (k >> 1 && ny > 1 ? 1 : 0);
            const double w = ((k & 1) ? tx : 1.0 - tx) * ((k >> 1) ? ty : 1.0 - ty);
            const double s = ls[static_cast<size_t>(jy) * nx + jx];
            const double a = ld[static_cast<size_t>(jy) * nx + jx] * kPi / 180.0;
            east += w * (s * sin(a));       // simulator.py:784
            north += w * (s * cos(a));      // simulator.py:785
        }
        const double spd = sqrt(east * east + north * north);
        double ang = atan2(east, north);                               // :790
        ang = fmod(ang + 2.0 * kPi, 2.0 * kPi);                        // :791
        wspeed[i] = spd;
        wdirn[i] = ang * 180.0 / kPi;
    }
}

}  // namespace ssrs

namespace ssrs {

// Lattice speed / direction -> east and north components (simulator.py:784-785), once
// per lattice point instead of once per raster cell and corner.
__global__ __launch_bounds__(kBlock) void k_lattice_components(const double *__restrict__ lat_speed,
                                                              const double *__restrict__ lat_dirn,
                                                              size_t n, double *__restrict__ east,
                                                              double *__restrict__ north)
{
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const double s = lat_speed[i], a = lat_dirn[i] * kPi / 180.0;
        east[i] = s * sin(a);
        north[i] = s * cos(a);
    }
}

// Scattered wind samples (the reference's general case, simulator.py:765-776: scipy griddata, method 'linear' =
// Delaunay triangulation + barycentric interpolation).  The triangulation is the host's (scipy.spatial.Delaunay, the
// very object griddata builds); here: (1) every triangle claims the cells of its bounding box whose centre it
// contains -- by scipy's own test, all barycentric coordinates within [-eps, 1 + eps], eps = 100 DBL_EPSILON -- and the
// LOWEST triangle index wins a cell on a shared edge (deterministic; the interpolant is continuous there);
// (2) per cell: c_j = sum_k T[j][k] (x_k - r_k), c_2 = 1 - c_0 - c_1 from the triangulation's affine transform,
// east / north = sum_j c_j value[vertex_j] in scipy's order, then the u/v recipe of simulator.py:778-792.  Cells
// outside the hull are NaN, as griddata's fill value.
__global__ __launch_bounds__(kBlock) void k_tri_owner(const double *__restrict__ pts, const int32_t *__restrict__ tri,
                                                     const double *__restrict__ transform, int ntri, double cell,
                                                     int rows, int cols, int32_t *__restrict__ owner)
{
    const int t = blockIdx.x;
    if (t >= ntri) return;
    const int32_t v0 = tri[3 * t], v1 = tri[3 * t + 1], v2 = tri[3 * t + 2];
    const double x0 = pts[2 * v0], y0 = pts[2 * v0 + 1], x1 = pts[2 * v1], y1 = pts[2 * v1 + 1], x2 = pts[2 * v2], y2 = pts[2 * v2 + 1];
    const double xmin = fmin(x0, fmin(x1, x2)), xmax = fmax(x0, fmax(x1, x2));
    const double ymin = fmin(y0, fmin(y1, y2)), ymax = fmax(y0, fmax(y1, y2));
    // cells whose centre (c cell, r cell) may lie in the box, one cell of margin for the tolerance
    long long c_lo = static_cast<long long>(floor(xmin / cell)) - 1, c_hi = static_cast<long long>(ceil(xmax / cell)) + 1;
    long long r_lo = static_cast<long long>(floor(ymin / cell)) - 1, r_hi = static_cast<long long>(ceil(ymax / cell)) + 1;
    c_lo = c_lo < 0 ? 0 : c_lo;  r_lo = r_lo < 0 ? 0 : r_lo;
    c_hi = c_hi > cols - 1 ? cols - 1 : c_hi;  r_hi = r_hi > rows - 1 ? rows - 1 : r_hi;
    if (c_lo > c_hi || r_lo > r_hi) return;
    const double *T = transform + 6 * static_cast<size_t>(t);        // [T00 T01; T10 T11; r0 r1]
    const double t00 = T[0], t01 = T[1], t10 = T[2], t11 = T[3], rx = T[4], ry = T[5];
    if (!(t00 == t00)) return;                                         // degenerate simplex (scipy: NaN transform)
    const double eps = 100.0 * 2.220446049250313e-16;
    const long long w = c_hi - c_lo + 1, n = w * (r_hi - r_lo + 1);
    for (long long q = static_cast<long long>(blockIdx.y) * kBlock + threadIdx.x; q < n; q += static_cast<long long>(gridDim.y) * kBlock) {
        const long long r = r_lo + q / w, c = c_lo + q % w;
        const double dx = static_cast<double>(c) * cell - rx, dy = static_cast<double>(r) * cell - ry;
        const double b0 = t00 * dx + t01 * dy, b1 = t10 * dx + t11 * dy, b2 = 1.0 - b0 - b1;
        const bool inside = b0 >= -eps && b0 <= 1.0 + eps && b1 >= -eps && b1 <= 1.0 + eps && b2 >= -eps && b2 <= 1.0 + eps;
        if (inside) atomicMin(&owner[static_cast<size_t>(r) * cols + static_cast<size_t>(c)], t);
    }
}

__global__ __launch_bounds__(kBlock) void k_wind_triangles(const int32_t *__restrict__ owner, const int32_t *__restrict__ tri,
                                                          const double *__restrict__ transform,
                                                          const double *__restrict__ east, const double *__restrict__ north,
                                                          int npts, double cell, double *__restrict__ wspeed,
                                                          double *__restrict__ wdirn, int rows, int cols, int batch)
{
    const size_t ncell = static_cast<size_t>(rows) * cols, total = ncell * batch;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * kBlock) {
        const int b = static_cast<int>(i / ncell);
        const size_t c0 = i - static_cast<size_t>(b) * ncell;
        const int r = static_cast<int>(c0 / cols), c = static_cast<int>(c0 % cols);
        const int32_t t = owner[c0];
        double spd = __longlong_as_double(0x7FF8000000000000ll), ang = spd;
        if (t != 0x7f7f7f7f) {
            const double *T = transform + 6 * static_cast<size_t>(t);
            const double dx = static_cast<double>(c) * cell - T[4], dy = static_cast<double>(r) * cell - T[5];
            const double b0 = T[0] * dx + T[1] * dy, b1 = T[2] * dx + T[3] * dy, b2 = 1.0 - b0 - b1;
            const double *e = east + static_cast<size_t>(b) * npts, *nn = north + static_cast<size_t>(b) * npts;
            const int32_t v0 = tri[3 * t], v1 = tri[3 * t + 1], v2 = tri[3 * t + 2];
            const double ee = b0 * e[v0] + b1 * e[v1] + b2 * e[v2];     // (left to right, as scipy sums them)
            const double en = b0 * nn[v0] + b1 * nn[v1] + b2 * nn[v2];
            spd = sqrt(ee * ee + en * en);
            ang = fmod(atan2(ee, en) + 2.0 * kPi, 2.0 * kPi) * 180.0 / kPi;
        }
        wspeed[i] = spd;
        wdirn[i] = ang;
    }
}

// Snapshot / seasonal K1: DEM tile in LDS -> Horn sums once per cell, then for every
// snapshot b the wind at the cell (bilinear in the lattice's east / north components,
// as k_wind_lattice) and the updraft in the trig-free form of k_updraft_from_dem: with
// (cos w, sin w) = (north, east) / speed the wind speed cancels,
//     w = -(Y north + X east) / sqrt(d^2 + X^2 + Y^2),
// so neither the per-cell wind rasters (16 B per cell and snapshot written and read
// again) nor atan2 / sin / cos per cell are needed: 8 B read per cell for the whole
// batch, 4-12 B written per cell and snapshot.
struct LatticeArgs {
    const double *east, *north;       // (batch, ny, nx)
    int nx, ny, batch;
    double x0, y0, inv_dx, inv_dy, cell;
};

template <typename Tin>
__global__ __launch_bounds__(kBlock) void k_updraft_from_dem_lattice(
    const Tin *__restrict__ dem, FusedArgs fa, LatticeArgs la, float *__restrict__ orograph,
    double *__restrict__ usable, int rows, int cols, int tiles_x, int ntiles)
{
    __shared__ double tile[LW * LH];
    const int t = xcd_tile(blockIdx.x, ntiles);
    const int r0 = (t / tiles_x) * TH, c0 = (t % tiles_x) * TW;
    stage_dem_tile(dem, rows, cols, r0, c0, tile);
    const int lc = threadIdx.x % TW + 1;
    const int c = c0 + lc - 1;
    if (c >= cols) return;
    constexpr int kRowsPerWave = TH / (kBlock / TW);
    const int lr0 = (threadIdx.x / TW) * kRowsPerWave + 1;
    // lattice column of this thread's cells (the same for all its rows)
    double fx = (c * la.cell - la.x0) * la.inv_dx;
    fx = fx < 0.0 ? 0.0 : (fx > la.nx - 1.0 ? la.nx - 1.0 : fx);
    int ix = static_cast<int>(fx);
    ix = ix > la.nx - 2 ? (la.nx > 1 ? la.nx - 2 : 0) : ix;
    const double tx = la.nx > 1 ? fx - ix : 0.0;
    const int jx1 = la.nx > 1 ? 1 : 0;
    const size_t ncell = static_cast<size_t>(rows) * cols;
    for (int lr = lr0; lr < lr0 + kRowsPerWave; ++lr) {
        const int r = r0 + lr - 1;
        if (r >= rows) break;
        const double *m = tile + (lr - 1) * LW + lc, *z = tile + lr * LW + lc, *p = tile + (lr + 1) * LW + lc;
        const bool interior = r > 0 && c > 0 && r < rows - 1 && c < cols - 1;
        // un-normalised Horn sums (layers.py:78-90; "x" = row axis)
        const double X = (p[1] + 2 * p[0] + p[-1]) - (m[1] + 2 * m[0] + m[-1]);
        const double Y = (m[1] + 2 * z[1] + p[1]) - (m[-1] + 2 * z[-1] + p[-1]);
        const double rs = rsqrt_pos(fa.d2 + (X * X + Y * Y));
        double fy = (r * la.cell - la.y0) * la.inv_dy;
        fy = fy < 0.0 ? 0.0 : (fy > la.ny - 1.0 ? la.ny - 1.0 : fy);
        int iy = static_cast<int>(fy);
        iy = iy > la.ny - 2 ? (la.ny > 1 ? la.ny - 2 : 0) : iy;
        const double ty = la.ny > 1 ? fy - iy : 0.0;
        const size_t o00 = static_cast<size_t>(iy) * la.nx + ix;
        const size_t o10 = o00 + (la.ny > 1 ? la.nx : 0);
        const double w00 = (1.0 - tx) * (1.0 - ty), w01 = tx * (1.0 - ty), w10 = (1.0 - tx) * ty, w11 = tx * ty;
        const size_t i = static_cast<size_t>(r) * cols + c;
        for (int b = 0; b < la.batch; ++b) {
            const double *le = la.east + static_cast<size_t>(b) * la.nx * la.ny;
            const double *ln = la.north + static_cast<size_t>(b) * la.nx * la.ny;
            // same accumulation order as k_wind_lattice
            double east = 0.0, north = 0.0;
            east += w00 * le[o00]; north += w00 * ln[o00];
            east += w01 * le[o00 + jx1]; north += w01 * ln[o00 + jx1];
            east += w10 * le[o10]; north += w10 * ln[o10];
            east += w11 * le[o10 + jx1]; north += w11 * ln[o10 + jx1];
            double w = 0.0;
            if (interior) {
                if (X != 0.0) {
                    const double P = -(Y * north + X * east);
                    if (P > 0.0) w = P * rs;
                } else {
                    // dz_dx == 0: the reference substitutes 1e-10 for the aspect only (layers.py:124)
                    const double spd = sqrt(east * east + north * north);
                    const double dzdy = Y / fa.d, dx = 1e-10;
                    const double g2 = dzdy * dzdy, gp2 = dx * dx + g2;
                    const double proj = spd > 0.0 ? -(dzdy * north + dx * east) : 0.0;
                    if (proj > 0.0 && g2 > 0.0) w = proj * sqrt(g2 / (gp2 * (1.0 + g2)));
                }
            }
            w = w > fa.min_val ? w : fa.min_val;
            const float w32 = static_cast<float>(w);
            if (orograph) orograph[b * ncell + i] = w32;
            if (usable)
                usable[b * ncell + i] = usable_updraft_fast(static_cast<double>(w32), fa.thr, fa.inv_thr, fa.scale);
        }
    }
}

}  // namespace ssrs

extern "C" size_t ssrs_lattice_workspace_bytes(int nx, int ny, int batch)
{
    if (nx <= 0 || ny <= 0 || batch <= 0) return 0;
    return static_cast<size_t>(nx) * ny * batch * 2 * sizeof(double);
}

extern "C" int ssrs_updraft_from_dem_lattice(const void *dem, int dem_type, double res,
                                             const double *lattice_speed, const double *lattice_dirn,
                                             int nx, int ny, double x0, double y0, double dx, double dy,
                                             double min_val, float *orograph, double threshold,
                                             double *usable, int rows, int cols, int batch,
                                             void *workspace, size_t workspace_bytes, void *stream)
{
    SSRS_REQUIRE(dem && lattice_speed && lattice_dirn, "ssrs_updraft_from_dem_lattice: NULL pointer");
    SSRS_REQUIRE(rows >= 3 && cols >= 3 && nx >= 1 && ny >= 1 && batch >= 1,
                 "ssrs_updraft_from_dem_lattice: bad sizes");
    SSRS_REQUIRE(res > 0.0 && dx > 0.0 && dy > 0.0, "ssrs_updraft_from_dem_lattice: res, dx, dy must be > 0");
    SSRS_REQUIRE(dem_type == SSRS_F32 || dem_type == SSRS_F64, "ssrs_updraft_from_dem_lattice: bad element type");
    SSRS_REQUIRE(!(usable && !(threshold > 0.0)),
                 "ssrs_updraft_from_dem_lattice: usable requested without a positive threshold");
    SSRS_REQUIRE(workspace && workspace_bytes >= ssrs_lattice_workspace_bytes(nx, ny, batch),
                 "ssrs_updraft_from_dem_lattice: workspace too small");
    if (!orograph && !usable) return SSRS_OK;
    hipStream_t st = as_stream(stream);
    const size_t npts = static_cast<size_t>(nx) * ny * batch;
    double *east = static_cast<double *>(workspace), *north = east + npts;
    hipLaunchKernelGGL(k_lattice_components, dim3(stream_grid(npts)), dim3(kBlock), 0, st, lattice_speed,
                       lattice_dirn, npts, east, north);
    const int tx = (cols + TW - 1) / TW, ty = (rows + TH - 1) / TH, nt = tx * ty;
    FusedArgs fa = {};
    fa.d = 8 * res;
    fa.d2 = fa.d * fa.d;
    fa.min_val = min_val;
    fa.thr = threshold;
    fa.inv_thr = threshold > 0.0 ? 1.0 / threshold : 0.0;
    fa.scale = threshold > 0.0 ? threshold / (exp(1.0) - 1.0) : 0.0;
    LatticeArgs la = {east, north, nx, ny, batch, x0, y0, 1.0 / dx, 1.0 / dy, res / 1000.0};
    if (dem_type == SSRS_F64)
        hipLaunchKernelGGL((k_updraft_from_dem_lattice<double>), dim3(nt), dim3(kBlock), 0, st,
                           static_cast<const double *>(dem), fa, la, orograph, usable, rows, cols, tx, nt);
    else
        hipLaunchKernelGGL((k_updraft_from_dem_lattice<float>), dim3(nt), dim3(kBlock), 0, st,
                           static_cast<const float *>(dem), fa, la, orograph, usable, rows, cols, tx, nt);
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}

extern "C" int ssrs_wind_from_lattice(const double *lattice_speed, const double *lattice_dirn,
                                      int nx, int ny, double x0, double y0, double dx, double dy,
                                      double cell_size, double *wspeed, double *wdirn, int rows,
                                      int cols, int batch, void *stream)
{
    SSRS_REQUIRE(lattice_speed && lattice_dirn && wspeed && wdirn,
                 "ssrs_wind_from_lattice: NULL pointer");
    SSRS_REQUIRE(nx >= 1 && ny >= 1 && rows > 0 && cols > 0 && batch > 0,
                 "ssrs_wind_from_lattice: bad sizes");
    SSRS_REQUIRE(dx > 0.0 && dy > 0.0 && cell_size > 0.0,
                 "ssrs_wind_from_lattice: spacings must be > 0");
    const size_t total = static_cast<size_t>(rows) * cols * batch;
    hipLaunchKernelGGL(k_wind_lattice, dim3(stream_grid(total)), dim3(kBlock), 0,
                       as_stream(stream), lattice_speed, lattice_dirn, nx, ny, x0, y0, dx, dy,
                       cell_size, wspeed, wdirn, rows, cols, batch);
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}

extern "C" size_t ssrs_wind_triangles_workspace_bytes(int npts, int rows, int cols, int batch)
{
    if (npts <= 0 || rows <= 0 || cols <= 0 || batch <= 0) return 0;
    return static_cast<size_t>(rows) * cols * sizeof(int32_t) + 2 * static_cast<size_t>(npts) * batch * sizeof(double) + 512;
}

extern "C" int ssrs_wind_from_triangles(const double *points, const int32_t *triangles, const double *transform,
                                        const double *speed, const double *dirn, int npts, int ntri, double cell_size,
                                        double *wspeed, double *wdirn, int rows, int cols, int batch,
                                        void *workspace, size_t workspace_bytes, void *stream)
{
    SSRS_REQUIRE(points && triangles && transform && speed && dirn && wspeed && wdirn,
                 "ssrs_wind_from_triangles: NULL pointer");
    SSRS_REQUIRE(npts >= 3 && ntri >= 1 && rows > 0 && cols > 0 && batch > 0, "ssrs_wind_from_triangles: bad sizes");
    SSRS_REQUIRE(cell_size > 0.0, "ssrs_wind_from_triangles: cell_size must be > 0");
    SSRS_REQUIRE(workspace && workspace_bytes >= ssrs_wind_triangles_workspace_bytes(npts, rows, cols, batch),
                 "ssrs_wind_from_triangles: workspace too small");
    hipStream_t st = as_stream(stream);
    const size_t ncell = static_cast<size_t>(rows) * cols, nval = static_cast<size_t>(npts) * batch;
    int32_t *owner = static_cast<int32_t *>(workspace);
    double *east = reinterpret_cast<double *>(static_cast<char *>(workspace) + ((ncell * sizeof(int32_t) + 255) / 256) * 256);
    double *north = east + nval;
    SSRS_HIP_CHECK(hipMemsetAsync(owner, 0x7f, ncell * sizeof(int32_t), st));      // 0x7f7f7f7f: above any index
    hipLaunchKernelGGL(k_lattice_components, dim3(stream_grid(nval)), dim3(kBlock), 0, st, speed, dirn, nval, east, north);
    hipLaunchKernelGGL(k_tri_owner, dim3(static_cast<unsigned>(ntri), 64), dim3(kBlock), 0, st, points, triangles, transform, ntri,
                       cell_size, rows, cols, owner);
    hipLaunchKernelGGL(k_wind_triangles, dim3(stream_grid(ncell * batch)), dim3(kBlock), 0, st, owner, triangles, transform, east,
                       north, npts, cell_size, wspeed, wdirn, rows, cols, batch);
    SSRS_HIP_CHECK(hipGetLastError());
    return SSRS_OK;
}
