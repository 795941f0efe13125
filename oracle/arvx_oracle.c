/*
 * arvx_oracle.c -- CPU restatement of the AR_Voxel_Project carving hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see arvx_oracle.h).  PARITY PARTLY PINNED: marching cubes,
 * surface selection and closure geometry against the reference's own 1.off, the face-colour
 * rule against its 2.off / 3.off; the cv::gemm / cv::norm arithmetic of the carve path is
 * unpinned (the reference cannot be built here).
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference checkout).  Two third-party calls sit on the path; their
 * arithmetic is restated from the published OpenCV 4.x sources
 * (modules/core/src/matmul.dispatch.cpp, matmul.simd.hpp, and
 * include/opencv2/core/base.hpp), version unpinned by the reference
 * (CMakeLists.txt:16; README badge 4.6.0):
 *
 *  (G1) `intr * pose` (3x3 * 3x4, CV_32F): cv::gemm's small-matrix branch
 *       (flags==0, 2<=len<=4, len==d_size.height).  Each output is
 *       t = a0*b0 + a1*b1 + a2*b2 evaluated in float, left to right, then
 *       (float)(t*alpha + c*beta) with alpha=1.0, c=0, beta=0 in double,
 *       which returns t unchanged.  Unfused here (-ffp-contract=off); an
 *       OpenCV build whose AVX2 dispatch contracts to FMA can differ in the
 *       last ulp, so the product's C-ABI takes M itself.
 *  (G2) `M * world` (3x4 * 4x1, CV_32F): len==4 matches neither d_size
 *       dimension, so the generic GEMMSingleMul<float,double> runs; with
 *       d_size.width==1 and B continuous gemm turns the product into the A*Bt
 *       branch, 4-way unrolled: s0..s3 (double) each hold one exact product
 *       M[r][k]*w[k], and the row result is float(s*alpha), alpha = 1.0, where
 *       s is the sum of the four accumulators.  HOW they are summed is a
 *       recollection of OpenCV's source (none in this image):
 *         LEFT  (default)  `s0 = (s0+s1+s2+s3)*alpha`  = ((p0+p1)+p2)+p3
 *         RIGHT            `s0 += s1 + s2 + s3`        = p0+((p1+p2)+p3)
 *       SURVEY.md 8(c)(3) recalls the second form; the reviewer of round 2 and
 *       a second reading of the 4.x source (matmul.simd.hpp, "A * Bt" branch)
 *       recall the first, which is also what the loop gives when
 *       CV_ENABLE_UNROLLED is off -- so LEFT is the default, here and in the
 *       kernels, and the other one is a run-time switch on both sides
 *       (arvx_oracle_set_assoc; kernels: arvx_ctx_set_projection_assoc).
 *       tests/test_assoc_*.py hold a voxel on which the two groupings give
 *       different pixels and check kernel == oracle under each; on a host with
 *       OpenCV include/arvx/opencv_dropin.hpp / tools/pin_with_opencv.py settle
 *       it with one cv::gemm call.  The two differ only through double
 *       rounding (about one voxel-view in 1e8 on ordinary scenes).
 *  (N1) cv::norm(Vec4f) = sqrt(normL2Sqr<float,double>): one 4-way unrolled
 *       step, s += v0*v0 + v1*v1 + v2*v2 + v3*v3 in double, s starting at 0.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include "arvx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- reference constants: src/Model.h:90-91 ---------------------------- */
static const float MODEL_COLOR[4] = {50.f, 168.f, 141.f, 1.f};
static const float UNSEEN_COLOR[4] = {204.f, 0.f, 0.f, 1.f};

static inline long flatten(int X, int Y, int x, int y, int z) {
    /* src/Model.h:104-106 */
    return (long)x + (long)X * ((long)y + (long)Y * (long)z);
}

/* (G1)  src/VoxelCarving.cpp:19 `intr * pose` */
void arvx_oracle_compose(const float K[9], const float Rt[12], float M[12]) {
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 4; ++c) {
            float t = K[r * 3 + 0] * Rt[0 * 4 + c];
            t = t + K[r * 3 + 1] * Rt[1 * 4 + c];
            t = t + K[r * 3 + 2] * Rt[2 * 4 + c];
            M[r * 4 + c] = (float)((double)t * 1.0 + 0.0 * 0.0);
        }
    }
}

/* src/Model.h:134-140  toWord: (y*s, x*s, -1*z*s, 1) -- x/y swapped, z negated */
static inline void to_word(float s, int x, int y, int z, float w[4]) {
    w[0] = (float)y * s;
    w[1] = (float)x * s;
    w[2] = (float)(-1 * z) * s;
    w[3] = 1.f;
}

/* grouping of (G2): 1 = ((p0+p1)+p2)+p3 (default), 0 = p0+((p1+p2)+p3).  Set between calls
 * only (the worker threads of a call read it). */
static int g_assoc_left = 1;

/* (G2)  src/VoxelCarving.cpp:18-21 worldToCamera, the `* world` factor */
static inline void mat_vec(const float M[12], const float w[4], float proj[3]) {
    for (int r = 0; r < 3; ++r) {
        double p0 = (double)M[r * 4 + 0] * (double)w[0];
        double p1 = (double)M[r * 4 + 1] * (double)w[1];
        double p2 = (double)M[r * 4 + 2] * (double)w[2];
        double p3 = (double)M[r * 4 + 3] * (double)w[3];
        double s0 = g_assoc_left ? ((p0 + p1) + p2) + p3   /* s0 = (s0+s1+s2+s3)*alpha */
                                 : p0 + ((p1 + p2) + p3); /* s0 += s1 + s2 + s3 */
        s0 = s0 * 1.0; /* alpha */
        proj[r] = (float)s0;
    }
}

/* which grouping of (G2) is in force: 1 = ((p0+p1)+p2)+p3 (default), 0 = p0+((p1+p2)+p3) */
int arvx_oracle_assoc(void) { return g_assoc_left; }
void arvx_oracle_set_assoc(int left) { g_assoc_left = left ? 1 : 0; }

void arvx_oracle_project_raw(const float M[12], float s, int x, int y, int z,
                             float out[5]) {
    float w[4], proj[3];
    to_word(s, x, y, z, w);
    mat_vec(M, w, proj);
    out[0] = proj[0];
    out[1] = proj[1];
    out[2] = proj[2];
    out[3] = proj[0] / proj[2]; /* src/VoxelCarving.cpp:20, IEEE f32 divide */
    out[4] = proj[1] / proj[2];
}

/* src/VoxelCarving.cpp:44 `(int)std::round(u)`.  The cast is undefined for
 * non-finite or out-of-int-range values; x86 cvttss2si yields INT_MIN there,
 * which Rect::contains rejects -- restated as "outside". */
static inline int round_to_pixel(float u, int *ok) {
    float r = roundf(u);
    if (!(r >= -2147483648.0f && r < 2147483648.0f)) { /* also false for NaN */
        *ok = 0;
        return 0;
    }
    *ok = 1;
    return (int)r;
}

int arvx_oracle_project(const float M[12], float s, int x, int y, int z,
                        int W, int H, int *px, int *py) {
    float o[5];
    arvx_oracle_project_raw(M, s, x, y, z, o);
    int oku, okv;
    int iu = round_to_pixel(o[3], &oku);
    int iv = round_to_pixel(o[4], &okv);
    if (!oku || !okv) return 0;
    /* src/VoxelCarving.cpp:45 Point::inside(Rect(0,0,W,H)) */
    if (iu < 0 || iu >= W || iv < 0 || iv >= H) return 0;
    *px = iu;
    *py = iv;
    return 1;
}

static inline int mask_is_background(const uint8_t *mask, int C, long stride,
                                     int px, int py) {
    /* src/VoxelCarving.cpp:49-50: all channel bytes == 0 */
    const uint8_t *p = mask + (long)py * stride + (long)px * C;
    for (int c = 0; c < C; ++c)
        if (p[c] != 0) return 0;
    return 1;
}

/* src/VoxelCarving.cpp:38-55 -- loop order of for_each_voxel (src/Model.h:10-35):
 * x outer, y, z inner.  Order does not change the result. */
void arvx_oracle_carve_view(int X, int Y, int Z, float s, const float M[12],
                            const uint8_t *mask, int W, int H, int C,
                            long stride, uint8_t *state) {
    for (int x = 0; x < X; ++x)
        for (int y = 0; y < Y; ++y)
            for (int z = 0; z < Z; ++z) {
                int px, py;
                if (!arvx_oracle_project(M, s, x, y, z, W, H, &px, &py))
                    continue; /* :45-48 -- not seen */
                long i = flatten(X, Y, x, y, z);
                if (mask_is_background(mask, C, stride, px, py))
                    state[i] &= (uint8_t)~ARVX_ORACLE_OCC; /* :50-53 set(0,0,0,0) */
                state[i] |= ARVX_ORACLE_SEEN;              /* :54 see() */
            }
}

/* src/VoxelCarving.cpp:60-72 */
void arvx_oracle_carve(int X, int Y, int Z, float s, int V, const float *M,
                       const uint8_t *masks, int W, int H, int C, long stride,
                       uint8_t *state) {
    for (int i = 0; i < V; ++i)
        arvx_oracle_carve_view(X, Y, Z, s, M + 12 * i,
                               masks + (long)i * H * stride, W, H, C, stride,
                               state);
}

void arvx_oracle_carve_mt(int X, int Y, int Z, float s, int V, const float *M,
                          const uint8_t *masks, int W, int H, int C,
                          long stride, uint8_t *state, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel for collapse(2) schedule(static)
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) {
                long i = flatten(X, Y, x, y, z);
                uint8_t st = state[i];
                for (int v = 0; v < V; ++v) {
                    int px, py;
                    if (!arvx_oracle_project(M + 12 * v, s, x, y, z, W, H, &px,
                                             &py))
                        continue;
                    if (mask_is_background(masks + (long)v * H * stride, C,
                                           stride, px, py))
                        st &= (uint8_t)~ARVX_ORACLE_OCC;
                    st |= ARVX_ORACLE_SEEN;
                }
                state[i] = st;
            }
}

/* The same carve on a list of z planes only (a Z slab or the striped planes of one rank of
 * the multi-GPU split): state holds plane zlist[k] at k*X*Y.  src/VoxelCarving.cpp:38-55. */
void arvx_oracle_carve_planes_mt(int X, int Y, float s, int V, const float *M,
                                 const uint8_t *masks, int W, int H, int C, long stride,
                                 const int32_t *zlist, int nz, uint8_t *state, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nz; ++k)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) {
                const int z = zlist[k];
                long i = flatten(X, Y, x, y, k);
                uint8_t st = state[i];
                for (int v = 0; v < V; ++v) {
                    int px, py;
                    if (!arvx_oracle_project(M + 12 * v, s, x, y, z, W, H, &px, &py)) continue;
                    if (mask_is_background(masks + (long)v * H * stride, C, stride, px, py))
                        st &= (uint8_t)~ARVX_ORACLE_OCC;
                    st |= ARVX_ORACLE_SEEN;
                }
                state[i] = st;
            }
}

/* Reference-shaped CPU baseline (timing only): src/VoxelCarving.cpp:23-58
 * with its per-voxel `intr * pose * world`, AoS Vector4f voxels
 * (src/Model.h:100) and bit-packed seen (std::vector<bool>, :102).
 * rgba / seen_bits hold planes zlo..zhi-1 only (a bounded sample of the grid). */
void arvx_oracle_carve_ref(int X, int Y, int Z, float s, int V,
                           const float K[9], const float *Rt,
                           const uint8_t *masks, int W, int H, int C,
                           long stride, float *rgba, uint64_t *seen_bits,
                           int zlo, int zhi) {
    (void)Z;
    for (int v = 0; v < V; ++v) {
        const float *rt = Rt + 12 * v;
        const uint8_t *mask = masks + (long)v * H * stride;
        for (int x = 0; x < X; ++x)
            for (int y = 0; y < Y; ++y)
                for (int z = zlo; z < zhi; ++z) {
                    float M[12];
                    arvx_oracle_compose(K, rt, M); /* recomputed per voxel */
                    int px, py;
                    if (!arvx_oracle_project(M, s, x, y, z, W, H, &px, &py))
                        continue;
                    long i = flatten(X, Y, x, y, z - zlo);
                    if (mask_is_background(mask, C, stride, px, py)) {
                        rgba[4 * i + 0] = 0.f;
                        rgba[4 * i + 1] = 0.f;
                        rgba[4 * i + 2] = 0.f;
                        rgba[4 * i + 3] = 0.f;
                    }
                    seen_bits[i >> 6] |= (uint64_t)1 << (i & 63);
                }
    }
}

/* src/VoxelCarving.cpp:74-167 fastCarve.  visited()==seen (src/Model.h:154-160). */
void arvx_oracle_fast_carve(int X, int Y, int Z, float s, int V,
                            const float *M, const uint8_t *masks, int W,
                            int H, int C, long stride, uint8_t *state) {
    long N = (long)X * Y * Z;
    /* std::queue<Vec3i>: a voxel can be pushed by several neighbours before it
     * is popped, at most 6 times, plus the seed. */
    long cap = 6 * N + 8;
    int32_t *q = (int32_t *)malloc((size_t)cap * 3 * sizeof(int32_t));
    long head = 0, tail = 0;
    q[0] = q[1] = q[2] = 0;
    tail = 1;
    while (head < tail) {
        int cx = q[3 * head], cy = q[3 * head + 1], cz = q[3 * head + 2];
        ++head;
        long i = flatten(X, Y, cx, cy, cz);
        if (state[i] & ARVX_ORACLE_SEEN) continue; /* :105-107 */
        state[i] |= ARVX_ORACLE_SEEN;              /* :108 visit */
        int carved = 0;
        for (int v = 0; v < V; ++v) { /* :111-130 */
            int px, py;
            if (!arvx_oracle_project(M + 12 * v, s, cx, cy, cz, W, H, &px, &py))
                continue;
            if (mask_is_background(masks + (long)v * H * stride, C, stride, px,
                                   py)) {
                state[i] &= (uint8_t)~ARVX_ORACLE_OCC;
                carved = 1;
                break;
            }
        }
        if (!carved) continue;
        /* :132-163 push unvisited 6-neighbours */
        static const int d[6][3] = {{-1, 0, 0}, {1, 0, 0},  {0, -1, 0},
                                    {0, 1, 0},  {0, 0, -1}, {0, 0, 1}};
        for (int k = 0; k < 6; ++k) {
            int nx = cx + d[k][0], ny = cy + d[k][1], nz = cz + d[k][2];
            if (nx < 0 || nx >= X || ny < 0 || ny >= Y || nz < 0 || nz >= Z)
                continue;
            if (state[flatten(X, Y, nx, ny, nz)] & ARVX_ORACLE_SEEN) continue;
            q[3 * tail] = nx;
            q[3 * tail + 1] = ny;
            q[3 * tail + 2] = nz;
            ++tail;
        }
    }
    free(q);
}

/* src/Model.cpp:9-14 */
void arvx_oracle_model_init(float *rgba, long N) {
    for (long i = 0; i < N; ++i) memcpy(rgba + 4 * i, MODEL_COLOR, 16);
}

/* carve writes (0,0,0,0), src/VoxelCarving.cpp:52; untouched voxels keep
 * MODEL_COLOR.  Rebuilds the AoS array a fresh Model would hold after carve. */
void arvx_oracle_state_to_model(const uint8_t *state, float *rgba, long N) {
    for (long i = 0; i < N; ++i) {
        if (state[i] & ARVX_ORACLE_OCC)
            memcpy(rgba + 4 * i, MODEL_COLOR, 16);
        else
            memset(rgba + 4 * i, 0, 16);
    }
}

void arvx_oracle_model_to_state(const float *rgba, uint8_t *state, long N) {
    for (long i = 0; i < N; ++i)
        state[i] = (uint8_t)((state[i] & ARVX_ORACLE_SEEN) |
                             (rgba[4 * i + 3] != 0.f ? ARVX_ORACLE_OCC : 0));
}

/* src/Model.cpp:36-47 */
void arvx_oracle_handle_unseen(const uint8_t *state, float *rgba, long N) {
    for (long i = 0; i < N; ++i)
        if (!(state[i] & ARVX_ORACLE_SEEN)) memcpy(rgba + 4 * i, UNSEEN_COLOR, 16);
}

/* src/Model.h:119-124 get(): zero outside the grid */
static inline float get_w(const float *rgba, int X, int Y, int Z, int x, int y,
                          int z) {
    if (x < 0 || x >= X || y < 0 || y >= Y || z < 0 || z >= Z) return 0.f;
    return rgba[4 * flatten(X, Y, x, y, z) + 3];
}

/* src/Model.h:126-132 */
static inline int is_inner(const float *rgba, int X, int Y, int Z, int x,
                           int y, int z) {
    return get_w(rgba, X, Y, Z, x - 1, y, z) != 0 &&
           get_w(rgba, X, Y, Z, x + 1, y, z) != 0 &&
           get_w(rgba, X, Y, Z, x, y - 1, z) != 0 &&
           get_w(rgba, X, Y, Z, x, y + 1, z) != 0 &&
           get_w(rgba, X, Y, Z, x, y, z - 1) != 0 &&
           get_w(rgba, X, Y, Z, x, y, z + 1) != 0;
}

/* (N1) src/ColorReconstruction.h:59  cv::norm(cameras[i] - word_coord) */
float arvx_oracle_depth(const float campos[3], float s, int x, int y, int z) {
    float w[4];
    to_word(s, x, y, z, w);
    float d0 = campos[0] - w[0];
    float d1 = campos[1] - w[1];
    float d2 = campos[2] - w[2];
    float d3 = 1.f - w[3]; /* cameras[i] = (tx,ty,tz,1), :21 */
    double v0 = d0, v1 = d1, v2 = d2, v3 = d3;
    double acc = 0.0;
    acc += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    return (float)sqrt(acc); /* double -> float at addColor(), src/Model.h:142 */
}

/* src/ColorReconstruction.h:34-74 (voxel_pass_loops) +
 * src/ColorReconstruction.cpp:26-42 (closest) / :52-66 (average). */
void arvx_oracle_color(int X, int Y, int Z, float s, int V, const float *M,
                       const float *campos, const uint8_t *images, int W,
                       int H, long stride, int mode, float *rgba) {
    for (int x = 0; x < X; ++x)
        for (int y = 0; y < Y; ++y)
            for (int z = 0; z < Z; ++z) {
                long i = flatten(X, Y, x, y, z);
                if (rgba[4 * i + 3] == 0.f || is_inner(rgba, X, Y, Z, x, y, z))
                    continue; /* .h:46-49 */
                int n = 0;
                float sum[4] = {0.f, 0.f, 0.f, 1.f}; /* .cpp:59 */
                float best[4] = {0.f, 0.f, 0.f, 0.f};
                float best_depth = 0.f;
                for (int v = 0; v < V; ++v) { /* .h:50-60 */
                    int px, py;
                    if (!arvx_oracle_project(M + 12 * v, s, x, y, z, W, H, &px,
                                             &py))
                        continue;
                    const uint8_t *p = images + (long)v * H * stride +
                                       (long)py * stride + (long)px * 3;
                    float col[4] = {(float)p[2], (float)p[1], (float)p[0], 1.f};
                    float depth = arvx_oracle_depth(campos + 3 * v, s, x, y, z);
                    if (n == 0 || depth < best_depth) { /* .cpp:33-40, strict < */
                        memcpy(best, col, 16);
                        best_depth = depth;
                    }
                    for (int c = 0; c < 4; ++c) sum[c] = sum[c] + col[c];
                    ++n;
                }
                if (n == 0) continue; /* .cpp:29-31 / :55-57 */
                if (mode == 0) {
                    memcpy(rgba + 4 * i, best, 16); /* .cpp:41 */
                } else {
                    float fn = (float)n; /* Eigen Vector4f / size_t -> float */
                    rgba[4 * i + 0] = roundf(sum[0] / fn); /* .cpp:64-65 */
                    rgba[4 * i + 1] = roundf(sum[1] / fn);
                    rgba[4 * i + 2] = roundf(sum[2] / fn);
                    rgba[4 * i + 3] = 1.f;
                }
            }
}

/* src/Postprocessing3d.cpp:4-100, kernelSize = 3, restated literally: a
 * dilation pass into `temp`, then the erosion pass whose tests `w < thresh`
 * (thresh = 0) can only fire for negative w. */
void arvx_oracle_closure(int X, int Y, int Z, float *rgba) {
    const float thresh = 0.f;
    const int size = 1;
    long N = (long)X * Y * Z;
    float *temp = (float *)malloc((size_t)N * 16);
    for (int x = 0; x < X; ++x)
        for (int y = 0; y < Y; ++y)
            for (int z = 0; z < Z; ++z) {
                long i = flatten(X, Y, x, y, z);
                if (rgba[4 * i + 3] > thresh) { /* :23-27 */
                    memcpy(temp + 4 * i, rgba + 4 * i, 16);
                    continue;
                }
                int count = 0;
                float sum[4] = {0, 0, 0, 0};
                for (int a = -size; a <= size; ++a) {
                    int xn = x + a;
                    if (xn < 0 || xn >= X) continue;
                    for (int b = -size; b <= size; ++b) {
                        int yn = y + b;
                        if (yn < 0 || yn >= Y) continue;
                        for (int c = -size; c <= size; ++c) {
                            int zn = z + c;
                            if (zn < 0 || zn >= Z) continue;
                            const float *val = rgba + 4 * flatten(X, Y, xn, yn, zn);
                            if (val[3] > thresh) { /* :42-45 */
                                ++count;
                                for (int k = 0; k < 4; ++k) sum[k] = sum[k] + val[k];
                            }
                        }
                    }
                }
                if (count > 0) /* :49-51 Eigen `sum /= count` -> float(count) */
                    for (int k = 0; k < 4; ++k) sum[k] = sum[k] / (float)count;
                memcpy(temp + 4 * i, sum, 16);
            }
    for (int x = 0; x < X; ++x) /* :60-96 erosion */
        for (int y = 0; y < Y; ++y)
            for (int z = 0; z < Z; ++z) {
                long i = flatten(X, Y, x, y, z);
                if (rgba[4 * i + 3] < thresh) continue; /* :66-69 keeps value */
                int failed = 0;
                for (int a = -size; a <= size && !failed; ++a) {
                    int xn = x + a;
                    if (xn < 0 || xn >= X) continue;
                    for (int b = -size; b <= size && !failed; ++b) {
                        int yn = y + b;
                        if (yn < 0 || yn >= Y) continue;
                        for (int c = -size; c <= size && !failed; ++c) {
                            int zn = z + c;
                            if (zn < 0 || zn >= Z) continue;
                            if (temp[4 * flatten(X, Y, xn, yn, zn) + 3] < thresh)
                                failed = 1;
                        }
                    }
                }
                if (failed)
                    memset(rgba + 4 * i, 0, 16);
                else
                    memcpy(rgba + 4 * i, temp + 4 * i, 16);
            }
    free(temp);
}

/* Model::get, src/Model.h:119-124: w of a voxel, zero outside the grid. */
static float w_at(int X, int Y, int Z, const float *rgba, int x, int y, int z) {
    if (x < 0 || x >= X || y < 0 || y >= Y || z < 0 || z >= Z) return 0.f;
    return rgba[4 * ((size_t)x + (size_t)X * ((size_t)y + (size_t)Y * z)) + 3];
}

/* The cell walk of marchingCubes (src/MarchingCubes.cpp:12-18: x, then y, then
 * z, each from -1), the corner order of ProcessVoxel (src/MarchingCubes.h:537-552)
 * and the cube index of Polygonise (:479-484).  A cell whose edgeTable entry is
 * zero returns before it emits anything (:486-488); the table holds zero at
 * index 0 and 255 only. */
long arvx_oracle_mc_cells(int X, int Y, int Z, const float *rgba, float threshold,
                          int32_t *cells, long cap) {
    static const int corner[8][3] = {{1, 0, 0}, {0, 0, 0}, {0, 1, 0}, {1, 1, 0},
                                     {1, 0, 1}, {0, 0, 1}, {0, 1, 1}, {1, 1, 1}};
    long n = 0;
    for (int x = -1; x < X; x++)
        for (int y = -1; y < Y; y++)
            for (int z = -1; z < Z; z++) {
                int idx = 0;
                for (int i = 0; i < 8; i++)
                    if (w_at(X, Y, Z, rgba, x + corner[i][0], y + corner[i][1],
                             z + corner[i][2]) < threshold)
                        idx |= 1 << i;
                if (idx == 0 || idx == 255) continue;
                if (n < cap) {
                    cells[4 * n + 0] = x;
                    cells[4 * n + 1] = y;
                    cells[4 * n + 2] = z;
                    cells[4 * n + 3] = idx;
                }
                n++;
            }
    return n;
}

/* ---- marching cubes: src/MarchingCubes.cpp:8-31, src/MarchingCubes.h:414-578 ------------
 * Triangle table: Paul Bourke's (public domain), shared as data with the product
 * (include/arvx/mc_triangles.inc; the reference's copy: src/MarchingCubes.h:147-405).  The
 * edge table (:112-145) is not stored: an edge is cut iff its two corners differ. */
static const char *const MC_TRIANGLES[256] = {
#include "../include/arvx/mc_triangles.inc"
};
static const int MC_SECOND[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7}; /* :491 */

static int rgb_is_default(const float c[3]) { /* :454,458 MODEL_COLOR / UNSEEN_COLOR */
    return (c[0] == MODEL_COLOR[0] && c[1] == MODEL_COLOR[1] && c[2] == MODEL_COLOR[2]) ||
           (c[0] == UNSEEN_COLOR[0] && c[1] == UNSEEN_COLOR[1] && c[2] == UNSEEN_COLOR[2]);
}

/* VertexInterp, :428-468 */
static void vertex_interp(float threshold, const float p0[3], const float v0[4],
                          const float p1[3], const float v1[4], float coord[3],
                          float color[3]) {
    if (v0[3] == 0.0f && v1[3] != 0.0f) { /* :432-436 */
        memcpy(color, v1, 12);
        memcpy(coord, p1, 12);
        return;
    }
    if (v0[3] != 0.0f && v1[3] == 0.0f) { /* :437-441 */
        memcpy(color, v0, 12);
        memcpy(coord, p0, 12);
        return;
    }
    float f;
    if (v0[3] == v1[3]) f = 0.5f;
    else f = (threshold - v0[3]) / (v1[3] - v0[3]); /* :443-449 */
    const float g = 1 - f;
    for (int k = 0; k < 3; ++k) coord[k] = g * p0[k] + f * p1[k]; /* :450 (-ffp-contract=off) */
    if (rgb_is_default(v0)) memcpy(color, v1, 12);                 /* :454-457 */
    else if (rgb_is_default(v1)) memcpy(color, v0, 12);            /* :458-461 */
    else
        for (int k = 0; k < 3; ++k) color[k] = g * v0[k] + f * v1[k]; /* :464 */
}

static uint32_t mean_color_floats(float c1, float c2, float c3) { /* :414-416 */
    return (uint32_t)roundf((c1 + c2 + c3) / 3);
}

static int hexv(char c) { return c <= '9' ? c - '0' : c - 'a' + 10; }

/* The mesh marchingCubes() builds before it scales and writes it: 3 fresh vertices per
 * triangle (:561-568), vertex coordinates in voxel units; faces = (r, g, b) per triangle
 * (the vertex ids of triangle t are 3t, 3t+1, 3t+2).  Returns the number of triangles,
 * writes at most cap of them. */
long arvx_oracle_mc_mesh(int X, int Y, int Z, const float *rgba, float threshold,
                         float *verts, uint32_t *face_rgb, long cap) {
    static const int corner[8][3] = {{1, 0, 0}, {0, 0, 0}, {0, 1, 0}, {1, 1, 0},
                                     {1, 0, 1}, {0, 0, 1}, {0, 1, 1}, {1, 1, 1}}; /* :537-552 */
    static const float zero4[4] = {0.f, 0.f, 0.f, 0.f};
    long n = 0;
    for (int x = -1; x < X; x++)        /* src/MarchingCubes.cpp:12-18 */
        for (int y = -1; y < Y; y++)
            for (int z = -1; z < Z; z++) {
                const float *val[8];
                float p[8][3];
                int idx = 0;
                for (int i = 0; i < 8; i++) {
                    const int cx = x + corner[i][0], cy = y + corner[i][1],
                              cz = z + corner[i][2];
                    val[i] = (cx < 0 || cx >= X || cy < 0 || cy >= Y || cz < 0 || cz >= Z)
                                 ? zero4 /* Model::get, src/Model.h:119-124 */
                                 : rgba + 4 * flatten(X, Y, cx, cy, cz);
                    p[i][0] = (float)cx;
                    p[i][1] = (float)cy;
                    p[i][2] = (float)cz;
                    if (val[i][3] < threshold) idx |= 1 << i; /* :479-484 */
                }
                int edges = 0;
                for (int e = 0; e < 12; e++)
                    if (((idx >> (e % 8)) & 1) != ((idx >> MC_SECOND[e]) & 1)) edges |= 1 << e;
                if (edges == 0) continue; /* :486-488 */
                float coord[12][3], color[12][3];
                for (int e = 0; e < 12; e++) /* :493-497 */
                    if (edges & (1 << e))
                        vertex_interp(threshold, p[e % 8], val[e % 8], p[MC_SECOND[e]],
                                      val[MC_SECOND[e]], coord[e], color[e]);
                for (const char *t = MC_TRIANGLES[idx]; t[0]; t += 3) { /* :500-509 */
                    const int e0 = hexv(t[0]), e1 = hexv(t[1]), e2 = hexv(t[2]);
                    if (n < cap) {
                        memcpy(verts + 9 * n, coord[e0], 12);
                        memcpy(verts + 9 * n + 3, coord[e1], 12);
                        memcpy(verts + 9 * n + 6, coord[e2], 12);
                        /* col[2] = vertList[triTable[..][i + 1]].color: :506 */
                        const float *c0 = color[e0], *c1 = color[e1], *c2 = color[e1];
                        for (int k = 0; k < 3; k++) /* :570-573 */
                            face_rgb[3 * n + k] = mean_color_floats(c0[k], c1[k], c2[k]);
                    }
                    n++;
                }
            }
    return n;
}
