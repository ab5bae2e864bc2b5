/*
 * asan_driver.c — drives every entry point of oracle/splat_oracle.c under AddressSanitizer +
 * UndefinedBehaviorSanitizer (`make -C oracle asan` -> _build/orc_asan, run by
 * tests/test_oracle_sanitizers.py).  TEST INFRASTRUCTURE like the oracle itself.
 *
 * Every array is malloc'ed at EXACTLY the size the Python front-end (oracle/oracle.py) passes, so a
 * read or write one element past what the ctypes caller allocates is a heap-buffer-overflow report
 * here instead of silent corruption of a numpy heap there.  The scene is the S0 shape of
 * BASELINE.json's config 1 scaled by argv (default 10 000 Gaussians, 640x480): fwd + bwd with
 * precomputed colours (C = 4, 3-entry background: the reference's layout), fwd + bwd with SH degree 3
 * and a precomputed 3D covariance, both alpha modes, a row band, mark_visible, dist2 and exp2.
 * Prints one checksum line; exit code 0 means no sanitizer report.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc_settings {
    int32_t image_height, image_width;
    float tanfovx, tanfovy, scale_modifier;
    int32_t sh_degree, sh_coeffs, channels, bg_channels;
} orc_settings;

void orc_preprocess(const orc_settings*, int32_t, const float*, const float*, const float*, const float*, const float*,
                    const float*, const float*, const float*, const float*, int32_t*, float*, float*, float*, float*,
                    uint32_t*, float*, uint8_t*);
int64_t orc_bin(const orc_settings*, int32_t, const float*, const float*, const int32_t*, uint64_t*, uint32_t*, uint32_t*);
void orc_composite_fwd(const orc_settings*, const uint32_t*, const uint32_t*, const float*, const float*, const float*,
                       const float*, const float*, float*, float*, float*, float*, uint32_t*);
void orc_composite_bwd(const orc_settings*, const uint32_t*, const uint32_t*, const float*, const float*, const float*,
                       const float*, const float*, const float*, const float*, const float*, double*, double*, double*,
                       double*, double*);
void orc_preprocess_bwd(const orc_settings*, int32_t, const float*, const float*, const float*, const float*, const float*,
                        const float*, const float*, const float*, const int32_t*, const float*, const uint8_t*,
                        const double*, const double*, const double*, const double*, float*, float*, float*, float*,
                        float*, float*, double*, double*, double*);
void orc_mark_visible(int32_t, const float*, const float*, uint8_t*);
void orc_dist2(int32_t, const float*, float*);
void orc_exp2_array(int64_t, const float*, float*);
void orc_set_alpha_mode(int);
void orc_set_row_band(int, int);

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static float urand(void)
{ /* xorshift64*, 24 random bits -> [0, 1) */
    g_state ^= g_state >> 12;
    g_state ^= g_state << 25;
    g_state ^= g_state >> 27;
    return (float)((g_state * 0x2545F4914F6CDD1Dull) >> 40) * (1.0f / 16777216.0f);
}
static float nrand(void)
{
    const float u = urand() + 1e-7f, v = urand();
    return sqrtf(-2.0f * logf(u)) * cosf(6.2831853f * v);
}
static void* xm(size_t n, size_t sz)
{
    void* p = calloc(n ? n : 1, sz);   /* a zero-sized scene still hands out distinct, 1-element blocks */
    if (!p) { fprintf(stderr, "out of memory\n"); exit(2); }
    return p;
}

static double run_case(int P, int W, int H, int C, int bgc, int use_sh, int use_cov, int alpha_mode, int band)
{
    orc_settings st = {H, W, 1.0f, (float)H / (float)W, 1.0f, use_sh ? 3 : 0, use_sh ? 16 : 0, use_sh ? 3 : C, bgc};
    const int Cn = st.channels, gx = (W + 15) / 16, gy = (H + 15) / 16;
    float* means = xm((size_t)3 * P, 4); float* opac = xm(P, 4); float* scales = xm((size_t)3 * P, 4);
    float* rots = xm((size_t)4 * P, 4); float* cov = xm((size_t)6 * P, 4); float* feat = xm((size_t)P * C, 4);
    float* shs = xm((size_t)P * 48, 4);
    for (int i = 0; i < P; ++i) {
        const float z = 0.5f + 5.5f * urand();
        means[3 * i] = (2.f * urand() - 1.f) * 1.1f * z; means[3 * i + 1] = (2.f * urand() - 1.f) * 1.1f * z * st.tanfovy;
        means[3 * i + 2] = (i % 37 == 0) ? -z : z;   /* some behind the camera */
        opac[i] = 1.0f / (1.0f + expf(-1.5f * nrand()));
        float q[4], n2 = 0.f;
        for (int k = 0; k < 4; ++k) { q[k] = nrand(); n2 += q[k] * q[k]; }
        for (int k = 0; k < 4; ++k) rots[4 * i + k] = q[k] / sqrtf(n2 + 1e-12f);
        float L[9];
        for (int k = 0; k < 3; ++k) scales[3 * i + k] = expf(logf(0.02f) + 0.5f * nrand());
        for (int k = 0; k < 9; ++k) L[k] = 0.03f * nrand();
        cov[6 * i + 0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2]; cov[6 * i + 1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
        cov[6 * i + 2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8]; cov[6 * i + 3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
        cov[6 * i + 4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8]; cov[6 * i + 5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
        for (int k = 0; k < C; ++k) feat[(size_t)i * C + k] = urand();
        for (int k = 0; k < 48; ++k) shs[(size_t)i * 48 + k] = 0.5f * nrand();
    }
    /* identity pose, fx = fy = W/2, principal point ((W-1)/2, (H-1)/2), znear 0.01, zfar 100 (row-vector convention) */
    float V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, PM[16] = {0};
    const float zn = 0.01f, zf = 100.f;
    PM[0] = 1.0f / st.tanfovx; PM[5] = 1.0f / st.tanfovy; PM[8] = -1.0f / (float)W; PM[9] = -1.0f / (float)H;
    PM[10] = zf / (zf - zn); PM[11] = 1.0f; PM[14] = -(zf * zn) / (zf - zn);
    float campos[3] = {0, 0, 0}, bg[3] = {0.1f, 0.2f, 0.3f};

    int32_t* radii = xm(P, 4); float* xy = xm((size_t)2 * P, 4); float* depth = xm(P, 4); float* cov3 = xm((size_t)6 * P, 4);
    float* conop = xm((size_t)4 * P, 4); uint32_t* tt = xm(P, 4); float* rgb = xm((size_t)3 * P, 4); uint8_t* cl = xm((size_t)3 * P, 1);
    orc_set_alpha_mode(alpha_mode);
    if (band) orc_set_row_band(H / 4, H / 2); else orc_set_row_band(0, 0x7fffffff);
    orc_preprocess(&st, P, means, use_sh ? shs : NULL, opac, use_cov ? NULL : scales, use_cov ? NULL : rots,
                   use_cov ? cov : NULL, V, PM, campos, radii, xy, depth, cov3, conop, tt, rgb, cl);
    int64_t R = 0;
    for (int i = 0; i < P; ++i) R += tt[i];
    uint64_t* keys = xm((size_t)(R > 0 ? R : 1), 8); uint32_t* vals = xm((size_t)(R > 0 ? R : 1), 4);
    uint32_t* ranges = xm((size_t)2 * gx * gy, 4);
    const int64_t R2 = orc_bin(&st, P, xy, depth, radii, keys, vals, ranges);
    if (R2 != R) { fprintf(stderr, "orc_bin: %lld != %lld\n", (long long)R2, (long long)R); exit(3); }
    const float* f = use_sh ? rgb : feat;
    const size_t px = (size_t)H * W;
    float* oc = xm(px * Cn, 4); float* od = xm(px, 4); float* oa = xm(px, 4); float* fT = xm(px, 4); uint32_t* nc = xm(px, 4);
    orc_composite_fwd(&st, ranges, vals, xy, depth, conop, f, bg, oc, od, oa, fT, nc);
    float* gc = xm(px * Cn, 4); float* gd = xm(px, 4); float* ga = xm(px, 4);
    for (size_t k = 0; k < px * Cn; ++k) gc[k] = (2.f * urand() - 1.f) / (float)px;
    for (size_t k = 0; k < px; ++k) { gd[k] = (2.f * urand() - 1.f) / (float)px; ga[k] = (2.f * urand() - 1.f) / (float)px; }
    double* dm2 = xm((size_t)2 * P, 8); double* dcon = xm((size_t)3 * P, 8); double* dop = xm(P, 8);
    double* dcol = xm((size_t)P * Cn, 8); double* ddep = xm(P, 8);
    orc_composite_bwd(&st, ranges, vals, xy, depth, conop, f, bg, gc, gd, band ? NULL : ga, dm2, dcon, dop, dcol, ddep);
    float* dm3 = xm((size_t)3 * P, 4); float* dm2o = xm((size_t)3 * P, 4); float* dsc = xm((size_t)3 * P, 4);
    float* drot = xm((size_t)4 * P, 4); float* dcov = xm((size_t)6 * P, 4); float* dsh = xm((size_t)P * 48, 4);
    double dV[16] = {0}, dPM[16] = {0}, dcam[3] = {0};
    orc_preprocess_bwd(&st, P, means, use_sh ? shs : NULL, use_cov ? NULL : scales, use_cov ? NULL : rots,
                       use_cov ? cov : NULL, V, PM, campos, radii, cov3, cl, dm2, dcon, ddep, use_sh ? dcol : NULL, dm3, dm2o,
                       use_cov ? NULL : dsc, use_cov ? NULL : drot, use_cov ? dcov : NULL, use_sh ? dsh : NULL, dV, dPM, dcam);
    double sum = (double)R;
    for (size_t k = 0; k < px * Cn; ++k) sum += oc[k];
    for (int k = 0; k < 3 * P; ++k) sum += dm3[k];
    for (int k = 0; k < 16; ++k) sum += dV[k] + dPM[k];
    void* all[] = {means, opac, scales, rots, cov, feat, shs, radii, xy, depth, cov3, conop, tt, rgb, cl, keys, vals, ranges,
                   oc, od, oa, fT, nc, gc, gd, ga, dm2, dcon, dop, dcol, ddep, dm3, dm2o, dsc, drot, dcov, dsh};
    for (size_t k = 0; k < sizeof(all) / sizeof(all[0]); ++k) free(all[k]);
    orc_set_alpha_mode(0);
    orc_set_row_band(0, 0x7fffffff);
    return sum;
}

int main(int argc, char** argv)
{
    const int P = argc > 1 ? atoi(argv[1]) : 10000;
    const int W = argc > 2 ? atoi(argv[2]) : 640, H = argc > 3 ? atoi(argv[3]) : 480;
    double s = 0.0;
    s += run_case(P, W, H, 4, 3, 0, 0, 0, 0);          /* reference layout: [rgb | kp], 3-entry background */
    s += run_case(P, W, H, 3, 3, 0, 0, 1, 0);          /* S0: C = 3, lineage-literal alpha form */
    s += run_case(P / 4, W / 2 + 3, H / 2 + 5, 3, 3, 1, 1, 0, 0);   /* SH degree 3 + precomputed covariance, ragged size */
    s += run_case(P / 4, W, H, 35, 3, 0, 0, 0, 1);     /* wide rows, row band, no alpha gradient */
    s += run_case(0, 64, 48, 3, 3, 0, 0, 0, 0);        /* empty scene */
    s += run_case(1, 17, 9, 1, 0, 0, 0, 0, 0);         /* one Gaussian, image smaller than a tile, no background */
    /* mark_visible, dist2 (incl. N < 4), exp2 */
    for (int N = 0; N <= 5000; N = N < 5 ? N + 1 : N * 10) {
        float* pts = xm((size_t)3 * N, 4); float* out = xm(N, 4); uint8_t* vis = xm(N, 1);
        float V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        for (int k = 0; k < 3 * N; ++k) pts[k] = nrand();
        if (N > 5) { pts[15] = pts[12]; pts[16] = pts[13]; pts[17] = pts[14]; }
        orc_dist2(N, pts, out);
        orc_mark_visible(N, pts, V, vis);
        for (int k = 0; k < N; ++k) s += out[k] + vis[k];
        free(pts); free(out); free(vis);
    }
    {
        const int n = 100000;
        float* x = xm(n, 4); float* y = xm(n, 4);
        for (int k = 0; k < n; ++k) x[k] = -300.0f + 301.0f * urand();
        x[0] = -INFINITY; x[1] = NAN; x[2] = 1e30f; x[3] = -1e30f;
        orc_exp2_array(n, x, y);
        for (int k = 4; k < n; ++k) s += y[k];
        if (y[0] != 0.0f || y[1] != 0.0f || y[3] != 0.0f) { fprintf(stderr, "exp2 specials\n"); return 4; }
        free(x); free(y);
    }
    printf("orc_asan ok checksum %.9e\n", s);
    return 0;
}
