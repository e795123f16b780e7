/*
 * TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.  See ptr_oracle.h for scope and pinning.
 *
 * Every arithmetic expression below keeps the reference's evaluation order (left to right,
 * one IEEE-754 double rounding per operation).  Build with -ffp-contract=off: a fused
 * multiply-add would change low bits and break bit parity with the JavaScript reference.
 */
#include "ptr_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORC_INFINITY 1e8      /* RayTracer.js:6 — a real far clip, not IEEE infinity */
#define ORC_BIAS     1e-4     /* RayTracer.js:52 */

/* plane accessors for the 12-plane SoA block */
#define PL(k) (soa + (size_t)(k) * (size_t)n)

/* ---- JS Math helpers -------------------------------------------------------------------- */

/* Math.min(1, c): NaN if c is NaN (RayTracer.js:179-181) */
static double js_min1(double c) { return (c != c) ? c : (c < 1.0 ? c : 1.0); }

/* Math.max(0, x): NaN if x is NaN (RayTracer.js:87) */
static double js_max0(double x) { return (x != x) ? x : (x > 0.0 ? x : 0.0); }

/* Math.round: nearest integer, ties towards +infinity (RayTracer.js:184-186) */
static double js_round(double x)
{
    if (x != x || isinf(x)) return x;
    double f = floor(x);
    return (x - f >= 0.5) ? f + 1.0 : f;
}

/* ECMAScript ToInt32 (operands of << and |, RayTracer.js:188-192) */
static int32_t js_to_int32(double x)
{
    if (x != x || isinf(x)) return 0;
    double t = trunc(x);
    double m = fmod(t, 4294967296.0);          /* exact */
    if (m < 0) m += 4294967296.0;
    return (int32_t)(uint32_t)m;
}

uint32_t orc_pack(const double rgb[3])
{
    double r = js_round(js_min1(rgb[0]) * 255.0);
    double g = js_round(js_min1(rgb[1]) * 255.0);
    double b = js_round(js_min1(rgb[2]) * 255.0);
    uint32_t v = (uint32_t)(255u << 24)
               | ((uint32_t)js_to_int32(b) << 16)
               | ((uint32_t)js_to_int32(g) << 8)
               | (uint32_t)js_to_int32(r);
    return v;                                   /* stored through a Uint32Array view */
}

double orc_reference_angle(void)
{
    /* RayTracer.js:152-154: Math.tan(Math.PI * 0.5 * fov / 180), fov = 30 */
    double fov = 30.0;
    return tan(3.141592653589793 * 0.5 * fov / 180.0);
}

/* ---- Vector3.normalize (Vector3.js:23-33): invLen = 1/sqrt(len2), three multiplies ------- */
static void v_normalize(double v[3])
{
    double len2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (len2 > 0) {
        double inv = 1.0 / sqrt(len2);
        v[0] *= inv; v[1] *= inv; v[2] *= inv;
    }
}

/* ---- Sphere.intersect (Sphere.js:15-33) -------------------------------------------------- */
static inline int intersect_r2(const double o[3], const double d[3],
                               double cx, double cy, double cz, double radius2,
                               double *t0, double *t1, orc_counters *cnt)
{
    double lx = cx - o[0], ly = cy - o[1], lz = cz - o[2];
    double tca = lx * d[0] + ly * d[1] + lz * d[2];
    if (tca < 0) { if (cnt) cnt->exit_tca++; return 0; }
    double d2 = (lx * lx + ly * ly + lz * lz) - tca * tca;
    if (d2 > radius2) { if (cnt) cnt->exit_d2++; return 0; }
    double thc = sqrt(radius2 - d2);
    *t0 = tca - thc;
    *t1 = tca + thc;
    if (cnt) cnt->exit_hit++;
    return 1;
}

int orc_intersect(const double o[3], const double d[3], const double c[3], double radius,
                  double *t0, double *t1)
{
    return intersect_r2(o, d, c[0], c[1], c[2], radius * radius /* Sphere.js:11 */, t0, t1, 0);
}

/* ---- RayTracer.trace (RayTracer.js:18-95) ------------------------------------------------ */
void orc_trace(const double *soa, int n, const double bg[3],
               const double o[3], const double d[3], double rgb[3],
               orc_hit *hit, orc_counters *cnt)
{
    const double *cx = PL(0), *cy = PL(1), *cz = PL(2), *rad = PL(3);
    const double *sr = PL(4), *sg = PL(5), *sb = PL(6);
    const double *er = PL(9), *eg = PL(10), *eb = PL(11);

    if (cnt) cnt->primary++;

    /* nearest hit, scene order, strict '<' so the first element wins ties (:26-41) */
    double tnear = ORC_INFINITY;
    int element = -1;
    for (int i = 0; i < n; i++) {
        double t0 = ORC_INFINITY, t1 = ORC_INFINITY;
        if (intersect_r2(o, d, cx[i], cy[i], cz[i], rad[i] * rad[i], &t0, &t1, cnt)) {
            if (t0 < 0) t0 = t1;
            if (t0 < tnear) { tnear = t0; element = i; }
        }
    }
    if (hit) { hit->element = element; hit->tnear = tnear; }
    if (element < 0) {                       /* :43-46 */
        rgb[0] = bg[0]; rgb[1] = bg[1]; rgb[2] = bg[2];
        if (hit) memset(hit->point, 0, sizeof hit->point), memset(hit->normal, 0, sizeof hit->normal);
        return;
    }
    if (cnt) cnt->hits++;

    /* hit frame (:48-58) */
    double P[3] = { o[0] + d[0] * tnear, o[1] + d[1] * tnear, o[2] + d[2] * tnear };
    double N[3] = { P[0] - cx[element], P[1] - cy[element], P[2] - cz[element] };   /* Sphere.js:47-51 */
    v_normalize(N);
    if (d[0] * N[0] + d[1] * N[1] + d[2] * N[2] > 0) { N[0] = -N[0]; N[1] = -N[1]; N[2] = -N[2]; }
    if (hit) { memcpy(hit->point, P, sizeof P); memcpy(hit->normal, N, sizeof N); }

    /* direct lighting from every emissive element, binary shadow term (:60-91) */
    double acc[3] = { 0, 0, 0 };
    for (int i = 0; i < n; i++) {
        if (er[i] > 0 || eg[i] > 0 || eb[i] > 0) {
            if (cnt) { cnt->shadow_rays++; cnt->light_terms++; }
            double tr = 1.0;
            double L[3] = { cx[i] - P[0], cy[i] - P[1], cz[i] - P[2] };
            v_normalize(L);
            for (int j = 0; j < n; j++) {
                if (i != j) {
                    /* origin recomputed per j in the reference; same value every time (:77) */
                    double so[3] = { P[0] + N[0] * ORC_BIAS, P[1] + N[1] * ORC_BIAS, P[2] + N[2] * ORC_BIAS };
                    double t0, t1;
                    if (intersect_r2(so, L, cx[j], cy[j], cz[j], rad[j] * rad[j], &t0, &t1, cnt)) {
                        tr = 0.0;             /* any hit blocks, distance ignored (F7) */
                        break;
                    }
                }
            }
            double ratio = js_max0(N[0] * L[0] + N[1] * L[1] + N[2] * L[2]);
            /* surfaceColor.clone().product(transmission).product(emission.clone().multiply(ratio)) (:89) */
            acc[0] += (sr[element] * tr) * (er[i] * ratio);
            acc[1] += (sg[element] * tr) * (eg[i] * ratio);
            acc[2] += (sb[element] * tr) * (eb[i] * ratio);
        }
    }
    acc[0] += er[element]; acc[1] += eg[element]; acc[2] += eb[element];     /* :93 */
    rgb[0] = acc[0]; rgb[1] = acc[1]; rgb[2] = acc[2];
}

/* ---- rendercache-js: Sample.doRaytracing -> Raytracer.trace (RC/src/Sample.js:49-75, RC/src/Raytracer.js:24-123) --
 * Same colour arithmetic as RayTracer.trace; the render cache also keeps sample.hit and sample.normalDir:
 * miss: hit = o + d * 1e5, normalDir = -hit (:62-68); hit: the hit point and the element's outward normal
 * normalized a second time, before it is turned towards the ray (:71-76). */
void orc_trace_sample(const double *soa, int n, const double bg[3], const double o[3], const double d[3],
                      double rgb[3], double rgb8[3], double hit_xyz[3], double normal_xyz[3])
{
    orc_hit h;
    orc_trace(soa, n, bg, o, d, rgb, &h, 0);
    if (h.element < 0) {
        for (int a = 0; a < 3; a++) { hit_xyz[a] = o[a] + d[a] * 100000.0; normal_xyz[a] = -hit_xyz[a]; }
    } else {
        const double *cx = PL(0), *cy = PL(1), *cz = PL(2);
        double N[3] = { h.point[0] - cx[h.element], h.point[1] - cy[h.element], h.point[2] - cz[h.element] };
        v_normalize(N);
        v_normalize(N);
        memcpy(hit_xyz, h.point, sizeof h.point);
        memcpy(normal_xyz, N, sizeof N);
    }
    for (int a = 0; a < 3; a++) rgb8[a] = js_round(js_min1(rgb[a]) * 255.0);
}

/* ---- RayTracer.requestPixel (RayTracer.js:98-115) ---------------------------------------- */
void orc_request_pixel(const double *soa, int n, const double bg[3], const double d[3], double rgb[3])
{
    const double o[3] = { 0, 0, 0 };
    double c[3];
    orc_trace(soa, n, bg, o, d, c, 0, 0);
    rgb[0] = js_round(js_min1(c[0]) * 255.0);
    rgb[1] = js_round(js_min1(c[1]) * 255.0);
    rgb[2] = js_round(js_min1(c[2]) * 255.0);
}

/* ---- RayTracer.render (RayTracer.js:132-197) --------------------------------------------- */
static void primary_dir(int x, int y, double invW, double invH, double angle, double aspect, double dir[3])
{
    /* :171-174 */
    double xx = (2 * ((x + 0.5) * invW) - 1) * angle * aspect;
    double yy = (1 - 2 * ((y + 0.5) * invH)) * angle;
    dir[0] = xx; dir[1] = yy; dir[2] = -1;
    v_normalize(dir);
}

static void render_rows(const double *soa, int n, const double bg[3],
                        int width, int height, int start_y, int scan_height,
                        uint32_t flags, uint32_t *out, orc_counters *cnt,
                        int row_first, int row_step)
{
    const double invW = 1.0 / width, invH = 1.0 / height;
    const double aspect = (double)width / (double)height;
    const double angle = orc_reference_angle();
    const double origin[3] = { 0, 0, 0 };
    for (int y = start_y + row_first; y < start_y + scan_height; y += row_step) {
        for (int x = 0; x < width; x++) {
            long long idx;
            if (flags & ORC_QUIRK_ROW_SHIFT)
                idx = (long long)width * (y - start_y) - (width - x);   /* :169 */
            else
                idx = (long long)width * (y - start_y) + x;
            double dir[3], c[3];
            primary_dir(x, y, invW, invH, angle, aspect, dir);
            orc_trace(soa, n, bg, origin, dir, c, 0, cnt);
            if (idx >= 0)                      /* negative typed-array index: silently dropped */
                out[idx] = orc_pack(c);
        }
    }
}

void orc_render(const double *soa, int n, const double bg[3],
                int width, int height, int start_y, int scan_height,
                uint32_t flags, uint32_t *out, orc_counters *cnt)
{
    memset(out, 0, (size_t)width * (size_t)scan_height * 4);   /* fresh ArrayBuffer (:148) */
    render_rows(soa, n, bg, width, height, start_y, scan_height, flags, out, cnt, 0, 1);
}

typedef struct {
    const double *soa; int n; const double *bg;
    int width, height, start_y, scan_height; uint32_t flags; uint32_t *out;
    int first, step;
} mt_job;

static void *mt_main(void *p)
{
    mt_job *j = (mt_job *)p;
    render_rows(j->soa, j->n, j->bg, j->width, j->height, j->start_y, j->scan_height,
                j->flags, j->out, 0, j->first, j->step);
    return 0;
}

void orc_render_mt(const double *soa, int n, const double bg[3],
                   int width, int height, int start_y, int scan_height,
                   uint32_t flags, uint32_t *out, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    memset(out, 0, (size_t)width * (size_t)scan_height * 4);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    mt_job *jobs = (mt_job *)malloc(sizeof(mt_job) * (size_t)nthreads);
    for (int t = 0; t < nthreads; t++) {
        mt_job j = { soa, n, bg, width, height, start_y, scan_height, flags, out, t, nthreads };
        jobs[t] = j;
        pthread_create(&th[t], 0, mt_main, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], 0);
    free(th); free(jobs);
}

void orc_render_rgb64(const double *soa, int n, const double bg[3],
                      int width, int height, int start_y, int scan_height, double *out_rgb)
{
    const double invW = 1.0 / width, invH = 1.0 / height;
    const double aspect = (double)width / (double)height;
    const double angle = orc_reference_angle();
    const double origin[3] = { 0, 0, 0 };
    for (int y = start_y; y < start_y + scan_height; y++)
        for (int x = 0; x < width; x++) {
            double dir[3];
            primary_dir(x, y, invW, invH, angle, aspect, dir);
            orc_trace(soa, n, bg, origin, dir,
                      out_rgb + 3 * ((size_t)width * (size_t)(y - start_y) + (size_t)x), 0, 0);
        }
}


/* ======================= EXTENSION modes (spec; see ptr_oracle.h) ============================ */

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    /* Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11) */
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#define ORC_MAX_DEPTH 16

typedef struct { double o[3], d[3], w[3]; int depth; } ext_ray;

/* planes and Phong terms (include/ptr.h, "planes and Phong"): plane k's values are XP(row)[k], rows
 * px py pz nx ny nz sr sg sb refl transp er eg eb; phong = ks[n + n_planes] then exponent[n + n_planes] */
#define XP(k) (x->planes + (size_t)(k) * (size_t)x->n_planes)

/* Plane.intersect of the specification: denom = d.n; 0 -> miss; t = ((p - o).n) / denom; t < 0 -> miss */
static int plane_intersect(const orc_ext_extra *x, int k, const double o[3], const double d[3], double *t0, orc_counters *cnt)
{
    const double nx = XP(3)[k], ny = XP(4)[k], nz = XP(5)[k];
    const double denom = d[0] * nx + d[1] * ny + d[2] * nz;
    if (denom == 0) { if (cnt) cnt->exit_tca++; return 0; }
    const double t = ((XP(0)[k] - o[0]) * nx + (XP(1)[k] - o[1]) * ny + (XP(2)[k] - o[2]) * nz) / denom;
    if (t < 0) { if (cnt) cnt->exit_tca++; return 0; }
    if (cnt) cnt->exit_hit++;
    *t0 = t;
    return 1;
}

/* nearest hit + hit frame shared by every node of the expansion (same arithmetic as orc_trace); spheres, then planes */
static int ext_nearest(const double *soa, int n, const orc_ext_extra *x, const double o[3], const double d[3], double *tnear_out,
                       orc_counters *cnt)
{
    const double *cx = PL(0), *cy = PL(1), *cz = PL(2), *rad = PL(3);
    double tnear = ORC_INFINITY;
    int element = -1;
    for (int i = 0; i < n; i++) {
        double t0 = ORC_INFINITY, t1 = ORC_INFINITY;
        if (intersect_r2(o, d, cx[i], cy[i], cz[i], rad[i] * rad[i], &t0, &t1, cnt)) {
            if (t0 < 0) t0 = t1;
            if (t0 < tnear) { tnear = t0; element = i; }
        }
    }
    for (int k = 0; x && k < x->n_planes; k++) {
        double t0 = ORC_INFINITY;
        if (plane_intersect(x, k, o, d, &t0, cnt) && t0 < tnear) { tnear = t0; element = n + k; }
    }
    *tnear_out = tnear;
    return element;
}

static void trace_ext_counted(const double *soa, int n, const orc_ext_extra *x, const double bg[3],
                              const double o[3], const double d[3], int max_depth, double rgb[3],
                              orc_counters *cnt, uint64_t *secondary);

void orc_trace_ext(const double *soa, int n, const double bg[3],
                   const double o[3], const double d[3], int max_depth, double rgb[3])
{
    trace_ext_counted(soa, n, 0, bg, o, d, max_depth, rgb, 0, 0);
}

void orc_trace_ext2(const double *soa, int n, const orc_ext_extra *x, const double bg[3],
                    const double o[3], const double d[3], int max_depth, double rgb[3])
{
    trace_ext_counted(soa, n, x, bg, o, d, max_depth, rgb, 0, 0);
}

/* the material / position values of element e (a sphere, or plane e - n) */
typedef struct { double c[3], sc[3], em[3], refl, transp; } ext_elem;
static ext_elem ext_element(const double *soa, int n, const orc_ext_extra *x, int e)
{
    ext_elem m;
    if (e < n) {
        m.c[0] = PL(0)[e]; m.c[1] = PL(1)[e]; m.c[2] = PL(2)[e];
        m.sc[0] = PL(4)[e]; m.sc[1] = PL(5)[e]; m.sc[2] = PL(6)[e];
        m.refl = PL(7)[e]; m.transp = PL(8)[e];
        m.em[0] = PL(9)[e]; m.em[1] = PL(10)[e]; m.em[2] = PL(11)[e];
    } else {
        const int k = e - n;
        m.c[0] = XP(0)[k]; m.c[1] = XP(1)[k]; m.c[2] = XP(2)[k];
        m.sc[0] = XP(6)[k]; m.sc[1] = XP(7)[k]; m.sc[2] = XP(8)[k];
        m.refl = XP(9)[k]; m.transp = XP(10)[k];
        m.em[0] = XP(11)[k]; m.em[1] = XP(12)[k]; m.em[2] = XP(13)[k];
    }
    return m;
}

static void trace_ext_counted(const double *soa, int n, const orc_ext_extra *x, const double bg[3],
                              const double o[3], const double d[3], int max_depth, double rgb[3],
                              orc_counters *cnt, uint64_t *secondary)
{
    const double *cx = PL(0), *cy = PL(1), *cz = PL(2), *rad = PL(3);
    const int np = x ? x->n_planes : 0, ne = n + np;
    if (max_depth > ORC_MAX_DEPTH) max_depth = ORC_MAX_DEPTH;
    ext_ray stack[ORC_MAX_DEPTH + 2];
    int top = 0;
    double acc[3] = { 0, 0, 0 };
    memcpy(stack[0].o, o, sizeof stack[0].o); memcpy(stack[0].d, d, sizeof stack[0].d);
    stack[0].w[0] = stack[0].w[1] = stack[0].w[2] = 1.0; stack[0].depth = 0;
    top = 1;
    while (top > 0) {
        const ext_ray r = stack[--top];
        double tnear;
        const int e = ext_nearest(soa, n, x, r.o, r.d, &tnear, cnt);
        if (e < 0) {
            acc[0] += r.w[0] * bg[0]; acc[1] += r.w[1] * bg[1]; acc[2] += r.w[2] * bg[2];
            continue;
        }
        if (cnt) cnt->hits++;
        const ext_elem me = ext_element(soa, n, x, e);
        const double *er = 0, *eg = 0, *eb = 0; (void)er; (void)eg; (void)eb;
        const double sr_e = me.sc[0], sg_e = me.sc[1], sb_e = me.sc[2], refl_e = me.refl, transp_e = me.transp;
        acc[0] += r.w[0] * me.em[0]; acc[1] += r.w[1] * me.em[1]; acc[2] += r.w[2] * me.em[2];
        double P[3] = { r.o[0] + r.d[0] * tnear, r.o[1] + r.d[1] * tnear, r.o[2] + r.d[2] * tnear };
        double N[3];
        if (e < n) {
            N[0] = P[0] - cx[e]; N[1] = P[1] - cy[e]; N[2] = P[2] - cz[e];
            v_normalize(N);
        } else {                       /* a plane's normal is the unit vector it was given */
            N[0] = XP(3)[e - n]; N[1] = XP(4)[e - n]; N[2] = XP(5)[e - n];
        }
        int inside = 0;
        double dn = r.d[0] * N[0] + r.d[1] * N[1] + r.d[2] * N[2];
        if (dn > 0) { N[0] = -N[0]; N[1] = -N[1]; N[2] = -N[2]; inside = 1; dn = -dn; }
        if ((transp_e > 0 || refl_e > 0) && r.depth < max_depth) {
            const double facing = -dn;
            const double f = 1 - facing;
            const double fresnel = 1 * 0.1 + (f * f * f) * (1 - 0.1);
            const double wsc[3] = { r.w[0] * sr_e, r.w[1] * sg_e, r.w[2] * sb_e };
            if (transp_e > 0) {
                const double eta = inside ? 1.1 : 1 / 1.1;
                const double cosi = -dn;
                const double k = 1 - eta * eta * (1 - cosi * cosi);
                if (k >= 0) {
                    ext_ray *c = &stack[top++];
                    if (secondary) (*secondary)++;
                    const double g = eta * cosi - sqrt(k);
                    const double wt = (1 - fresnel) * transp_e;
                    for (int a = 0; a < 3; a++) {
                        c->o[a] = P[a] - N[a] * ORC_BIAS;
                        c->d[a] = r.d[a] * eta + N[a] * g;
                        c->w[a] = wsc[a] * wt;
                    }
                    v_normalize(c->d);
                    c->depth = r.depth + 1;
                }
            }
            ext_ray *c = &stack[top++];
            if (secondary) (*secondary)++;
            const double two = 2 * dn;
            for (int a = 0; a < 3; a++) {
                c->o[a] = P[a] + N[a] * ORC_BIAS;
                c->d[a] = r.d[a] - N[a] * two;
                c->w[a] = wsc[a] * fresnel;
            }
            v_normalize(c->d);
            c->depth = r.depth + 1;
        } else {
            /* the reference's direct lighting (RayTracer.js:60-91), over spheres and planes, with the Phong term */
            double dif[3] = { 0, 0, 0 };
            const double ks = (x && x->phong) ? x->phong[e] : 0.0;
            const double two_dn = 2 * dn;
            const double Rv[3] = { r.d[0] - N[0] * two_dn, r.d[1] - N[1] * two_dn, r.d[2] - N[2] * two_dn };
            for (int i = 0; i < ne; i++) {
                const ext_elem li = ext_element(soa, n, x, i);
                if (li.em[0] > 0 || li.em[1] > 0 || li.em[2] > 0) {
                    double tr = 1.0;
                    double L[3] = { li.c[0] - P[0], li.c[1] - P[1], li.c[2] - P[2] };
                    const double len2 = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];     /* a plane only blocks in front of the light */
                    v_normalize(L);
                    if (cnt) cnt->shadow_rays++;
                    for (int j = 0; j < ne; j++) {
                        if (i != j) {
                            double so[3] = { P[0] + N[0] * ORC_BIAS, P[1] + N[1] * ORC_BIAS, P[2] + N[2] * ORC_BIAS };
                            double t0, t1;
                            if (j < n ? intersect_r2(so, L, cx[j], cy[j], cz[j], rad[j] * rad[j], &t0, &t1, cnt)
                                      : (plane_intersect(x, j - n, so, L, &t0, cnt) && t0 * t0 < len2)) { tr = 0.0; break; }
                        }
                    }
                    double ratio = js_max0(N[0] * L[0] + N[1] * L[1] + N[2] * L[2]);
                    dif[0] += (sr_e * tr) * (li.em[0] * ratio);
                    dif[1] += (sg_e * tr) * (li.em[1] * ratio);
                    dif[2] += (sb_e * tr) * (li.em[2] * ratio);
                    if (ks > 0) {
                        const double sp = js_max0(Rv[0] * L[0] + Rv[1] * L[1] + Rv[2] * L[2]);
                        double ph = 1.0, b = sp;
                        for (unsigned q = (unsigned)x->phong[ne + e]; q > 0; q >>= 1) { if (q & 1u) ph = ph * b; b = b * b; }
                        const double kp = ks * ph;
                        dif[0] += tr * (li.em[0] * kp);
                        dif[1] += tr * (li.em[1] * kp);
                        dif[2] += tr * (li.em[2] * kp);
                    }
                }
            }
            acc[0] += r.w[0] * dif[0]; acc[1] += r.w[1] * dif[1]; acc[2] += r.w[2] * dif[2];
        }
    }
    rgb[0] = acc[0]; rgb[1] = acc[1]; rgb[2] = acc[2];
}

typedef struct {
    const double *soa; int n; const double *bg; int width, height, start_y, scan_height;
    int max_depth, spp; uint32_t seed; uint32_t *out8; double *out64; int first, step;
    orc_counters *cnt; uint64_t *secondary;          /* single-threaded callers only */
    const orc_ext_extra *x;
} ext_job;

static void *ext_main(void *p)
{
    const ext_job *j = (const ext_job *)p;
    const double invW = 1.0 / j->width, invH = 1.0 / j->height;
    const double aspect = (double)j->width / (double)j->height, angle = orc_reference_angle();
    const double origin[3] = { 0, 0, 0 };
    for (int y = j->start_y + j->first; y < j->start_y + j->scan_height; y += j->step)
        for (int x = 0; x < j->width; x++) {
            double sum[3] = { 0, 0, 0 };
            for (int s = 0; s < j->spp; s++) {
                double px = x + 0.5, py = y + 0.5;
                if (j->spp > 1) {
                    const uint32_t ctr[4] = { (uint32_t)x, (uint32_t)y, (uint32_t)s, 0u };
                    const uint32_t key[2] = { j->seed, 0x50545221u };
                    uint32_t rnd[4];
                    orc_philox4x32_10(ctr, key, rnd);
                    px = x + (rnd[0] + 0.5) / 4294967296.0;
                    py = y + (rnd[1] + 0.5) / 4294967296.0;
                }
                double dir[3] = { (2 * (px * invW) - 1) * angle * aspect, (1 - 2 * (py * invH)) * angle, -1 };
                v_normalize(dir);
                double c[3];
                if (j->cnt) j->cnt->primary++;
                trace_ext_counted(j->soa, j->n, j->x, j->bg, origin, dir, j->max_depth, c, j->cnt, j->secondary);
                sum[0] += c[0]; sum[1] += c[1]; sum[2] += c[2];
            }
            if (j->spp > 1) { sum[0] /= j->spp; sum[1] /= j->spp; sum[2] /= j->spp; }
            const size_t idx = (size_t)j->width * (size_t)(y - j->start_y) + (size_t)x;
            if (j->out8) j->out8[idx] = orc_pack(sum);
            if (j->out64) { j->out64[3 * idx] = sum[0]; j->out64[3 * idx + 1] = sum[1]; j->out64[3 * idx + 2] = sum[2]; }
        }
    return 0;
}

void orc_render_ext(const double *soa, int n, const double bg[3],
                    int width, int height, int start_y, int scan_height,
                    int max_depth, int spp, uint32_t seed,
                    uint32_t *out_rgba8, double *out_rgb64, int nthreads)
{
    orc_render_ext2(soa, n, 0, bg, width, height, start_y, scan_height, max_depth, spp, seed, out_rgba8, out_rgb64, nthreads);
}

void orc_render_ext2(const double *soa, int n, const orc_ext_extra *x, const double bg[3],
                     int width, int height, int start_y, int scan_height,
                     int max_depth, int spp, uint32_t seed,
                     uint32_t *out_rgba8, double *out_rgb64, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (spp < 1) spp = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    ext_job *jobs = (ext_job *)malloc(sizeof(ext_job) * (size_t)nthreads);
    for (int t = 0; t < nthreads; t++) {
        ext_job j = { soa, n, bg, width, height, start_y, scan_height, max_depth, spp, seed, out_rgba8, out_rgb64, t, nthreads, 0, 0, x };
        jobs[t] = j;
        pthread_create(&th[t], 0, ext_main, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], 0);
    free(th); free(jobs);
}

void orc_count_ext(const double *soa, int n, const double bg[3],
                   int width, int height, int start_y, int scan_height,
                   int max_depth, int spp, uint32_t seed, orc_counters *cnt, uint64_t *secondary)
{
    orc_count_ext2(soa, n, 0, bg, width, height, start_y, scan_height, max_depth, spp, seed, cnt, secondary);
}

void orc_count_ext2(const double *soa, int n, const orc_ext_extra *x, const double bg[3],
                    int width, int height, int start_y, int scan_height,
                    int max_depth, int spp, uint32_t seed, orc_counters *cnt, uint64_t *secondary)
{
    memset(cnt, 0, sizeof *cnt);
    *secondary = 0;
    ext_job j = { soa, n, bg, width, height, start_y, scan_height, max_depth, spp < 1 ? 1 : spp, seed, 0, 0, 0, 1, cnt, secondary, x };
    ext_main(&j);
}


/* ======================= rendercache-js: reprojection pass ===========================
 * Driver.reprojectFrame (RC/src/Driver.js:417-486) with Camera.reprojectPixel (RC/src/Camera.js:52-74) and the
 * three.js r139 arithmetic it calls (Vector3.project = applyMatrix4(matrixWorldInverse).applyMatrix4(projectionMatrix),
 * each with its own perspective divide; RC/js/three.js Vector3.applyMatrix4).  Matrices are column-major as in
 * THREE.Matrix4.elements.  Sequential, item order = cache order, exactly as the reference. */
static void three_apply_matrix4(double v[3], const double e[16])
{
    const double x = v[0], y = v[1], z = v[2];
    const double w = 1 / (e[3] * x + e[7] * y + e[11] * z + e[15]);
    v[0] = (e[0] * x + e[4] * y + e[8] * z + e[12]) * w;
    v[1] = (e[1] * x + e[5] * y + e[9] * z + e[13]) * w;
    v[2] = (e[2] * x + e[6] * y + e[10] * z + e[14]) * w;
}

/* Camera.reprojectPixel: returns 0 when depth === 0; x, y are the Math.round()ed numbers (may be NaN/Inf) */
int orc_rc_reproject_pixel(const double from[3], const double view[16], const double proj[16], int width, int height,
                           const double hit[3], double *x, double *y, double *depth)
{
    const double dx = hit[0] - from[0], dy = hit[1] - from[1], dz = hit[2] - from[2];
    *depth = sqrt(dx * dx + dy * dy + dz * dz);
    double v[3] = { hit[0], hit[1], hit[2] };
    three_apply_matrix4(v, view);
    three_apply_matrix4(v, proj);
    const double hx = 0.5 * width, hy = 0.5 * height;
    *x = js_round(v[0] * hx + hx);
    *y = js_round(v[1] * hy + hy);
    return *depth == 0 ? 0 : 1;
}

void orc_rc_reproject_frame(const double from[3], const double view[16], const double proj[16], int width, int height,
                            int n_items, const double *hit_xyz /* NaN x = no hit */, double *age, double age_factor,
                            double depth_threshold, int32_t *item_pixel, int32_t *pixel_element, double *pixel_depth)
{
    const int stride = width + 2;
    /* detachFrameFromCache (:405-416): interior pixels only */
    for (int y = 1; y <= height; y++)
        for (int x = 1; x <= width; x++) { pixel_element[y * stride + x] = -1; pixel_depth[y * stride + x] = depth_threshold; }
    for (int i = 0; i < n_items; i++) {
        item_pixel[i] = -1;                                         /* cacheItem.pixel = null */
        const double *h = hit_xyz + 3 * (size_t)i;
        if (h[0] != h[0]) continue;                                 /* cacheItem.hit === null */
        double x, y, depth;
        if (orc_rc_reproject_pixel(from, view, proj, width, height, h, &x, &y, &depth)) {
            if (x >= 1 && x <= width && y >= 1 && y <= height) {
                const int p = (int)y * stride + (int)x;
                if (depth < pixel_depth[p]) {
                    if (pixel_element[p] >= 0) age[pixel_element[p]] += age_factor;
                    item_pixel[i] = p;                              /* attach: item.pixel = pixel, pixel.element = item */
                    pixel_element[p] = i;
                    pixel_depth[p] = depth;
                } else {
                    age[i] += age_factor;
                }
            } else {
                age[i] += age_factor * 2;
            }
        } else {
            age[i] += age_factor * 4;
        }
    }
}


/* ======================= rendercache-js: depth culling, gap filling, frames ===========
 * Driver.depthCulling (RC/src/Driver.js:488-554), Driver.fillGaps (:556-679), getColorFrame / getReprojectionFrame /
 * getPriorityFrame (:960-1018), on the state reprojectFrame leaves.  pixel.weight is 0 for every pixel in the
 * reference (Driver.allocBuffers sets this.weight on the Driver, :236), so the neighbourhood weight is 0. */
void orc_rc_shade_frame(int width, int height, int n_items, const int32_t *pixel_element, const double *pixel_depth,
                        double *item_age, const double *item_rgb, double age_factor,
                        double interpolation_zero, double interpolation_random, double priority_max,
                        double *pixel_rgb, double *pixel_priority, double *hist /* 3 x 256 */, double stats[3],
                        uint32_t *color_frame, uint32_t *reprojection_frame, uint32_t *priority_frame)
{
    const int stride = width + 2;
    (void)n_items;
    /* depthCulling */
    for (int y = 1; y <= height; y++)
        for (int x = 1; x <= width; x++) {
            const int p = y * stride + x, e = pixel_element[p];
            if (e >= 0) {
                double items = 0, sum = 0;
                for (int i = -1; i <= 1; i++)
                    for (int j = -1; j <= 1; j++)
                        if (i != 0 && j != 0) {
                            const int q = (y + j) * stride + (x + i);
                            /* (border cells count too when they carry an element: the reference never detaches them) */
                            if (pixel_element[q] >= 0) { sum += pixel_depth[q]; items++; }
                        }
                if (items != 0) {
                    const double ratio = sum / (items * pixel_depth[p]);
                    if (ratio < 0.9 || ratio > 1.1) item_age[e] += (age_factor * 10);
                }
                pixel_rgb[3 * p] = item_rgb[3 * e]; pixel_rgb[3 * p + 1] = item_rgb[3 * e + 1]; pixel_rgb[3 * p + 2] = item_rgb[3 * e + 2];
            } else {
                pixel_rgb[3 * p] = pixel_rgb[3 * p + 1] = pixel_rgb[3 * p + 2] = 0;
            }
        }
    /* fillGaps */
    for (int k = 0; k < 3 * 256; k++) hist[k] = 0;
    double total = 0, member_total = 0, completeness = 0;
    for (int y = 1; y <= height; y++)
        for (int x = 1; x <= width; x++) {
            const int p = y * stride + x, e = pixel_element[p];
            double prio;
            if (e < 0) {
                double age = 0, weight = 0, items = 0, cw = 0, r = 0, g = 0, b = 0;
                for (int i = -1; i <= 1; i++)
                    for (int j = -1; j <= 1; j++) {
                        const int q = (y + j) * stride + (x + i);
                        const double iw = (i != 0 && j != 0) ? 1 : 2;
                        weight += 0;                                             /* pixel.weight */
                        if (pixel_element[q] >= 0) {
                            r += pixel_rgb[3 * q] * iw; g += pixel_rgb[3 * q + 1] * iw; b += pixel_rgb[3 * q + 2] * iw;
                            age += item_age[pixel_element[q]];
                            cw += iw;
                            items++;
                        }
                    }
                if (items > 0) {
                    pixel_rgb[3 * p] = floor(r / cw); pixel_rgb[3 * p + 1] = floor(g / cw); pixel_rgb[3 * p + 2] = floor(b / cw);
                    prio = age / items;
                    prio += (interpolation_zero + (weight - items) * interpolation_random);
                    prio = js_round(prio);
                    if (prio >= 0 && prio < 256 && prio == floor(prio)) hist[0 * 256 + (int)prio]++;
                    member_total += prio;
                } else {
                    pixel_rgb[3 * p] = pixel_rgb[3 * p + 1] = pixel_rgb[3 * p + 2] = 0;
                    prio = priority_max;
                    if (prio >= 0 && prio < 256 && prio == floor(prio)) hist[1 * 256 + (int)prio]++;
                    total += prio;
                }
            } else {
                prio = (double)(js_to_int32(item_age[e]) >> 1);
                if (prio > 0) total += prio;
                if (prio >= 0 && prio < 256) hist[2 * 256 + (int)prio]++;
                completeness += 1.0;
            }
            pixel_priority[p] = prio;
        }
    stats[0] = 100.0 * (completeness / ((double)width * (double)height));
    stats[1] = total;
    stats[2] = member_total;
    /* frames: rows bottom-up (:965) */
    size_t at = 0;
    for (int y = height - 1; y >= 0; y--)
        for (int x = 0; x < width; x++, at++) {
            const int p = (y + 1) * stride + (x + 1), e = pixel_element[p];
            const uint32_t alpha = 255u << 24;
            color_frame[at] = alpha | ((uint32_t)js_to_int32(pixel_rgb[3 * p + 2]) << 16) | ((uint32_t)js_to_int32(pixel_rgb[3 * p + 1]) << 8)
                              | (uint32_t)js_to_int32(pixel_rgb[3 * p]);
            reprojection_frame[at] = e >= 0 ? (alpha | ((uint32_t)js_to_int32(item_rgb[3 * e + 2]) << 16) | ((uint32_t)js_to_int32(item_rgb[3 * e + 1]) << 8)
                                               | (uint32_t)js_to_int32(item_rgb[3 * e])) : alpha;
            const uint32_t pr = (uint32_t)js_to_int32(pixel_priority[p]);
            priority_frame[at] = alpha | (pr << 16) | (pr << 8) | pr;
        }
}
