/*
 * ptr_napi.c — thin N-API addon over the C ABI of include/ptr.h (libptr_hip.so).
 *
 * This is the binding a maintainer of raytracer-js adds to replace the canvas/worker render loop
 * (RT/src/RayTracer.js:132-197 called from RT/examples/example1.js:65 and RT/src/RenderWorker.js:61).
 * It holds no rendering logic: every call converts arguments and forwards to one ptr_* function.
 * Raw node_api.h (C), N-API version <= 8, context-aware (NAPI_MODULE_INIT, no global state).
 *
 *   create(device)                                   -> external handle      ptr_create
 *   destroy(h)                                                               ptr_destroy
 *   setScene(h, Float64Array soa, n, nFrames)                                ptr_set_scene_sequence_f64
 *   setBackground(h, r, g, b)                                                ptr_set_background_f64
 *   setCamera(h, Float64Array(13) | null)                                    ptr_set_camera
 *   setExtension(h, Float64Array planes | null, nPlanes, Float64Array phong | null, nElements)   ptr_set_extension_f64
 *   render(h, w, height, startY, scanHeight, flags)  -> ArrayBuffer (RGBA8)  ptr_render
 *   renderInto(h, target, w, height, startY, scanHeight, flags) -> target  ptr_render (into the caller's own memory)
 *   renderAsync(h, w, height, startY, scanHeight, flags) -> Promise<ArrayBuffer>
 *   traceSamples(h, Float64Array origin(3), Float64Array dirs) -> {rgb, rgb8, hit, normal}   ptr_trace_samples_f64
 *   traceBatch(h, Float64Array origins | null, Float64Array dirs)
 *        -> { rgb: Float64Array, element: Int32Array, hit: Float64Array, normal: Float64Array }
 *                                                                            ptr_trace_batch_f64
 *   commInitAll([h...])                                                      ptr_comm_init_all
 *   renderMulti([h...], w, height, flags[, maxDepth, spp, seed]) -> ArrayBuffer   ptr_render_multi
 *   rcCacheCreate / rcCacheUpload / rcCacheDownload / rcNextFrame / rcGetFrame / rcCacheDestroy: the render cache's frame
 *        loop on a cache resident on the GPU                                  ptr_rc_cache_*, ptr_rc_next_frame, ptr_rc_get_frames
 *   stats(h) -> { primary, hits, shadowRays, exitTca, exitD2, exitHit, retraced, reshaded }   ptr_get_stats
 */
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../../include/ptr.h"

#define NAPI_TRY(env, call)                                                                    \
    do {                                                                                       \
        if ((call) != napi_ok) {                                                               \
            napi_throw_error((env), NULL, "N-API call failed: " #call);                        \
            return NULL;                                                                       \
        }                                                                                      \
    } while (0)

static napi_value throw_ptr(napi_env env, ptr_ctx *ctx, int rc)
{
    const char *msg = ptr_last_error(ctx);
    const char *name = rc == PTR_ERR_INVALID ? "PTR_ERR_INVALID" : rc == PTR_ERR_NO_DEVICE ? "PTR_ERR_NO_DEVICE"
                     : rc == PTR_ERR_HIP ? "PTR_ERR_HIP" : rc == PTR_ERR_NO_SCENE ? "PTR_ERR_NO_SCENE"
                     : rc == PTR_ERR_UNSUPPORTED ? "PTR_ERR_UNSUPPORTED" : "PTR_ERR";
    if (rc == PTR_ERR_INVALID) napi_throw_range_error(env, name, msg);
    else napi_throw_error(env, name, msg);
    return NULL;
}

static void finalize_ctx(napi_env env, void *data, void *hint)
{
    (void)env; (void)hint;
    ptr_destroy((ptr_ctx *)data);          /* the GC owns handles that were never destroy()ed */
}

/* Handles are napi externals whose data pointer is cleared by destroy() via a wrapper cell.  renderAsync jobs run
 * ptr_render on a libuv worker thread: while any is in flight (`jobs`) a destroy() only marks the handle dead
 * (`closing`) and the last job's completion callback destroys the context.  Both fields are touched on the JavaScript
 * thread only (destroy(), the job's start and its completion callback all run there).                              */
typedef struct { ptr_ctx *ctx; int jobs; int closing; } handle_cell;

static void finalize_cell(napi_env env, void *data, void *hint)
{
    handle_cell *cell = (handle_cell *)data;
    if (cell->ctx) finalize_ctx(env, cell->ctx, hint);
    free(cell);
}

static ptr_ctx *get_ctx(napi_env env, napi_value v)
{
    handle_cell *cell = NULL;
    if (napi_get_value_external(env, v, (void **)&cell) != napi_ok || !cell || !cell->ctx || cell->closing) {
        napi_throw_type_error(env, NULL, "not a live renderer handle");
        return NULL;
    }
    return cell->ctx;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv)
{
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        napi_throw_type_error(env, NULL, "wrong number of arguments");
        return 0;
    }
    return 1;
}

/* at least `want`, at most `max` arguments; missing optional ones are left as undefined */
static int get_args_opt(napi_env env, napi_callback_info info, size_t want, size_t max, napi_value *argv)
{
    size_t argc = max;
    napi_value undef;
    if (napi_get_undefined(env, &undef) != napi_ok) return 0;
    for (size_t k = 0; k < max; k++) argv[k] = undef;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        napi_throw_type_error(env, NULL, "wrong number of arguments");
        return 0;
    }
    return 1;
}

static int get_i32(napi_env env, napi_value v, int32_t *out)
{
    if (napi_get_value_int32(env, v, out) != napi_ok) {
        napi_throw_type_error(env, NULL, "expected an integer");
        return 0;
    }
    return 1;
}

/* Float64Array -> (pointer, element count); null/undefined -> (NULL, 0) */
static int get_f64_array(napi_env env, napi_value v, double **data, size_t *len)
{
    napi_valuetype t;
    *data = NULL; *len = 0;
    if (napi_typeof(env, v, &t) != napi_ok) return 0;
    if (t == napi_null || t == napi_undefined) return 1;
    bool is_ta = false;
    napi_typedarray_type tt;
    if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta ||
        napi_get_typedarray_info(env, v, &tt, len, (void **)data, NULL, NULL) != napi_ok || tt != napi_float64_array) {
        napi_throw_type_error(env, NULL, "expected a Float64Array");
        return 0;
    }
    return 1;
}

static napi_value js_create(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    int32_t device = 0;
    if (!get_args(env, info, 1, argv) || !get_i32(env, argv[0], &device)) return NULL;
    ptr_ctx *ctx = NULL;
    int rc = ptr_create(device, &ctx);
    if (rc != PTR_OK) return throw_ptr(env, NULL, rc);
    handle_cell *cell = (handle_cell *)malloc(sizeof *cell);
    if (!cell) { ptr_destroy(ctx); napi_throw_error(env, NULL, "out of memory"); return NULL; }
    cell->ctx = ctx; cell->jobs = 0; cell->closing = 0;
    napi_value h;
    if (napi_create_external(env, cell, finalize_cell, NULL, &h) != napi_ok) {
        ptr_destroy(ctx); free(cell);
        napi_throw_error(env, NULL, "napi_create_external failed");
        return NULL;
    }
    return h;
}

static napi_value js_destroy(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    handle_cell *cell = NULL;
    if (napi_get_value_external(env, argv[0], (void **)&cell) == napi_ok && cell && cell->ctx) {
        if (cell->jobs > 0) {
            cell->closing = 1;             /* a worker thread is inside ptr_render: render_complete destroys */
        } else {
            ptr_destroy(cell->ctx);
            cell->ctx = NULL;
        }
    }
    return NULL;
}

static napi_value js_set_scene(napi_env env, napi_callback_info info)
{
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    double *soa; size_t len;
    int32_t n, frames;
    if (!get_f64_array(env, argv[1], &soa, &len) || !get_i32(env, argv[2], &n) || !get_i32(env, argv[3], &frames))
        return NULL;
    if (n < 0 || frames < 1 || len < (size_t)12 * (size_t)n * (size_t)frames) {
        napi_throw_range_error(env, "PTR_ERR_INVALID", "soa must hold 12 * n * nFrames numbers");
        return NULL;
    }
    int rc = ptr_set_scene_sequence_f64(ctx, soa, n, frames);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    return NULL;
}

static napi_value js_set_background(napi_env env, napi_callback_info info)
{
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    double rgb[3];
    for (int k = 0; k < 3; k++)
        if (napi_get_value_double(env, argv[1 + k], &rgb[k]) != napi_ok) {
            napi_throw_type_error(env, NULL, "background channels must be numbers");
            return NULL;
        }
    int rc = ptr_set_background_f64(ctx, rgb);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    return NULL;
}

static napi_value js_set_camera(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    double *v; size_t len;
    if (!get_f64_array(env, argv[1], &v, &len)) return NULL;
    int rc;
    if (!v) rc = ptr_set_camera(ctx, NULL);
    else {
        if (len < 13) { napi_throw_range_error(env, "PTR_ERR_INVALID", "camera needs 13 numbers"); return NULL; }
        ptr_camera cam;
        memcpy(cam.origin, v, 3 * sizeof(double));
        memcpy(cam.right, v + 3, 3 * sizeof(double));
        memcpy(cam.up, v + 6, 3 * sizeof(double));
        memcpy(cam.forward, v + 9, 3 * sizeof(double));
        cam.tan_half_fov = v[12];
        rc = ptr_set_camera(ctx, &cam);
    }
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    return NULL;
}

/* setExtension(h, Float64Array planes | null, nPlanes, Float64Array phong | null, nElements): planes and Phong terms
 * (extension scene data, ptr_set_extension_f64) */
static napi_value js_set_extension(napi_env env, napi_callback_info info)
{
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    double *pl, *ph; size_t npl, nph;
    int32_t n_planes, n_elements;
    if (!get_f64_array(env, argv[1], &pl, &npl) || !get_i32(env, argv[2], &n_planes) ||
        !get_f64_array(env, argv[3], &ph, &nph) || !get_i32(env, argv[4], &n_elements)) return NULL;
    if (n_planes < 0 || n_elements < 0 || (n_planes > 0 && npl < (size_t)14 * (size_t)n_planes) || (ph && nph < (size_t)2 * (size_t)n_elements)) {
        napi_throw_range_error(env, "PTR_ERR_INVALID", "planes must hold 14 * nPlanes numbers and phong 2 * nElements");
        return NULL;
    }
    int rc = ptr_set_extension_f64(ctx, n_planes > 0 ? pl : NULL, n_planes, ph, n_elements);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    return NULL;
}

static int read_params(napi_env env, napi_value *argv, ptr_params *p)
{
    int32_t flags;
    memset(p, 0, sizeof *p);
    if (!get_i32(env, argv[0], &p->width) || !get_i32(env, argv[1], &p->height) ||
        !get_i32(env, argv[2], &p->start_y) || !get_i32(env, argv[3], &p->scan_height) ||
        !get_i32(env, argv[4], &flags))
        return 0;
    p->flags = (uint32_t)flags;
    p->spp = 1;
    /* optional: maxDepth, spp, seed (extension modes; include/ptr.h) */
    napi_valuetype t;
    if (napi_typeof(env, argv[5], &t) == napi_ok && t != napi_undefined && !get_i32(env, argv[5], &p->max_depth)) return 0;
    if (napi_typeof(env, argv[6], &t) == napi_ok && t != napi_undefined && !get_i32(env, argv[6], &p->spp)) return 0;
    if (napi_typeof(env, argv[7], &t) == napi_ok && t != napi_undefined) {
        if (napi_get_value_uint32(env, argv[7], &p->seed) != napi_ok) {
            napi_throw_type_error(env, NULL, "seed must be an unsigned 32-bit integer");
            return 0;
        }
    }
    return 1;
}

static int check_frame_size(napi_env env, const ptr_params *p)
{
    /* RT/src/RayTracer.js:148 allocates width*scanHeight*4 bytes; keep it inside an ArrayBuffer */
    if (p->width < 1 || p->height < 1 || p->scan_height < 1 ||
        (uint64_t)p->width * (uint64_t)p->scan_height * 4u > 0x7fffffffull) {
        napi_throw_range_error(env, "PTR_ERR_INVALID", "width/height/scanHeight must be >= 1 and width*scanHeight*4 < 2^31");
        return 0;
    }
    return 1;
}

static napi_value js_render(napi_env env, napi_callback_info info)
{
    napi_value argv[9];
    if (!get_args_opt(env, info, 6, 9, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    ptr_params p;
    if (!read_params(env, argv + 1, &p) || !check_frame_size(env, &p)) return NULL;
    void *data = NULL;
    napi_value buf;
    NAPI_TRY(env, napi_create_arraybuffer(env, (size_t)p.width * (size_t)p.scan_height * 4, &data, &buf));
    int rc = ptr_render(ctx, &p, (uint32_t *)data, NULL);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    return buf;            /* a fresh ArrayBuffer owned by the caller, RT/src/RayTracer.js:148,196 */
}

/* renderInto(h, target, w, height, startY, scanHeight, flags[, maxDepth, spp, seed]): the same frame into memory the caller
 * already has (an ArrayBuffer or a typed array of at least w * scanHeight * 4 bytes).  A loop that renders frame after frame
 * spends more time on a fresh ArrayBuffer per frame (allocation, first-touch page faults, garbage collection) than on the
 * frame itself; the reference's API hands out a new buffer per call, so render() keeps doing that and this is the addition. */
static napi_value js_render_into(napi_env env, napi_callback_info info)
{
    napi_value argv[10];
    if (!get_args_opt(env, info, 7, 10, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    ptr_params p;
    if (!read_params(env, argv + 2, &p) || !check_frame_size(env, &p)) return NULL;
    void *data = NULL;
    size_t bytes = 0;
    bool is_ab = false, is_ta = false;
    if (napi_is_arraybuffer(env, argv[1], &is_ab) == napi_ok && is_ab) {
        NAPI_TRY(env, napi_get_arraybuffer_info(env, argv[1], &data, &bytes));
    } else if (napi_is_typedarray(env, argv[1], &is_ta) == napi_ok && is_ta) {
        napi_typedarray_type tt;
        size_t len = 0;
        NAPI_TRY(env, napi_get_typedarray_info(env, argv[1], &tt, &len, &data, NULL, NULL));
        const size_t width = (tt == napi_int8_array || tt == napi_uint8_array || tt == napi_uint8_clamped_array) ? 1
                           : (tt == napi_int16_array || tt == napi_uint16_array) ? 2
                           : (tt == napi_int32_array || tt == napi_uint32_array || tt == napi_float32_array) ? 4 : 8;
        bytes = len * width;
    } else {
        napi_throw_type_error(env, NULL, "target must be an ArrayBuffer or a typed array");
        return NULL;
    }
    if (bytes < (size_t)p.width * (size_t)p.scan_height * 4 || ((uintptr_t)data & 3u)) {
        napi_throw_range_error(env, "PTR_ERR_INVALID", "target must hold width * scanHeight * 4 bytes, 4-byte aligned");
        return NULL;
    }
    int rc = ptr_render(ctx, &p, (uint32_t *)data, NULL);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    return argv[1];
}

/* ---- renderAsync: the worker-shaped path (RT/src/RenderWorker.js:55-70) ------------------- */
typedef struct {
    ptr_ctx *ctx;
    ptr_params p;
    uint32_t *pixels;           /* the result ArrayBuffer's own memory: the frame is copied off the GPU straight into it */
    napi_ref result_ref;        /* ... and this keeps that ArrayBuffer alive until the promise is settled */
    int rc;
    char err[512];
    napi_deferred deferred;
    napi_async_work work;
    napi_ref handle_ref;        /* keeps the handle from being collected while the job runs */
    handle_cell *cell;          /* ... and this keeps an explicit destroy() from freeing the context under it */
} render_job;

static void render_execute(napi_env env, void *data)
{
    (void)env;
    render_job *j = (render_job *)data;
    j->rc = ptr_render(j->ctx, &j->p, j->pixels, NULL);
    if (j->rc != PTR_OK) {
        strncpy(j->err, ptr_last_error(j->ctx), sizeof j->err - 1);
        j->err[sizeof j->err - 1] = 0;
    }
}

static void render_complete(napi_env env, napi_status status, void *data)
{
    render_job *j = (render_job *)data;
    napi_value result;
    if (status == napi_ok && j->rc == PTR_OK) {
        if (napi_get_reference_value(env, j->result_ref, &result) == napi_ok && result) {
            napi_resolve_deferred(env, j->deferred, result);
        } else {
            napi_value msg, err;
            napi_create_string_utf8(env, "out of memory", NAPI_AUTO_LENGTH, &msg);
            napi_create_error(env, NULL, msg, &err);
            napi_reject_deferred(env, j->deferred, err);
        }
    } else {
        napi_value msg, err;
        napi_create_string_utf8(env, j->rc != PTR_OK ? j->err : "async work cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &err);
        napi_reject_deferred(env, j->deferred, err);
    }
    if (--j->cell->jobs == 0 && j->cell->closing && j->cell->ctx) {
        ptr_destroy(j->cell->ctx);         /* destroy() was called while this job ran */
        j->cell->ctx = NULL;
        j->cell->closing = 0;
    }
    napi_delete_reference(env, j->handle_ref);
    napi_delete_reference(env, j->result_ref);
    napi_delete_async_work(env, j->work);
    free(j);
}

static napi_value js_render_async(napi_env env, napi_callback_info info)
{
    napi_value argv[9];
    if (!get_args_opt(env, info, 6, 9, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    render_job *j = (render_job *)calloc(1, sizeof *j);
    if (!j) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    j->ctx = ctx;
    if (napi_get_value_external(env, argv[0], (void **)&j->cell) != napi_ok || !j->cell) { free(j); return NULL; }
    if (!read_params(env, argv + 1, &j->p) || !check_frame_size(env, &j->p)) { free(j); return NULL; }
    /* the caller's ArrayBuffer is made here, on the JavaScript thread, and filled by the job (its memory does not move) */
    napi_value result_buf;
    void *data = NULL;
    if (napi_create_arraybuffer(env, (size_t)j->p.width * (size_t)j->p.scan_height * 4, &data, &result_buf) != napi_ok ||
        napi_create_reference(env, result_buf, 1, &j->result_ref) != napi_ok) {
        free(j);
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    j->pixels = (uint32_t *)data;
    napi_value promise, name;
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok ||
        napi_create_reference(env, argv[0], 1, &j->handle_ref) != napi_ok ||
        napi_create_string_utf8(env, "ptr_render", NAPI_AUTO_LENGTH, &name) != napi_ok ||
        napi_create_async_work(env, NULL, name, render_execute, render_complete, j, &j->work) != napi_ok ||
        napi_queue_async_work(env, j->work) != napi_ok) {
        napi_delete_reference(env, j->result_ref);
        if (j->handle_ref) napi_delete_reference(env, j->handle_ref);
        free(j);
        napi_throw_error(env, NULL, "could not start the async render");
        return NULL;
    }
    j->cell->jobs++;
    return promise;
}

static napi_value js_trace_batch(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    double *org, *dir; size_t no, nd;
    if (!get_f64_array(env, argv[1], &org, &no) || !get_f64_array(env, argv[2], &dir, &nd)) return NULL;
    if (!dir || nd % 3 != 0 || (org && no != nd)) {
        napi_throw_range_error(env, "PTR_ERR_INVALID", "dirs must hold 3*n numbers and origins, if given, as many");
        return NULL;
    }
    size_t n = nd / 3;
    if (n > 0x3fffffff) { napi_throw_range_error(env, "PTR_ERR_INVALID", "too many rays"); return NULL; }
    napi_value ab_rgb, ab_el, ab_hit, ab_nrm, rgb, el, hit, nrm, out;
    void *d_rgb, *d_el, *d_hit, *d_nrm;
    NAPI_TRY(env, napi_create_arraybuffer(env, n * 3 * sizeof(double), &d_rgb, &ab_rgb));
    NAPI_TRY(env, napi_create_arraybuffer(env, n * sizeof(int32_t), &d_el, &ab_el));
    NAPI_TRY(env, napi_create_arraybuffer(env, n * 3 * sizeof(double), &d_hit, &ab_hit));
    NAPI_TRY(env, napi_create_arraybuffer(env, n * 3 * sizeof(double), &d_nrm, &ab_nrm));
    int rc = ptr_trace_batch_f64(ctx, org, dir, (int)n, (double *)d_rgb, (int32_t *)d_el, (double *)d_hit, (double *)d_nrm);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    NAPI_TRY(env, napi_create_typedarray(env, napi_float64_array, n * 3, ab_rgb, 0, &rgb));
    NAPI_TRY(env, napi_create_typedarray(env, napi_int32_array, n, ab_el, 0, &el));
    NAPI_TRY(env, napi_create_typedarray(env, napi_float64_array, n * 3, ab_hit, 0, &hit));
    NAPI_TRY(env, napi_create_typedarray(env, napi_float64_array, n * 3, ab_nrm, 0, &nrm));
    NAPI_TRY(env, napi_create_object(env, &out));
    NAPI_TRY(env, napi_set_named_property(env, out, "rgb", rgb));
    NAPI_TRY(env, napi_set_named_property(env, out, "element", el));
    NAPI_TRY(env, napi_set_named_property(env, out, "hit", hit));
    NAPI_TRY(env, napi_set_named_property(env, out, "normal", nrm));
    return out;
}

/* traceSamples(h, Float64Array origin(3), Float64Array dirs) -> {rgb, rgb8, hit, normal}: the render cache's
 * Sample.doRaytracing, batched (ptr_trace_samples_f64) */
static napi_value js_trace_samples(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    double *org, *dir; size_t no, nd;
    if (!get_f64_array(env, argv[1], &org, &no) || !get_f64_array(env, argv[2], &dir, &nd)) return NULL;
    if (!org || no != 3 || !dir || nd % 3 != 0) {
        napi_throw_range_error(env, "PTR_ERR_INVALID", "origin must hold 3 numbers and dirs 3*n");
        return NULL;
    }
    size_t n = nd / 3;
    if (n > 0x3fffffff) { napi_throw_range_error(env, "PTR_ERR_INVALID", "too many rays"); return NULL; }
    napi_value ab_rgb, ab_b, ab_hit, ab_nrm, rgb, b8, hit, nrm, out;
    void *d_rgb, *d_b, *d_hit, *d_nrm;
    NAPI_TRY(env, napi_create_arraybuffer(env, n * 3 * sizeof(double), &d_rgb, &ab_rgb));
    NAPI_TRY(env, napi_create_arraybuffer(env, n * 3, &d_b, &ab_b));
    NAPI_TRY(env, napi_create_arraybuffer(env, n * 3 * sizeof(double), &d_hit, &ab_hit));
    NAPI_TRY(env, napi_create_arraybuffer(env, n * 3 * sizeof(double), &d_nrm, &ab_nrm));
    int rc = ptr_trace_samples_f64(ctx, org, dir, (int)n, (double *)d_rgb, (uint8_t *)d_b, (double *)d_hit, (double *)d_nrm);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    NAPI_TRY(env, napi_create_typedarray(env, napi_float64_array, n * 3, ab_rgb, 0, &rgb));
    NAPI_TRY(env, napi_create_typedarray(env, napi_uint8_array, n * 3, ab_b, 0, &b8));
    NAPI_TRY(env, napi_create_typedarray(env, napi_float64_array, n * 3, ab_hit, 0, &hit));
    NAPI_TRY(env, napi_create_typedarray(env, napi_float64_array, n * 3, ab_nrm, 0, &nrm));
    NAPI_TRY(env, napi_create_object(env, &out));
    NAPI_TRY(env, napi_set_named_property(env, out, "rgb", rgb));
    NAPI_TRY(env, napi_set_named_property(env, out, "rgb8", b8));
    NAPI_TRY(env, napi_set_named_property(env, out, "hit", hit));
    NAPI_TRY(env, napi_set_named_property(env, out, "normal", nrm));
    return out;
}

/* rcReproject(h, Float64Array camera(35: from[3] view[16] proj[16]), width, height, Float64Array hits(3n, NaN x = no hit),
 *             Float64Array age(n, updated in place), ageFactor, depthThreshold)
 *   -> {itemPixel: Int32Array(n), pixelElement: Int32Array((w+2)(h+2)), pixelDepth: Float64Array((w+2)(h+2))}
 * Driver.reprojectFrame of the render cache (ptr_rc_reproject); the grid border keeps -1 / depthThreshold. */
static napi_value js_rc_reproject(napi_env env, napi_callback_info info)
{
    napi_value argv[8];
    if (!get_args(env, info, 8, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    double *cam, *hits, *age; size_t nc, nh, na;
    int32_t w, h;
    double age_factor, depth_threshold;
    if (!get_f64_array(env, argv[1], &cam, &nc) || !get_i32(env, argv[2], &w) || !get_i32(env, argv[3], &h) ||
        !get_f64_array(env, argv[4], &hits, &nh) || !get_f64_array(env, argv[5], &age, &na)) return NULL;
    if (napi_get_value_double(env, argv[6], &age_factor) != napi_ok || napi_get_value_double(env, argv[7], &depth_threshold) != napi_ok) {
        napi_throw_type_error(env, NULL, "ageFactor and depthThreshold must be numbers");
        return NULL;
    }
    if (!cam || nc != 35 || w < 1 || h < 1 || (uint64_t)(w + 2) * (uint64_t)(h + 2) > 0x3fffffffull || nh != 3 * na || na > 0x3fffffff) {
        napi_throw_range_error(env, "PTR_ERR_INVALID", "camera needs 35 numbers, hits 3 per item, age 1 per item");
        return NULL;
    }
    ptr_rc_camera c;
    memcpy(c.from, cam, 3 * sizeof(double));
    memcpy(c.view, cam + 3, 16 * sizeof(double));
    memcpy(c.proj, cam + 19, 16 * sizeof(double));
    c.width = w; c.height = h;
    size_t cells = (size_t)(w + 2) * (size_t)(h + 2);
    napi_value ab_i, ab_e, ab_d, ti, te, td, out;
    void *d_i, *d_e, *d_d;
    NAPI_TRY(env, napi_create_arraybuffer(env, (na ? na : 1) * sizeof(int32_t), &d_i, &ab_i));
    NAPI_TRY(env, napi_create_arraybuffer(env, cells * sizeof(int32_t), &d_e, &ab_e));
    NAPI_TRY(env, napi_create_arraybuffer(env, cells * sizeof(double), &d_d, &ab_d));
    for (size_t k = 0; k < cells; k++) { ((int32_t *)d_e)[k] = -1; ((double *)d_d)[k] = depth_threshold; }
    int rc = ptr_rc_reproject(ctx, &c, (int)na, hits, age, age_factor, depth_threshold, (int32_t *)d_i, (int32_t *)d_e, (double *)d_d);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    NAPI_TRY(env, napi_create_typedarray(env, napi_int32_array, na, ab_i, 0, &ti));
    NAPI_TRY(env, napi_create_typedarray(env, napi_int32_array, cells, ab_e, 0, &te));
    NAPI_TRY(env, napi_create_typedarray(env, napi_float64_array, cells, ab_d, 0, &td));
    NAPI_TRY(env, napi_create_object(env, &out));
    NAPI_TRY(env, napi_set_named_property(env, out, "itemPixel", ti));
    NAPI_TRY(env, napi_set_named_property(env, out, "pixelElement", te));
    NAPI_TRY(env, napi_set_named_property(env, out, "pixelDepth", td));
    return out;
}

static int get_typed(napi_env env, napi_value v, napi_typedarray_type want, void **data, size_t *len)
{
    bool is_ta = false;
    napi_typedarray_type tt;
    if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta ||
        napi_get_typedarray_info(env, v, &tt, len, data, NULL, NULL) != napi_ok || tt != want) {
        napi_throw_type_error(env, NULL, "typed array of the wrong kind");
        return 0;
    }
    return 1;
}

/* rcShade(h, width, height, Int32Array pixelElement, Float64Array pixelDepth, Float64Array itemAge (updated in place),
 *         Uint32Array itemRgb (r | g << 8 | b << 16), Float64Array params(4: ageFactor, interpolationZero,
 *         interpolationRandom, priorityMax))
 *   -> {pixelRgb: Uint32Array(grid), pixelPriority: Int32Array(grid), hist: Uint32Array(768), stats: Float64Array(3),
 *       colorFrame, reprojectionFrame, priorityFrame: Uint32Array(w*h)}
 * Driver.depthCulling + Driver.fillGaps + the frame getters of the render cache (ptr_rc_shade). */
static napi_value js_rc_shade(napi_env env, napi_callback_info info)
{
    napi_value argv[9];
    if (!get_args_opt(env, info, 8, 9, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int32_t w, h;
    void *pel, *pdep, *age, *rgb, *prgb_in = NULL; size_t npel, npdep, nage, nrgb, nprgb = 0;
    napi_valuetype t9;
    if (napi_typeof(env, argv[8], &t9) != napi_ok) return NULL;
    if (t9 != napi_undefined && t9 != napi_null && !get_typed(env, argv[8], napi_uint32_array, &prgb_in, &nprgb)) return NULL;
    double *par; size_t npar;
    if (!get_i32(env, argv[1], &w) || !get_i32(env, argv[2], &h) ||
        !get_typed(env, argv[3], napi_int32_array, &pel, &npel) || !get_typed(env, argv[4], napi_float64_array, &pdep, &npdep) ||
        !get_typed(env, argv[5], napi_float64_array, &age, &nage) || !get_typed(env, argv[6], napi_uint32_array, &rgb, &nrgb) ||
        !get_f64_array(env, argv[7], &par, &npar)) return NULL;
    if (w < 1 || h < 1 || (uint64_t)(w + 2) * (uint64_t)(h + 2) > 0x3fffffffull) {
        napi_throw_range_error(env, "PTR_ERR_INVALID", "bad frame size");
        return NULL;
    }
    size_t cells = (size_t)(w + 2) * (size_t)(h + 2), px = (size_t)w * (size_t)h;
    if (npel != cells || npdep != cells || nage != nrgb || nage > 0x3fffffff || !par || npar != 4) {
        napi_throw_range_error(env, "PTR_ERR_INVALID", "pixel arrays need (w+2)*(h+2) entries, itemAge and itemRgb one per item, params 4 numbers");
        return NULL;
    }
    ptr_rc_shade_params p = { w, h, par[0], par[1], par[2], par[3] };
    napi_value ab[7], ta[7], out;
    void *d[7];
    const size_t bytes[7] = { cells * 4, cells * 4, 768 * 4, 3 * 8, px * 4, px * 4, px * 4 };
    for (int k = 0; k < 7; k++) {
        NAPI_TRY(env, napi_create_arraybuffer(env, bytes[k], &d[k], &ab[k]));
        memset(d[k], 0, bytes[k]);
    }
    if (prgb_in) {                     /* optional 9th argument: the grid's colours before the pass (border cells are read) */
        if (nprgb != cells) { napi_throw_range_error(env, "PTR_ERR_INVALID", "pixelRgb needs (w+2)*(h+2) entries"); return NULL; }
        memcpy(d[0], prgb_in, cells * 4);
    }
    int rc = ptr_rc_shade(ctx, &p, (int)nage, (const int32_t *)pel, (const double *)pdep, (double *)age, (const uint32_t *)rgb,
                          (uint32_t *)d[0], (int32_t *)d[1], (uint32_t *)d[2], (double *)d[3], (uint32_t *)d[4], (uint32_t *)d[5], (uint32_t *)d[6]);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    const napi_typedarray_type types[7] = { napi_uint32_array, napi_int32_array, napi_uint32_array, napi_float64_array,
                                            napi_uint32_array, napi_uint32_array, napi_uint32_array };
    const size_t counts[7] = { cells, cells, 768, 3, px, px, px };
    const char *names[7] = { "pixelRgb", "pixelPriority", "hist", "stats", "colorFrame", "reprojectionFrame", "priorityFrame" };
    NAPI_TRY(env, napi_create_object(env, &out));
    for (int k = 0; k < 7; k++) {
        NAPI_TRY(env, napi_create_typedarray(env, types[k], counts[k], ab[k], 0, &ta[k]));
        NAPI_TRY(env, napi_set_named_property(env, out, names[k], ta[k]));
    }
    return out;
}

/* ---- the render cache's frame loop on a cache that stays on the GPU (ptr_rc_cache_*, ptr_rc_next_frame) ------------ */

/* rcCacheCreate(h, width, height, nItems) */
static napi_value js_rc_cache_create(napi_env env, napi_callback_info info)
{
    napi_value argv[4];
    int32_t w, h, n;
    if (!get_args(env, info, 4, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_i32(env, argv[1], &w) || !get_i32(env, argv[2], &h) || !get_i32(env, argv[3], &n)) return NULL;
    int rc = ptr_rc_cache_create(ctx, w, h, n);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    return NULL;
}

static napi_value js_rc_cache_destroy(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (ctx) ptr_rc_cache_destroy(ctx);
    return NULL;
}

/* the nine state arrays of rcCacheUpload / rcCacheDownload, argv[1..9]: Int32Array itemPixel(n), Float64Array itemHit(3n),
 * Float64Array itemAge(n), Uint32Array itemRgb(n), Int32Array pixelElement(cells), Float64Array pixelDepth(cells),
 * Uint32Array pixelRgb(cells), Int32Array pixelPriority(cells), Uint8Array pixelSample(cells) */
static int get_rc_state(napi_env env, ptr_ctx *ctx, napi_value *argv, void **d)
{
    int w, h, n;
    int rc = ptr_rc_cache_shape(ctx, &w, &h, &n);
    if (rc != PTR_OK) { throw_ptr(env, ctx, rc); return 0; }
    const size_t cells = (size_t)(w + 2) * (size_t)(h + 2);
    static const napi_typedarray_type types[9] = { napi_int32_array, napi_float64_array, napi_float64_array, napi_uint32_array,
                                                   napi_int32_array, napi_float64_array, napi_uint32_array, napi_int32_array, napi_uint8_array };
    size_t len[9];
    for (int k = 0; k < 9; k++)
        if (!get_typed(env, argv[1 + k], types[k], &d[k], &len[k])) return 0;
    if (len[0] != (size_t)n || len[1] != 3 * len[0] || len[2] != len[0] || len[3] != len[0] || len[4] != cells || len[5] != cells ||
        len[6] != cells || len[7] != cells || len[8] != cells) {
        napi_throw_range_error(env, "PTR_ERR_INVALID", "item arrays need n (hits 3n) entries, pixel arrays (w+2)*(h+2)");
        return 0;
    }
    return 1;
}

/* rcCacheUpload(h, <nine arrays>, cachePointer): the lengths must be those the cache was created with (the library
 * reads n_items and (w+2)*(h+2) entries): checked against ptr_rc_cache_shape */
static napi_value js_rc_cache_upload(napi_env env, napi_callback_info info)
{
    napi_value argv[11];
    void *d[9];
    double cp;
    if (!get_args(env, info, 11, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_rc_state(env, ctx, argv, d)) return NULL;
    if (napi_get_value_double(env, argv[10], &cp) != napi_ok) { napi_throw_type_error(env, NULL, "cachePointer must be a number"); return NULL; }
    int rc = ptr_rc_cache_upload(ctx, (const int32_t *)d[0], (const double *)d[1], (const double *)d[2], (const uint32_t *)d[3],
                                 (const int32_t *)d[4], (const double *)d[5], (const uint32_t *)d[6], (const int32_t *)d[7],
                                 (const uint8_t *)d[8], (int64_t)cp);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    return NULL;
}

/* rcCacheDownload(h, <nine arrays, written in place>) -> cachePointer */
static napi_value js_rc_cache_download(napi_env env, napi_callback_info info)
{
    napi_value argv[10], out;
    void *d[9];
    int64_t cp = 0;
    if (!get_args(env, info, 10, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_rc_state(env, ctx, argv, d)) return NULL;
    int rc = ptr_rc_cache_download(ctx, (int32_t *)d[0], (double *)d[1], (double *)d[2], (uint32_t *)d[3], (int32_t *)d[4],
                                   (double *)d[5], (uint32_t *)d[6], (int32_t *)d[7], (uint8_t *)d[8], &cp);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    NAPI_TRY(env, napi_create_double(env, (double)cp, &out));
    return out;
}

/* rcNextFrame(h, Float64Array view(67: from[3] matrixWorld[16] matrixWorldInverse[16] projectionMatrix[16]
 *             projectionMatrixInverse[16]), Float64Array params(6: ageFactor, depthThreshold, interpolationZero,
 *             interpolationRandom, priorityMax, maximumSamplesPerFrame))
 *   -> Float64Array(8: threshold (NaN = null), requestFactor, candidates, requests, completeness, totalPriority,
 *                      increment of Driver.totalPriority, cache usage) */
static napi_value js_rc_next_frame(napi_env env, napi_callback_info info)
{
    napi_value argv[3], ab, out;
    double *v, *p; size_t nv, np;
    void *data;
    if (!get_args(env, info, 3, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_f64_array(env, argv[1], &v, &nv) || !get_f64_array(env, argv[2], &p, &np)) return NULL;
    if (!v || nv != 67 || !p || np != 6) {
        napi_throw_range_error(env, "PTR_ERR_INVALID", "view needs 67 numbers, params 6");
        return NULL;
    }
    ptr_rc_view view;
    ptr_rc_loop_params par = { p[0], p[1], p[2], p[3], p[4], p[5] };
    ptr_rc_frame_stats st;
    memcpy(view.from, v, 3 * sizeof(double));
    memcpy(view.world, v + 3, 16 * sizeof(double));
    memcpy(view.world_inv, v + 19, 16 * sizeof(double));
    memcpy(view.proj, v + 35, 16 * sizeof(double));
    memcpy(view.proj_inv, v + 51, 16 * sizeof(double));
    int rc = ptr_rc_next_frame(ctx, &view, &par, &st);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    NAPI_TRY(env, napi_create_arraybuffer(env, 8 * sizeof(double), &data, &ab));
    const double vals[8] = { st.threshold, st.request_factor, st.candidates, st.requests, st.completeness, st.total_priority,
                             st.total_priority_increment, st.cache_usage };
    memcpy(data, vals, sizeof vals);
    NAPI_TRY(env, napi_create_typedarray(env, napi_float64_array, 8, ab, 0, &out));
    return out;
}

/* rcGetFrame(h, which (0 colour, 1 reprojection, 2 priority, 3 sampling), Uint32Array out(w*h)) */
static napi_value js_rc_get_frame(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    int32_t which;
    void *out; size_t n;
    if (!get_args(env, info, 3, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_i32(env, argv[1], &which) || !get_typed(env, argv[2], napi_uint32_array, &out, &n)) return NULL;
    int w, h, items;
    int rc = ptr_rc_cache_shape(ctx, &w, &h, &items);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    if (which < 0 || which > 3 || n != (size_t)w * (size_t)h) {
        napi_throw_range_error(env, "PTR_ERR_INVALID", "which must be 0..3 and out hold width * height words");
        return NULL;
    }
    uint32_t *f[4] = { NULL, NULL, NULL, NULL };
    f[which] = (uint32_t *)out;
    rc = ptr_rc_get_frames(ctx, f[0], f[1], f[2], f[3]);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    return NULL;
}

/* handles: a JS array of renderer handles -> ptr_ctx*[] (at most 64); 0 on a bad argument (exception pending) */
static int get_ctx_array(napi_env env, napi_value arr, ptr_ctx **out, uint32_t *n)
{
    bool is_arr = false;
    if (napi_is_array(env, arr, &is_arr) != napi_ok || !is_arr || napi_get_array_length(env, arr, n) != napi_ok || *n < 1 || *n > 64) {
        napi_throw_type_error(env, NULL, "expected an array of 1..64 renderer handles");
        return 0;
    }
    for (uint32_t k = 0; k < *n; k++) {
        napi_value v;
        if (napi_get_element(env, arr, k, &v) != napi_ok) return 0;
        out[k] = get_ctx(env, v);
        if (!out[k]) return 0;
    }
    return 1;
}

/* commInitAll(handles): one RCCL communicator over the renderers of one job, one per GPU (ptr_comm_init_all) */
static napi_value js_comm_init_all(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    ptr_ctx *ctxs[64]; uint32_t n;
    if (!get_args(env, info, 1, argv) || !get_ctx_array(env, argv[0], ctxs, &n)) return NULL;
    int rc = ptr_comm_init_all(ctxs, (int)n);
    if (rc != PTR_OK) return throw_ptr(env, ctxs[0], rc);
    return NULL;
}

/* renderMulti(handles, width, height, flags[, maxDepth, spp, seed]) -> ArrayBuffer: one whole frame rendered in cyclic
 * strips on all the renderers' GPUs and gathered on the first one's (ptr_render_multi) */
static napi_value js_render_multi(napi_env env, napi_callback_info info)
{
    napi_value argv[7];
    ptr_ctx *ctxs[64]; uint32_t n;
    if (!get_args_opt(env, info, 4, 7, argv) || !get_ctx_array(env, argv[0], ctxs, &n)) return NULL;
    ptr_params p;
    int32_t flags;
    memset(&p, 0, sizeof p);
    if (!get_i32(env, argv[1], &p.width) || !get_i32(env, argv[2], &p.height) || !get_i32(env, argv[3], &flags)) return NULL;
    p.scan_height = p.height; p.flags = (uint32_t)flags; p.spp = 1;
    napi_valuetype t;
    if (napi_typeof(env, argv[4], &t) == napi_ok && t != napi_undefined && !get_i32(env, argv[4], &p.max_depth)) return NULL;
    if (napi_typeof(env, argv[5], &t) == napi_ok && t != napi_undefined && !get_i32(env, argv[5], &p.spp)) return NULL;
    if (napi_typeof(env, argv[6], &t) == napi_ok && t != napi_undefined && napi_get_value_uint32(env, argv[6], &p.seed) != napi_ok) {
        napi_throw_type_error(env, NULL, "seed must be an unsigned 32-bit integer");
        return NULL;
    }
    if (!check_frame_size(env, &p)) return NULL;
    void *data = NULL;
    napi_value buf;
    NAPI_TRY(env, napi_create_arraybuffer(env, (size_t)p.width * (size_t)p.height * 4, &data, &buf));
    int rc = ptr_render_multi(ctxs, (int)n, &p, (uint32_t *)data);
    if (rc != PTR_OK) return throw_ptr(env, ctxs[0], rc);
    return buf;
}

static napi_value js_stats(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    ptr_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    ptr_stats st;
    int rc = ptr_get_stats(ctx, &st);
    if (rc != PTR_OK) return throw_ptr(env, ctx, rc);
    napi_value out, v;
    NAPI_TRY(env, napi_create_object(env, &out));
    const char *names[] = {"primary", "hits", "shadowRays", "exitTca", "exitD2", "exitHit", "retraced", "reshaded", "secondary"};
    const uint64_t vals[] = {st.primary, st.hits, st.shadow_rays, st.exit_tca, st.exit_d2, st.exit_hit, st.retraced, st.reshaded, st.secondary};
    for (int k = 0; k < 9; k++) {
        NAPI_TRY(env, napi_create_double(env, (double)vals[k], &v));
        NAPI_TRY(env, napi_set_named_property(env, out, names[k], v));
    }
    return out;
}

NAPI_MODULE_INIT()
{
    const struct { const char *name; napi_callback fn; } fns[] = {
        {"create", js_create}, {"destroy", js_destroy}, {"setScene", js_set_scene},
        {"setBackground", js_set_background}, {"setCamera", js_set_camera}, {"setExtension", js_set_extension}, {"render", js_render}, {"renderInto", js_render_into},
        {"renderAsync", js_render_async}, {"traceBatch", js_trace_batch}, {"traceSamples", js_trace_samples}, {"rcReproject", js_rc_reproject}, {"rcShade", js_rc_shade}, {"stats", js_stats},
        {"commInitAll", js_comm_init_all}, {"renderMulti", js_render_multi},
        {"rcCacheCreate", js_rc_cache_create}, {"rcCacheDestroy", js_rc_cache_destroy}, {"rcCacheUpload", js_rc_cache_upload},
        {"rcCacheDownload", js_rc_cache_download}, {"rcNextFrame", js_rc_next_frame}, {"rcGetFrame", js_rc_get_frame},
    };
    for (size_t k = 0; k < sizeof fns / sizeof fns[0]; k++) {
        napi_value f;
        if (napi_create_function(env, fns[k].name, NAPI_AUTO_LENGTH, fns[k].fn, NULL, &f) != napi_ok ||
            napi_set_named_property(env, exports, fns[k].name, f) != napi_ok) {
            napi_throw_error(env, NULL, "could not export the addon functions");
            return NULL;
        }
    }
    napi_value v;
    const struct { const char *name; uint32_t val; } consts[] = {
        {"QUIRK_ROW_SHIFT", PTR_QUIRK_ROW_SHIFT}, {"MODE_EXACT_F64", PTR_MODE_EXACT_F64},
        {"MODE_F32_ONLY", PTR_MODE_F32_ONLY}, {"MODE_TOLERANT", PTR_MODE_TOLERANT}, {"FLAG_STATS", PTR_FLAG_STATS}, {"ABI_VERSION", PTR_ABI_VERSION},
    };
    for (size_t k = 0; k < sizeof consts / sizeof consts[0]; k++) {
        napi_create_uint32(env, consts[k].val, &v);
        napi_set_named_property(env, exports, consts[k].name, v);
    }
    return exports;
}
