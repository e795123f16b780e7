/*
 * ptr.h — C ABI of the MI355X-native per-pixel path of raytracer-js ("path-tracer").
 *
 * This is the whole drop-in boundary: plain pointers and sizes, `int` status codes, no C++
 * or torch types.  The Node N-API addon (path-tracer_amd/js/addon/ptr_napi.c), the Python
 * ctypes binding used by tests/bench (path-tracer_amd/py/ptr_amd) and any other host bind to
 * exactly these symbols.  Each entry point cites the reference interface it replaces
 * (paths relative to the reference repo, RT/ = raytracer-js-master/).
 *
 * The reference has no FFI of its own; the boundary is its JavaScript class API
 * (SURVEY.md §8b).  The JS classes in path-tracer_amd/js/src keep that API and call down here.
 *
 * Threading: calls on one ptr_ctx are serialised by the library (internal mutex); different
 * contexts may be used from different threads.  No C++ exception crosses this boundary.
 * All device memory is owned by the library; host buffers are owned by the caller.
 */
#ifndef PTR_H
#define PTR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTR_ABI_VERSION 1

/* ---- status codes (0 ok, <0 error; text via ptr_last_error) ---------------------------- */
#define PTR_OK               0
#define PTR_ERR_INVALID     -1   /* bad argument (null pointer, non-positive size, overflow)   */
#define PTR_ERR_NO_DEVICE   -2   /* no usable gfx950 device / HIP runtime failure at create    */
#define PTR_ERR_HIP         -3   /* a HIP call failed; message carries hipGetErrorString       */
#define PTR_ERR_NO_SCENE    -4   /* render/trace before ptr_set_scene*                         */
#define PTR_ERR_UNSUPPORTED -5   /* feature bit not built into this library version            */

/* ---- ptr_params.flags ------------------------------------------------------------------- */
/* Reproduce RT/src/RayTracer.js:169 exactly: row y is stored at row (y-start_y-1), the band's
 * first row is dropped and its last stored row stays 0x00000000.  The drop-in
 * RayTracer.render() sets it; renderFrame()/benchmarks do not. */
#define PTR_QUIRK_ROW_SHIFT   0x0001u
/* Precision mode (at most one).  Default (no mode bit): HYBRID — every pixel is traced in
 * float32 with conservative uncertainty tests; pixels whose hit/miss, ordering, shadow or
 * 8-bit rounding decision is not certain are queued and re-traced by the float64 kernel,
 * which follows the reference operation for operation.  Output target: identical bytes.   */
#define PTR_MODE_EXACT_F64    0x0002u  /* whole band in float64, op-for-op the reference      */
#define PTR_MODE_F32_ONLY     0x0004u  /* float32 only, no re-trace (tolerance mode)          */
/* HYBRID within north_star's tolerance instead of identical bytes: every pixel is traced in float32 and only the
 * DISCONTINUOUS doubts (hit/miss, nearest-vs-second, far clip, shadow any-hit: a flip there moves a pixel by far more
 * than 1/255) are re-traced in float64.  A pixel whose decisions are all certain keeps its float32 colour when the summed
 * error bounds of its N.L terms keep that colour within one 8-bit unit of the float64 one (re-traced otherwise), so every
 * channel of every pixel is within 1 of the reference's byte; what the default pays for equal bytes - the 8-bit rounding
 * doubt, its error sums and the shading-only re-do - is not paid.  Reference semantics only (max_depth 0, spp 1, no planes,
 * no Phong terms).  With more than 12 spheres (the many-sphere kernel) the rounding doubt alone is traded for one test of the
 * colour's error bound (0.9 units), the other doubts and both kinds of re-do stay.  With an extension mode the flag is accepted
 * and the call renders as the default does (float64, the specification bit for bit), which meets the same bar.               */
#define PTR_MODE_TOLERANT     0x0020u
/* Accumulate work counters (ptr_get_stats) during this call; slower, not for timing.        */
#define PTR_FLAG_STATS        0x0008u
/* Bracket this call's kernels with HIP events on the launch stream (ptr_get_kernel_ms).        */
#define PTR_FLAG_TIMING       0x0010u
/* Diagnostics, with PTR_MODE_F32_ONLY: out_rgba8 receives, instead of a colour, the bit mask of
 * the reasons the hybrid mode would re-trace the pixel (1 hit/miss band, 2 origin on a surface,
 * 4 nearest-vs-second ordering, 8 far clip, 16 shadow band, 32 rounding band).               */
#define PTR_DEBUG_REASONS     0x8000u
/* Diagnostics/tests: use the many-sphere float32 kernel (sphere list staged through LDS, chosen by
 * itself above 12 spheres) whatever the sphere count.  Same results contract as the default.     */
#define PTR_DEBUG_FORCE_MANY  0x4000u
/* Diagnostics/tests: render an extension mode's float64 form with the lane-by-lane kernel (the specification written out per
 * pixel) instead of the wave-level one that sphere scenes take by default.  Same bytes: the tests compare the two.           */
#define PTR_DEBUG_EXT_LANEWISE 0x1000u

typedef struct ptr_ctx ptr_ctx;

/* Replaces the argument list of RayTracer.render(width,height,startY,scanHeight,isFirstFrame)
 * (RT/src/RayTracer.js:132).  isFirstFrame is unused by the reference and has no field.
 * max_depth/spp/seed select the EXTENSION modes (reflection/refraction bounces, stochastic
 * sampling), which the reference does not implement (MAX_RAY_DEPTH and
 * Material.reflection/.transparency are never read): no reference parity exists for them.  Their
 * specification (float64, one rounding per operation, evaluated in the order written):
 *   stack <- (origin, dir, depth 0, weight (1,1,1));  acc <- 0
 *   pop (o, d, depth, w):  nearest hit exactly as RayTracer.trace (RayTracer.js:26-41);
 *     miss: acc += w * background
 *     hit:  acc += w * emission;  N flipped to face the ray, inside = it was flipped
 *       if (transparency > 0 || reflection > 0) && depth < max_depth:
 *           f = 1 - (-(d.N));  fresnel = 1*0.1 + (f*f*f)*(1 - 0.1)
 *           if transparency > 0: eta = inside ? 1.1 : 1/1.1;  cosi = -(N.d);
 *               k = 1 - eta*eta*(1 - cosi*cosi);
 *               if k >= 0: push (P - N*1e-4, normalize(d*eta + N*(eta*cosi - sqrt(k))), depth+1,
 *                                w * surfaceColor * ((1 - fresnel) * transparency))
 *           push (P + N*1e-4, normalize(d - N*(2*(d.N))), depth+1, w * surfaceColor * fresnel)
 *       else: acc += w * (the shadowed Lambert sum of RayTracer.js:60-91)
 *   spp > 1: sample s of pixel (x, y) goes through (x + u0, y + u1), (u0, u1) = the first two words
 *   of Philox4x32-10(counter (x, y, s, 0), key (seed, 0x50545221)), u = (word + 0.5) / 2^32; the
 *   pixel is the mean of the samples (summed in sample order, one divide), then packed as ever.
 * With max_depth 0 and spp 1 a call has the reference semantics.  In the extension modes the default (no mode bit) and
 * PTR_MODE_EXACT_F64 render that specification bit for bit in float64; PTR_MODE_F32_ONLY renders it in float32, the
 * throughput form (tolerance: see tests/test_gpu_ext.py - PSNR > 50 dB, but single pixels on discontinuities and their
 * mirror images can be off by more than 1/255, which is why it is not the default; its samples are added up in blocks of 32
 * consecutive samples, then the blocks' sums in order - with at most 32 samples that is the plain sum - so that a pixel's
 * blocks can be traced by different waves: same bytes whether they are or not - the library does so when the blocks' sums
 * of the call, 16 bytes per pixel and block, fit in 2 GiB of device memory that the context then holds, and never in
 * ptr_render_device_noalloc); PTR_QUIRK_ROW_SHIFT is rejected.  */
#define PTR_MAX_DEPTH_LIMIT 16
#define PTR_MAX_SPP         65536
typedef struct ptr_params {
    int32_t  width;        /* full image width  (RayTracer.js:150,153); 1 .. 2^24 - 1          */
    int32_t  height;       /* full image height (RayTracer.js:151,153)                        */
    int32_t  start_y;      /* first scanline of the band (RayTracer.js:134-136), default 0    */
    int32_t  scan_height;  /* rows in the band (RayTracer.js:138-140), default = height       */
    int32_t  max_depth;    /* 0 = reference semantics (no bounce exists in the reference); <= 16 */
    int32_t  spp;          /* 1 = pixel centre; > 1: Philox4x32-10 jittered samples, averaged   */
    uint32_t seed;         /* Philox key word 0; unused at spp = 1                             */
    uint32_t flags;        /* PTR_QUIRK_* | PTR_MODE_* | PTR_FLAG_*                           */
} ptr_params;

/* Camera.  The reference hot path ignores the user's Camera object and hard-codes origin 0,
 * fov 30 degrees, looking down -Z (RT/src/RayTracer.js:150-155,171-174); that is
 * the default.  dir = normalize(xx*right + yy*up + forward) with xx,yy as RayTracer.js:171-172.
 * tan_half_fov is the reference's `angle` (= Math.tan(Math.PI*0.5*fov/180)), computed by the
 * host language so that its libm decides the bits, exactly as in the reference.            */
typedef struct ptr_camera {
    double origin[3];
    double right[3];
    double up[3];
    double forward[3];
    double tan_half_fov;
} ptr_camera;

/* Work counters of the last call that had PTR_FLAG_STATS set.  Same counting rule as the
 * algorithmic-FLOP model of SURVEY.md §8(d).                                                */
typedef struct ptr_stats {
    uint64_t primary;       /* primary rays traced                                            */
    uint64_t hits;          /* primary rays that hit an element                               */
    uint64_t shadow_rays;   /* one per (hit pixel, emissive element)                          */
    uint64_t exit_tca;      /* Sphere.intersect calls leaving at tca<0   (RT/src/Sphere.js:18) */
    uint64_t exit_d2;       /* ... at d2>radius2                          (RT/src/Sphere.js:23) */
    uint64_t exit_hit;      /* ... returning true                         (RT/src/Sphere.js:32) */
    uint64_t retraced;      /* hybrid mode: pixels re-traced in float64 (last call, always set) */
    uint64_t reshaded;      /* hybrid mode: pixels whose nearest element was certain and only the
                               shading was redone in float64 (last call, always set)            */
    uint64_t secondary;     /* extension modes: reflection + refraction rays (primary counts camera rays) */
} ptr_stats;

/* ---- lifetime ---------------------------------------------------------------------------- */

/* Replaces `new RayTracer(backgroundColor, scene[, camera])` (RT/src/RayTracer.js:12-16) as the
 * owner of per-renderer state.  device = HIP device ordinal.  Fails (PTR_ERR_NO_DEVICE) when no
 * GPU is present: there is no CPU fallback.                                                  */
int  ptr_create(int device, ptr_ctx **out);
void ptr_destroy(ptr_ctx *ctx);

/* Text of the last error on ctx (or, with ctx == NULL, of the calling thread's last failed
 * ptr_create).  Never NULL.                                                                   */
const char *ptr_last_error(ptr_ctx *ctx);

int ptr_abi_version(void);

/* ---- scene upload ------------------------------------------------------------------------ */

/* Replaces Scene.add()/getElements() + Sphere/Material field reads done by trace()
 * (RT/src/Scene.js:9-15, RT/src/RayTracer.js:22,29,60-66) and the worker wire format
 * (RT/src/Sphere.js:53-83, RT/src/RenderWorker.js:27-33).
 * soa = 12 planes of n values, in the order
 *     cx cy cz radius   sr sg sb   reflection transparency   er eg eb
 * scene order preserved (it decides ties and the shadow loop's first blocker).  radius2 and the
 * light list are derived natively.  The library copies; it keeps no reference to soa.
 * n == 0 is allowed (every pixel is background).
 * JavaScript numbers are float64, so the _f64 form is the one the drop-in classes use; the
 * float form promotes each value to double first.                                            */
int ptr_set_scene(ptr_ctx *ctx, const float *soa, int n_spheres);
int ptr_set_scene_f64(ptr_ctx *ctx, const double *soa, int n_spheres);

/* A sequence of n_frames scenes with the same sphere count (animation loops of
 * RT/examples/example2.js:98-113 and example3.js:89-98 re-render the scene with moved centres).
 * soa = n_frames consecutive 12-plane blocks.  ptr_set_scene* is the n_frames == 1 case.      */
int ptr_set_scene_sequence_f64(ptr_ctx *ctx, const double *soa, int n_spheres, int n_frames);

/* Replaces the backgroundColor constructor argument (RT/src/RayTracer.js:13, returned on a miss
 * at :43-46).  Unclamped value; the pack clamps.                                             */
int ptr_set_background(ptr_ctx *ctx, const float rgb[3]);
int ptr_set_background_f64(ptr_ctx *ctx, const double rgb[3]);

/* cam == NULL restores the reference's fixed camera (RT/src/RayTracer.js:150-155).            */
int ptr_set_camera(ptr_ctx *ctx, const ptr_camera *cam);

/* Planes and Phong terms: EXTENSION scene data.  north_star names a ray-plane primitive and Phong shading; the reference
 * has neither (its only primitive is RT/src/Sphere.js:15-33, its shading Lambert, RT/src/RayTracer.js:87-89; SURVEY.md
 * F2), so, like the bounce loop, this is specified here and self-pinned (tests/golden/ext, "parity unpinned").  A scene
 * with planes or Phong terms is rendered by the extension kernels whatever max_depth / spp are; RayTracer.render's
 * reference path is untouched without them.  Specification (float64, evaluated left to right, on top of ptr_params'):
 *   elements = the spheres in scene order, then the planes (element n_spheres + k).  Plane k = point p, UNIT normal n.
 *   Plane.intersect(o, d):  denom = d.n;  denom == 0: miss;  t = ((p - o).n) / denom;  t < 0: miss;  else t0 = t
 *   nearest hit: all spheres, then all planes, strict <.  A plane hit's normal is n (then turned to face the ray, which
 *     sets `inside`, like any other).  Shadow rays test the planes too, after the spheres, but a plane blocks only IN FRONT of
 *     the light: t0*t0 < |C_i - P|^2 (the squared length normalize() computes; a sphere keeps the reference's distance-blind
 *     any-hit rule, RayTracer.js:74-85 - for an infinite plane that rule would darken everything that faces it).  A plane
 *     with emission > 0 is a light at p.  Lights: emissive spheres in scene order, then emissive planes.
 *   Phong: element e has ks >= 0 and an integer exponent m in 1..65536.  In the Lambert sum, after light i's diffuse term:
 *     if ks > 0:  Rv = d - N*(2*(d.N));  s = max(0, Rv.L);  ph = 1; b = s; q = m; while q > 0 { if q odd: ph = ph*b;  b = b*b;
 *       q >>= 1 };  dif += tr * (emission_i * (ks * ph))      (per channel; tr = the light's transmission, 0 or 1)
 * planes_soa = 14 rows of n_planes values: px py pz nx ny nz  sr sg sb  reflection transparency  er eg eb (NULL / 0: no
 * planes).  phong = ks[n_elements] then exponent[n_elements], n_elements = n_spheres + n_planes as they are at the next
 * render (NULL: no Phong terms).  Replaces the previous extension data as a whole; (NULL, 0, NULL, 0) clears it.
 * The single-ray entries (ptr_trace_*) refuse scenes with extension data.                                        */
int ptr_set_extension_f64(ptr_ctx *ctx, const double *planes_soa, int n_planes, const double *phong, int n_elements);

/* ---- render ------------------------------------------------------------------------------ */

/* Replaces RayTracer.render() (RT/src/RayTracer.js:132-197): ray generation, trace, clamp,
 * RGBA8 pack over the band [start_y, start_y+scan_height).  Synchronous.
 * out_rgba8: caller-allocated HOST buffer of width*scan_height uint32 (A<<24|B<<16|G<<8|R),
 * fully written.  out_rgba32f (nullable): HOST buffer of width*scan_height*4 floats receiving
 * the unclamped trace colour (r,g,b,1), never row-shifted.  Renders frame 0 of a sequence.    */
int ptr_render(ptr_ctx *ctx, const ptr_params *p, uint32_t *out_rgba8, float *out_rgba32f);

/* Same work, but outputs are DEVICE pointers (e.g. a torch tensor's data_ptr) and the call only
 * enqueues on `stream` (a hipStream_t, NULL = the context's own stream) without synchronising.
 * Calls on one context may use different streams and may follow ptr_set_* calls without a host wait in between: the
 * library orders a call's kernels, and a scene upload, behind the kernels of the context's previous call.
 * Renders frames [first_frame, first_frame+n_frames) of the uploaded sequence back to back into
 * d_rgba8 (n_frames * width * scan_height uint32) with one launch per kernel.
 * This is the entry bench.py times: inputs resident in HBM, outputs left in HBM.              */
int ptr_render_device(ptr_ctx *ctx, const ptr_params *p, int first_frame, int n_frames,
                      void *d_rgba8, void *d_rgba32f /* nullable */, void *stream);

/* Cyclic-strip form of the reference's band split (RT/src/RenderPlanner.js:75-76 gives worker i
 * the contiguous band [i*floor(h/n), (i+1)*floor(h/n)); the example frame is ~3x costlier at the
 * bottom than at the top, so for multi-GPU balance the rows are dealt out in strips instead).
 * The image rows are cut into strips of strip_height rows; strip k belongs to part k % n_parts.
 * This call renders the rows of `part`, in increasing y, compactly: p->scan_height must be the
 * number of such rows (ptr_strip_rows), p->start_y must be 0, PTR_QUIRK_ROW_SHIFT is not allowed.
 * frame_stride_px = distance in pixels between consecutive frames in the outputs
 * (0 = width*scan_height), so that equally sized buffers can be gathered across ranks.          */
typedef struct ptr_strips {
    int32_t n_parts;       /* >= 1 (1 = the whole image)                                       */
    int32_t part;          /* 0 .. n_parts-1                                                   */
    int32_t strip_height;  /* >= 1                                                             */
    int32_t reserved;      /* 0                                                                */
} ptr_strips;
int ptr_strip_rows(const ptr_strips *s, int height);
/* image row of local row `local_row` of the part (the rule ptr_render_strips_device renders by and the gather places
 * rows by); < 0: bad arguments.  No GPU needed: hosts that collect the parts themselves use it.                   */
int ptr_strip_image_row(const ptr_strips *s, int local_row);
int ptr_render_strips_device(ptr_ctx *ctx, const ptr_params *p, const ptr_strips *s,
                             int first_frame, int n_frames, void *d_rgba8, void *d_rgba32f,
                             uint64_t frame_stride_px, void *stream);

/* ---- multi-GPU: collecting the parts of a frame on one GPU ------------------------------------------------------ */
/* Replaces the collection of the workers' bands, RT/src/RenderPlanner.js:85-100 (onmessage -> onUpdateReceived) and
 * RT/examples/example3.js:66-86 (pieces put side by side into the canvas): every GPU renders one part of the cyclic
 * strip partition (ptr_render_strips_device) and the parts meet in ONE frame buffer on the root GPU, over RCCL
 * (xGMI): a direct gather - every peer sends over its own link to the root, grouped ncclSend / ncclRecv - followed by
 * one copy kernel on the root that puts the strips at their rows.  RCCL (librccl.so.1) is loaded when the first of
 * these calls needs it; rendering on one GPU never touches it.
 *
 * A communicator joins the contexts of one job, one context per GPU:
 *   one process per GPU (torchrun style): rank 0 calls ptr_comm_unique_id, hands the 128 bytes to the other
 *     processes by whatever channel it has, and every process calls ptr_comm_init_rank on its context;
 *   one process, several GPUs (the Node host): ptr_comm_init_all on the array of contexts (rank = index).
 * n_ranks == 1 is allowed (nothing is sent).  ptr_destroy ends a context's membership; ptr_comm_destroy does so early. */
#define PTR_COMM_ID_BYTES 128
#define PTR_COMM_MAX_RANKS 64      /* ranks of one communicator (ptr_comm_init_rank / _all, ptr_gather_strips_all, ptr_render_multi) */
/* Sizes of one gather, as ptr_gather_strips_device / _all work them out (pure host arithmetic, no GPU, no context): for a
 * host that allocates the buffers, and for checking the 64-bit arithmetic at sizes no test box holds (8 ranks x 2 048
 * frames).  Replaces nothing in the reference (its bands are separate Uint8ClampedArrays, RenderPlanner.js:85-100).       */
typedef struct ptr_gather_sizes {
    int32_t  rows_max;          /* rows of the largest part (part 0): every rank's buffer holds this many per frame       */
    int32_t  rows;              /* rows of this part                                                                      */
    uint64_t frame_stride_px;   /* pixels between two frames of a rank's compact buffer                                   */
    uint64_t count;             /* pixels a rank sends = frame_stride_px * n_frames (the ncclSend / ncclRecv count)       */
    uint64_t local_bytes;       /* bytes of a rank's compact buffer                                                       */
    uint64_t root_stage_bytes;  /* bytes of the root's staging buffer: (n_parts - 1) * local_bytes                        */
    uint64_t full_bytes;        /* bytes of the whole frames on the root                                                  */
    uint64_t last_full_index;   /* pixel index, in the whole frames, of this part's last pixel of the last frame          */
} ptr_gather_sizes;
int ptr_gather_plan(int width, int height, int strip_height, int n_frames, uint64_t frame_stride_px, int part, int n_parts,
                    ptr_gather_sizes *out);
int ptr_comm_unique_id(unsigned char id[PTR_COMM_ID_BYTES]);
int ptr_comm_init_rank(ptr_ctx *ctx, const unsigned char id[PTR_COMM_ID_BYTES], int rank, int n_ranks);
int ptr_comm_init_all(ptr_ctx *const *ctxs, int n_ctxs);
int ptr_comm_destroy(ptr_ctx *ctx);

/* Gather on `stream` (enqueue only).  This context's rank is its part of the partition, the number of ranks the
 * number of parts.  d_local: what ptr_render_strips_device wrote for this part - n_frames frames, frame_stride_px
 * pixels apart (0 = the rows of part 0, the largest part, times width: the same value on every rank, as the equally
 * sized messages need).  d_full (root only, others pass NULL): n_frames * height * width RGBA8 words, every row at its
 * place.  The root keeps a staging buffer of (n_ranks - 1) local buffers, grown on demand.  Collective: every rank of
 * the communicator calls it with the same sizes.                                                                  */
int ptr_gather_strips_device(ptr_ctx *ctx, int width, int height, int strip_height, int n_frames,
                             const void *d_local, uint64_t frame_stride_px, void *d_full, int root, void *stream);
/* The copy step of the gather on its own, for hosts that move the parts themselves (another collective library, PCIe):
 * the rows of part s->part, as ptr_render_strips_device wrote them at d_part (n_frames frames, frame_stride_px apart, 0 =
 * this part's rows times width), go to their places in the whole frames at d_full.  No communicator needed.  Enqueues on
 * `stream`.                                                                                                        */
int ptr_place_strips_device(ptr_ctx *ctx, const ptr_strips *s, int width, int height, int n_frames,
                            const void *d_part, uint64_t frame_stride_px, void *d_full, void *stream);
/* The same for a communicator made by ptr_comm_init_all, all ranks from one thread: d_locals[i] belongs to
 * ctxs[i], the root is ctxs[0], every part travels on its context's own stream (ptr_sync(ctxs[0]) waits for the frame). */
int ptr_gather_strips_all(ptr_ctx *const *ctxs, int n_ctxs, int width, int height, int strip_height, int n_frames,
                          const void *const *d_locals, uint64_t frame_stride_px, void *d_full);

/* RayTracer.render for one frame on several GPUs, host buffer out - what RenderPlanner does with its workers
 * (RT/src/RenderPlanner.js:27-47,85-100), in one call: ctxs[i] renders part i of the 8-row cyclic strip partition of frame
 * 0 of ITS scene (the caller gives every context the same scene, background and camera, as RenderPlanner.js:61-74 sends
 * them to every worker), the parts are gathered on ctxs[0]'s GPU and copied to out_rgba8 (width * height words).
 * Needs a communicator over exactly these contexts (ptr_comm_init_all).  p->start_y must be 0, p->scan_height ==
 * p->height, no PTR_QUIRK_ROW_SHIFT.  Synchronous.                                                                */
int ptr_render_multi(ptr_ctx *const *ctxs, int n_ctxs, const ptr_params *p, uint32_t *out_rgba8);

/* ptr_render_device for callers that cannot allocate.  No counterpart in the reference: it exists for callers that
 * may not call hipMalloc at this point - code under HIP graph capture, tight serving loops, harnesses that forbid it.
 * Same arguments, same kernels and the same output bytes as ptr_render_device, but the call allocates nothing (no
 * hipMalloc / hipFree / event creation, in any mode): the float32 kernels' re-trace queue, its counters and the work
 * counters live in a 32 MiB array that is part of the library's code object, one copy per GPU, held until the process
 * ends.  Limits and rules:
 *   - the hybrid mode needs 8 bytes of queue per pixel of the call's tiles (12 with more than 12 spheres): about 4 M pixels per call at most (one
 *     1920x1080 frame fits; width * scan_height * n_frames beyond that returns PTR_ERR_INVALID, nothing is launched);
 *   - the scene must already be resident: call ptr_sync() after the last ptr_set_scene* / ptr_set_background* /
 *     ptr_set_camera (the upload is what allocates); otherwise PTR_ERR_INVALID;
 *   - PTR_FLAG_TIMING is rejected (it creates events); PTR_FLAG_STATS works;
 *   - all contexts on one GPU share the array, and every call clears the part that must start at zero: two such calls
 *     that overlap in time on one GPU (different streams or contexts) race; calls on one stream are safe back to back. */
int ptr_render_device_noalloc(ptr_ctx *ctx, const ptr_params *p, int first_frame, int n_frames,
                              void *d_rgba8, void *d_rgba32f /* nullable */, void *stream);

/* Block until everything enqueued on the context's own stream has finished.  Also uploads a scene, background or
 * camera that was set since the last render (the render calls otherwise do that on their first use).  Work enqueued on
 * a caller's stream is the caller's to wait for.                                               */
int ptr_sync(ptr_ctx *ctx);

/* ---- single rays ------------------------------------------------------------------------- */

/* Batched RayTracer.requestPixel(rayDir) (RT/src/RayTracer.js:98-115): origin = camera origin,
 * out_rgb = 3 bytes per ray (Math.round(min(1,c)*255) & 0xff).  out_hit_xyz / out_normal_xyz
 * (nullable, 3 floats per ray) receive the hit point and shading normal, zeros on a miss —
 * what the render cache's Sample needs (rendercache-js/src/Sample.js:49-75).                   */
int ptr_trace_batch(ptr_ctx *ctx, const float *dirs_xyz, int n,
                    uint8_t *out_rgb, float *out_hit_xyz, float *out_normal_xyz);

/* Batched RayTracer.trace(rayOrigin, rayDir, depth) (RT/src/RayTracer.js:18-95) in float64,
 * op-for-op: out_rgb64 = unclamped colour (3 doubles per ray).  origins == NULL means the
 * camera origin for every ray.  out_element (nullable) = index of the hit element or -1;
 * out_hit_xyz / out_normal_xyz (nullable) as above but double.  All HOST pointers.           */
int ptr_trace_batch_f64(ptr_ctx *ctx, const double *origins_xyz, const double *dirs_xyz, int n,
                        double *out_rgb64, int32_t *out_element,
                        double *out_hit_xyz, double *out_normal_xyz);

/* Batched Sample.doRaytracing(raytracer, from, request) of the render cache (rendercache-js/src/Sample.js:49-75
 * -> Raytracer.trace, rendercache-js/src/Raytracer.js:24-123), float64, op-for-op.  Every ray starts at
 * origin_xyz (the cache's camera.from).  The colour arithmetic is that of RayTracer.trace; the outputs follow
 * the render cache: out_rgb64 (nullable) = the unclamped colour trace() returns, out_rgb8 (nullable) = the three
 * Math.round(Math.min(1, c) * 255) values of sample.color; out_hit_xyz (nullable) = sample.hit — the hit point,
 * or origin + dir * 1e5 on a miss (Raytracer.js:62-68); out_normal_xyz (nullable) = sample.normalDir — the
 * element's outward normal normalized a second time (Raytracer.js:75-76, NOT turned towards the ray), or -hit
 * on a miss.  At least one of out_rgb64 / out_rgb8 must be given.  All HOST pointers.                    */
int ptr_trace_samples_f64(ptr_ctx *ctx, const double origin_xyz[3], const double *dirs_xyz, int n,
                          double *out_rgb64, uint8_t *out_rgb8, double *out_hit_xyz, double *out_normal_xyz);

/* ---- render cache: reprojection pass ------------------------------------------------- */

/* The camera state Camera.reprojectPixel reads (rendercache-js/src/Camera.js:52-74): camera.from and the two
 * matrices of its THREE.PerspectiveCamera at that moment, as THREE.Matrix4.elements (column-major):
 * view = cam.matrixWorldInverse, proj = cam.projectionMatrix; width/height = camera.scope.                 */
typedef struct ptr_rc_camera {
    double  from[3];
    double  view[16];
    double  proj[16];
    int32_t width, height;
} ptr_rc_camera;

/* Replaces Driver.reprojectFrame() (rendercache-js/src/Driver.js:417-486, with detachFrameFromCache :405-416 and
 * Camera.reprojectPixel), float64, same results as the reference's sequential walk over the cache:
 *   hit_xyz        3 doubles per cache item (cacheItem.hit); NaN in x = the item has no hit
 *   age            cacheItem.age, updated in place (+age_factor loser / displaced winner, +2 outside the frame,
 *                  +4 reprojection failed, i.e. depth 0)
 *   item_pixel     out: cacheItem.pixel as y * (width + 2) + x on the padded grid of Driver.allocBuffers, or -1
 *                  (an item displaced by a nearer one later in cache order keeps its pointer, as in the reference)
 *   pixel_element  out: pixel.element as a cache index or -1;  pixel_depth  out: pixel.depth
 * The pixel arrays hold (width + 2) * (height + 2) entries; only the interior 1..width x 1..height is written.
 * age_factor = Driver.ageFactor (1), depth_threshold = Driver.depthThreshold (1e6).  No scene is needed.
 * ptr_rc_reproject takes HOST pointers and waits; ptr_rc_reproject_device takes DEVICE pointers and only enqueues
 * on `stream` (NULL = the context's stream).                                                                   */
int ptr_rc_reproject(ptr_ctx *ctx, const ptr_rc_camera *camera, int n_items, const double *hit_xyz, double *age,
                     double age_factor, double depth_threshold, int32_t *item_pixel, int32_t *pixel_element,
                     double *pixel_depth);
int ptr_rc_reproject_device(ptr_ctx *ctx, const ptr_rc_camera *camera, int n_items, const void *d_hit_xyz, void *d_age,
                            double age_factor, double depth_threshold, void *d_item_pixel, void *d_pixel_element,
                            void *d_pixel_depth, void *stream);

/* The same pass for a cache whose items are split over several GPUs in cache order (BASELINE configs[4]: "point
 * reprojection at 2/4/8 GPUs"): rank r holds items [item_base, item_base + n_local) of the whole cache.  Results are
 * those of Driver.reprojectFrame over the WHOLE cache: every rank ends with its own items' age and item_pixel and with
 * the whole grid (pixel_element = index in the whole cache).  One exchange step: the ranks' local grids, 12 B per pixel.
 *   ptr_rc_reproject_shard_device   the local pass alone: local grid into d_local_element / d_local_depth (indices in
 *                                   the whole cache; interior only);
 *   ptr_rc_merge_shards_device      after the caller gathered all ranks' local grids (d_all_element / d_all_depth:
 *                                   n_ranks grids one after the other, rank order): the age of a local winner that
 *                                   loses against another rank, pixel pointers of items an earlier rank's item hides,
 *                                   and the whole grid.  Must follow this context's shard call (it reads the local
 *                                   pass's scratch);
 *   ptr_rc_reproject_multi_device   both around an ncclAllGather of the two grids on the context's communicator
 *                                   (ptr_comm_init_rank / _all; one rank: no RCCL needed).  Collective: every rank of
 *                                   the communicator calls it.
 * DEVICE pointers, enqueued on `stream` (NULL = the context's stream).                                          */
int ptr_rc_reproject_shard_device(ptr_ctx *ctx, const ptr_rc_camera *camera, int n_local, int item_base,
                                  const void *d_hit_xyz, void *d_age, double age_factor, double depth_threshold,
                                  void *d_item_pixel, void *d_local_element, void *d_local_depth, void *stream);
int ptr_rc_merge_shards_device(ptr_ctx *ctx, int width, int height, int n_ranks, int rank, const void *d_all_element,
                               const void *d_all_depth, int n_local, int item_base, void *d_age, double age_factor,
                               double depth_threshold, void *d_item_pixel, void *d_pixel_element, void *d_pixel_depth,
                               void *stream);
int ptr_rc_reproject_multi_device(ptr_ctx *ctx, const ptr_rc_camera *camera, int n_local, int item_base,
                                  const void *d_hit_xyz, void *d_age, double age_factor, double depth_threshold,
                                  void *d_item_pixel, void *d_pixel_element, void *d_pixel_depth, void *stream);

/* ---- render cache: depth culling, gap filling, frames ----------------------------------- */

typedef struct ptr_rc_shade_params {
    int32_t width, height;          /* camera.scope                                                         */
    double  age_factor;             /* Driver.ageFactor (1)                                                 */
    double  interpolation_zero;     /* Driver.interpolationZero (20)                                        */
    double  interpolation_random;   /* Driver.interpolationRandom (5)                                       */
    double  priority_max;           /* Driver.priorityMax (255)                                             */
} ptr_rc_shade_params;

/* Replaces Driver.depthCulling() (rendercache-js/src/Driver.js:488-554), Driver.fillGaps() (:556-679) and the frame
 * getters getColorFrame / getReprojectionFrame / getPriorityFrame (:960-1018), on the state ptr_rc_reproject leaves:
 *   pixel_element / pixel_depth   in, padded grid (width + 2) * (height + 2)
 *   item_age        in/out: + 10 age factors where a pixel's depth differs by more than 10 % from the mean depth of
 *                   its diagonal neighbours (:536)
 *   item_rgb        in: cacheItem.color as bytes, r | g << 8 | b << 16 (what Sample.doRaytracing leaves)
 *   pixel_rgb       out, grid: pixel.color in the same packing (the element's colour, or the weighted mean of the
 *                   neighbours that have one, corner 1 / edge 2, floored; 0 when none has)
 *   pixel_priority  out, grid: pixel.priority (age >> 1 / round(mean age + zero - items * random) / priority_max);
 *                   pixel.weight is 0 for every pixel in the reference (:236), so that term is 0 here too
 *   hist            out: 3 x 256 bins, priorities[INTERPOLATED], [URGENT], [SAMPLED]; priorities outside 0..255 are
 *                   not counted (the reference's arrays hold NaN properties for those)
 *   stats           out: statistics.completeness, statistics.totalPriority, increment of this.totalPriority
 *   *_frame         out, nullable, width * height RGBA8 words each, rows bottom-up as the reference stores them
 * Only the grid's interior is written.  A border cell may carry an element too (Driver.initializeCache draws pixel
 * positions from 0, and nothing ever detaches the border): its pixel_depth and pixel_rgb are read as the caller
 * passes them, its age is the element's.  Preconditions: ages are finite integers, colours are bytes.
 * ptr_rc_shade: HOST pointers, waits.  ptr_rc_shade_device: DEVICE pointers, enqueues on `stream`; d_hist = 768
 * uint32, d_totals = 3 int64 (pixels with an element, totalPriority, this.totalPriority increment).            */
int ptr_rc_shade(ptr_ctx *ctx, const ptr_rc_shade_params *params, int n_items, const int32_t *pixel_element,
                 const double *pixel_depth, double *item_age, const uint32_t *item_rgb, uint32_t *pixel_rgb,
                 int32_t *pixel_priority, uint32_t *hist, double stats[3], uint32_t *color_frame,
                 uint32_t *reprojection_frame, uint32_t *priority_frame);
int ptr_rc_shade_device(ptr_ctx *ctx, const ptr_rc_shade_params *params, int n_items, const void *d_pixel_element,
                        const void *d_pixel_depth, void *d_item_age, const void *d_item_rgb, void *d_pixel_rgb,
                        void *d_pixel_priority, void *d_hist, void *d_totals, void *d_color_frame,
                        void *d_reprojection_frame, void *d_priority_frame, void *stream);

/* ---- render cache: the frame loop on a cache that stays on the GPU (SURVEY.md 8f ranks 1-4 together) ----------- */

/* Driver.nextFrame() (rendercache-js/src/Driver.js:95-115) with the cache (Driver.cache, Sample.js) and the pixel grid
 * (Driver.buffer, Pixel.js) resident in HBM: per frame the camera state goes down and eight numbers come back; the
 * frames come up when a getter asks for them.  Same state after every frame as the reference's sequential loop:
 *   reprojectFrame (:417-486) and depthCulling + fillGaps (:488-679)   the passes of ptr_rc_reproject / ptr_rc_shade;
 *   directSamples (:682-827)    threshold over the priority histograms (:690-706), candidate pixels in the serpentine
 *                               walk order, picks at Math.round of a running sum of the request factor, the error
 *                               diffusion of distributeExceedingPriority (:829-839) with the stale pixel of :813-814;
 *   addRequest / allocCacheItem (:252-326)   pages of 8 items, the first empty one or the oldest, Sample.clear();
 *   requestSamples (:842-863)   Camera.computeDirToXY (Camera.js:39-50) + the trace of ptr_trace_samples_f64, colour
 *                               bytes and hit point into the item, the colour into the item's pixel;
 *   age (:936-942) of every item, getCacheUsage, the frame getters (:960-1046).
 * The first frame's initializeCache (:328-353) draws pixels from Math.random and stays with the caller: fill the cache
 * on the host (ptr_trace_samples_f64), then hand the state over with ptr_rc_cache_upload.
 *
 * Layout (as ptr_rc_reproject / ptr_rc_shade): n_items cache items — item_pixel (grid index or -1), item_hit_xyz
 * (3 doubles, NaN in x = none), item_age, item_rgb (r | g << 8 | b << 16); the padded grid of (width + 2) * (height + 2)
 * pixels — pixel_element (cache index or -1), pixel_depth, pixel_rgb, pixel_priority, pixel_sample (0 / 1).
 * cache_pointer = Driver.cachePointer.  One cache per context; it uses the context's scene (ptr_set_scene*: spheres
 * only, as the reference) and stream.  n_items >= 8, priority_max an integer in 0..255.                            */
typedef struct ptr_rc_view {
    double from[3];                 /* camera.from                                                           */
    double world[16];               /* cam.matrixWorld.elements (column-major, as THREE.Matrix4 stores them) */
    double world_inv[16];           /* cam.matrixWorldInverse                                                */
    double proj[16];                /* cam.projectionMatrix                                                  */
    double proj_inv[16];            /* cam.projectionMatrixInverse                                           */
} ptr_rc_view;

typedef struct ptr_rc_loop_params {
    double age_factor;              /* Driver.ageFactor (1)                                                  */
    double depth_threshold;         /* Driver.depthThreshold (1e6)                                           */
    double interpolation_zero;      /* Driver.interpolationZero (20)                                         */
    double interpolation_random;    /* Driver.interpolationRandom (5)                                        */
    double priority_max;            /* Driver.priorityMax (255)                                              */
    double max_samples_per_frame;   /* Driver.maximumSamplesPerFrame (scope.x * scope.y / ratio)             */
} ptr_rc_loop_params;

typedef struct ptr_rc_frame_stats { /* Driver.statistics after the frame                                     */
    double threshold;               /* NaN = null                                                            */
    double request_factor, candidates, requests, completeness, total_priority;
    double total_priority_increment;/* what the frame added to Driver.totalPriority                          */
    double cache_usage;             /* getCacheUsage()                                                       */
} ptr_rc_frame_stats;

int  ptr_rc_cache_create(ptr_ctx *ctx, int width, int height, int n_items);
void ptr_rc_cache_destroy(ptr_ctx *ctx);
int  ptr_rc_cache_shape(ptr_ctx *ctx, int *width, int *height, int *n_items);     /* what the cache was created with */
/* HOST pointers, all required; both wait. */
int  ptr_rc_cache_upload(ptr_ctx *ctx, const int32_t *item_pixel, const double *item_hit_xyz, const double *item_age,
                         const uint32_t *item_rgb, const int32_t *pixel_element, const double *pixel_depth,
                         const uint32_t *pixel_rgb, const int32_t *pixel_priority, const uint8_t *pixel_sample,
                         int64_t cache_pointer);
int  ptr_rc_cache_download(ptr_ctx *ctx, int32_t *item_pixel, double *item_hit_xyz, double *item_age, uint32_t *item_rgb,
                           int32_t *pixel_element, double *pixel_depth, uint32_t *pixel_rgb, int32_t *pixel_priority,
                           uint8_t *pixel_sample, int64_t *cache_pointer);
/* One frame; waits for it.  The reference's pick loop does not end when the request factor is 0 while candidates
 * exist (every priority outside 0..255): that state is refused with PTR_ERR_INVALID, the cache is then as the
 * shading pass left it.                                                                                         */
int  ptr_rc_next_frame(ptr_ctx *ctx, const ptr_rc_view *view, const ptr_rc_loop_params *params, ptr_rc_frame_stats *out);
/* The candidate numbers Driver.directSamples requests (:795-822: `index += requestFactor`, Math.round, no number twice,
 * none >= n_candidates), as ptr_rc_next_frame computes them: in closed form, piece by piece of the float64 running sum,
 * instead of adding it up.  Host only (no GPU is touched).  Returns how many were written to out, PTR_ERR_INVALID for a
 * request factor with which the reference's loop would not end or a capacity that is too small, PTR_ERR_UNSUPPORTED when
 * the sum has more pieces than the table holds (ptr_rc_next_frame then adds it up step by step).                  */
int  ptr_rc_pick_sequence(double request_factor, int n_candidates, int32_t *out, int capacity);
/* getColorFrame / getReprojectionFrame / getPriorityFrame / getSamplingFrame (:960-1046) from the live state: width *
 * height RGBA8 words each, rows bottom-up, HOST pointers, any may be NULL.                                       */
int  ptr_rc_get_frames(ptr_ctx *ctx, uint32_t *color_frame, uint32_t *reprojection_frame, uint32_t *priority_frame,
                       uint32_t *sampling_frame);

/* ---- introspection ----------------------------------------------------------------------- */
int ptr_get_stats(ptr_ctx *ctx, ptr_stats *out);

/* Device time of the kernels of the last call that had PTR_FLAG_TIMING set, from HIP events
 * recorded on the stream the kernels were launched on: main_ms = the per-pixel kernel
 * (float32 kernel in hybrid / f32-only mode, float64 kernel in exact mode), retrace_ms = the
 * float64 re-trace of queued pixels (0 outside hybrid mode).  Waits for those kernels.        */
int ptr_get_kernel_ms(ptr_ctx *ctx, float *main_ms, float *retrace_ms);

#ifdef __cplusplus
}
#endif
#endif /* PTR_H */
