// ptr_api_comm.hip — host side of the C ABI for several GPUs: RCCL (resolved at run time), the gather of cyclic strips to a
// root GPU, the single-process forms over several contexts, and the reprojection of a render cache split over the ranks.
// Declarations and reference sites: include/ptr.h.
#include "ptr_host.h"

// ---- multi-GPU gather over RCCL ---------------------------------------------------------------------------------
namespace ptrh {

// The few RCCL entry points the gather needs, resolved at run time from librccl.so.1 (the process may already have one
// loaded - PyTorch brings its own - and then that instance is the one used; a host without RCCL can still render on one GPU).
// Types as in <rccl/rccl.h>: ncclComm_t is a pointer, ncclUniqueId 128 bytes passed by value, results and data types ints.
struct RcclId { char internal[PTR_COMM_ID_BYTES]; };
struct Rccl {
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    char why[256] = "";
};
constexpr int RCCL_UINT32 = 3;      // ncclUint32 (rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3)
constexpr int RCCL_INT32 = 2, RCCL_FLOAT64 = 8;

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) { snprintf(r.why, sizeof r.why, "cannot load librccl.so.1: %s", dlerror()); return; }
        bool all = true;
        auto sym = [&](const char* name) { void* p = dlsym(h, name); if (!p) { all = false; snprintf(r.why, sizeof r.why, "librccl.so.1 lacks %s", name); } return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = all;
    });
    return r;
}

#define RCCL_TRY(c, call)                                                                       \
    do {                                                                                        \
        int r_ = (call);                                                                        \
        if (r_ != 0) return fail((c), PTR_ERR_HIP, "%s failed: %s", #call, rccl().GetErrorString(r_)); \
    } while (0)

struct GatherShape { int rows_max, rows; uint64_t stride; size_t count; };

// sizes of one gather; fails on arguments that do not describe one.  part / n_parts = the context's rank / ranks.
int gather_shape(ptr_ctx* c, int width, int height, int strip_height, int n_frames, uint64_t frame_stride_px, int part, int n_parts,
                 GatherShape* g)
{
    if (width < 1 || height < 1 || strip_height < 1 || n_frames < 1 || n_frames > 65535 || width >= (1 << 24))
        return fail(c, PTR_ERR_INVALID, "gather: bad sizes (width %d, height %d, strip_height %d, n_frames %d)", width, height, strip_height, n_frames);
    const ptr_strips s0{n_parts, 0, strip_height, 0}, sp{n_parts, part, strip_height, 0};
    g->rows_max = ptr_strip_rows(&s0, height);
    g->rows = ptr_strip_rows(&sp, height);
    if (g->rows_max < 0 || g->rows < 0 || g->rows_max > 65535) return fail(c, PTR_ERR_INVALID, "gather: bad strip partition");
    g->stride = frame_stride_px ? frame_stride_px : (uint64_t)g->rows_max * (uint64_t)width;
    if (g->stride < (uint64_t)g->rows_max * (uint64_t)width)
        return fail(c, PTR_ERR_INVALID, "gather: frame_stride_px %llu is smaller than the largest part (%d rows of %d pixels)",
                    (unsigned long long)g->stride, g->rows_max, width);
    g->count = (size_t)g->stride * (size_t)n_frames;
    if ((uint64_t)width * (uint64_t)height * (uint64_t)n_frames >= (1ull << 40)) return fail(c, PTR_ERR_INVALID, "gather: too many pixels");
    return PTR_OK;
}

// the root's copy step for the rows of `part` that lie at src
int gather_place(ptr_ctx* root, const uint32_t* src, uint32_t* full, int width, int height, int strip_height, int n_frames,
                 const GatherShape& g, int part, int n_parts, hipStream_t s)
{
    GatherArgs a;
    a.src = src; a.full = full; a.src_frame_stride_px = g.stride;
    a.width = width; a.height = height; a.n_frames = n_frames;
    a.strip_h = strip_height; a.n_parts = n_parts; a.part = part;
    const ptr_strips sp{n_parts, part, strip_height, 0};
    a.rows = ptr_strip_rows(&sp, height);
    HIP_TRY(root, launch_gather_rows(a, s));
    return PTR_OK;
}

}  // namespace ptrh

int ptr_gather_plan(int width, int height, int strip_height, int n_frames, uint64_t frame_stride_px, int part, int n_parts,
                    ptr_gather_sizes* out)
{
    if (!out || n_parts < 1 || n_parts > PTR_COMM_MAX_RANKS || part < 0 || part >= n_parts) return PTR_ERR_INVALID;
    GatherShape g;
    const int rc = gather_shape(nullptr, width, height, strip_height, n_frames, frame_stride_px, part, n_parts, &g);
    if (rc != PTR_OK) return rc;
    out->rows_max = g.rows_max; out->rows = g.rows;
    out->frame_stride_px = g.stride;
    out->count = (uint64_t)g.count;
    out->local_bytes = (uint64_t)g.count * sizeof(uint32_t);
    out->root_stage_bytes = (uint64_t)(n_parts - 1) * (uint64_t)g.count * sizeof(uint32_t);
    out->full_bytes = (uint64_t)width * (uint64_t)height * (uint64_t)n_frames * sizeof(uint32_t);
    // where the last pixel of this part's last frame lands in the whole frames (what the copy kernel addresses, in pixels)
    const ptr_strips sp{n_parts, part, strip_height, 0};
    const int last_row = g.rows > 0 ? ptr_strip_image_row(&sp, g.rows - 1) : 0;
    out->last_full_index = ((uint64_t)(n_frames - 1) * (uint64_t)height + (uint64_t)last_row) * (uint64_t)width + (uint64_t)(width - 1);
    return PTR_OK;
}

int ptr_comm_unique_id(unsigned char id[PTR_COMM_ID_BYTES])
{
    if (!id) return fail(nullptr, PTR_ERR_INVALID, "id is NULL");
    Rccl& r = rccl();
    if (!r.ok) return fail(nullptr, PTR_ERR_UNSUPPORTED, "%s", r.why);
    RcclId u;
    RCCL_TRY(nullptr, r.GetUniqueId(&u));
    std::memcpy(id, u.internal, PTR_COMM_ID_BYTES);
    return PTR_OK;
}

int ptr_comm_init_rank(ptr_ctx* c, const unsigned char id[PTR_COMM_ID_BYTES], int rank, int n_ranks)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (n_ranks < 1 || n_ranks > PTR_COMM_MAX_RANKS || rank < 0 || rank >= n_ranks)
        return fail(c, PTR_ERR_INVALID, "rank %d of %d (at most %d ranks)", rank, n_ranks, PTR_COMM_MAX_RANKS);
    if (c->comm_n) return fail(c, PTR_ERR_INVALID, "the context already belongs to a communicator (ptr_comm_destroy first)");
    if (n_ranks > 1) {
        if (!id) return fail(c, PTR_ERR_INVALID, "id is NULL");
        Rccl& r = rccl();
        if (!r.ok) return fail(c, PTR_ERR_UNSUPPORTED, "%s", r.why);
        HIP_TRY(c, hipSetDevice(c->device));
        RcclId u;
        std::memcpy(u.internal, id, PTR_COMM_ID_BYTES);
        RCCL_TRY(c, r.CommInitRank(&c->comm, n_ranks, u, rank));
    }
    c->comm_rank = rank; c->comm_n = n_ranks;
    return PTR_OK;
}

int ptr_comm_init_all(ptr_ctx* const* ctxs, int n)
{
    if (!ctxs || n < 1 || n > PTR_COMM_MAX_RANKS || !ctxs[0]) return fail(nullptr, PTR_ERR_INVALID, "ptr_comm_init_all: bad context list");
    for (int i = 0; i < n; i++) {                      // (messages go to ctxs[0], the context a host asks afterwards)
        if (!ctxs[i]) return fail(ctxs[0], PTR_ERR_INVALID, "ptr_comm_init_all: context %d is NULL", i);
        if (ctxs[i]->comm_n) return fail(ctxs[0], PTR_ERR_INVALID, "context %d already belongs to a communicator", i);
        for (int j = 0; j < i; j++)
            if (ctxs[j]->device == ctxs[i]->device)
                return fail(ctxs[0], PTR_ERR_INVALID, "contexts %d and %d are on the same GPU (%d): one rank per GPU", j, i, ctxs[i]->device);
    }
    if (n > 1) {
        Rccl& r = rccl();
        if (!r.ok) return fail(ctxs[0], PTR_ERR_UNSUPPORTED, "%s", r.why);
        void* comms[64];
        int devs[64];
        for (int i = 0; i < n; i++) devs[i] = ctxs[i]->device;
        RCCL_TRY(ctxs[0], r.CommInitAll(comms, n, devs));
        for (int i = 0; i < n; i++) ctxs[i]->comm = comms[i];
    }
    for (int i = 0; i < n; i++) { ctxs[i]->comm_rank = i; ctxs[i]->comm_n = n; }
    return PTR_OK;
}

int ptr_comm_destroy(ptr_ctx* c)
{
    if (!c) return PTR_ERR_INVALID;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)rccl().CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->comm_n = 0; c->comm_rank = 0;
    return PTR_OK;
}

// ---- reprojection of a cache split over several GPUs ---------------------------------------------------------------

int ptr_rc_reproject_multi_device(ptr_ctx* c, const ptr_rc_camera* cam, int n_local, int item_base, const void* d_hit_xyz, void* d_age,
                                  double age_factor, double depth_threshold, void* d_item_pixel, void* d_pixel_element,
                                  void* d_pixel_depth, void* stream)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (c->comm_n < 1) return fail(c, PTR_ERR_INVALID, "the context belongs to no communicator (ptr_comm_init_rank / _all)");
    PTR_TRY(rc_check(c, cam, n_local, d_hit_xyz, d_age, d_item_pixel, d_pixel_element, d_pixel_depth));
    PTR_TRY(rc_shard_check(c, n_local, item_base));
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    const int n = c->comm_n, me = c->comm_rank;
    const size_t cells = (size_t)(cam->width + 2) * (size_t)(cam->height + 2);
    PTR_TRY(grow(c, c->rcm_all_el, (size_t)n * cells * sizeof(int32_t)));
    PTR_TRY(grow(c, c->rcm_all_dep, (size_t)n * cells * sizeof(double)));
    int32_t* all_el = static_cast<int32_t*>(c->rcm_all_el.ptr);
    double* all_dep = static_cast<double*>(c->rcm_all_dep.ptr);
    // the local pass writes straight into this rank's place of the gathered grids when it is alone; otherwise into a
    // buffer of its own (ncclAllGather's send buffer must not lie inside the receive buffer except at its own offset)
    int32_t* loc_el = all_el + (size_t)me * cells;
    double* loc_dep = all_dep + (size_t)me * cells;
    PTR_TRY(rc_enqueue(c, cam, n_local, static_cast<const double*>(d_hit_xyz), static_cast<double*>(d_age), age_factor, depth_threshold,
                       static_cast<int32_t*>(d_item_pixel), loc_el, loc_dep, s, item_base));
    if (n > 1) {
        Rccl& r = rccl();
        if (!r.ok) return fail(c, PTR_ERR_UNSUPPORTED, "%s", r.why);
        RCCL_TRY(c, r.GroupStart());                                    // in place: each rank's grid already sits at its offset
        const int rc1 = r.AllGather(loc_el, all_el, cells, RCCL_INT32, c->comm, s);
        const int rc2 = rc1 == 0 ? r.AllGather(loc_dep, all_dep, cells, RCCL_FLOAT64, c->comm, s) : 0;
        const int rc3 = r.GroupEnd();
        if (rc1 != 0 || rc2 != 0 || rc3 != 0)
            return fail(c, PTR_ERR_HIP, "ncclAllGather failed: %s", r.GetErrorString(rc1 ? rc1 : rc2 ? rc2 : rc3));
    }
    return rc_merge_enqueue(c, cam->width, cam->height, n, me, all_el, all_dep, n_local, item_base, d_age, age_factor, depth_threshold,
                            d_item_pixel, d_pixel_element, d_pixel_depth, s);
}

int ptr_gather_strips_device(ptr_ctx* c, int width, int height, int strip_height, int n_frames,
                             const void* d_local, uint64_t frame_stride_px, void* d_full, int root, void* stream)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->comm_n < 1) return fail(c, PTR_ERR_INVALID, "gather: the context belongs to no communicator (ptr_comm_init_rank / _all)");
    const int n = c->comm_n, me = c->comm_rank;
    if (root < 0 || root >= n) return fail(c, PTR_ERR_INVALID, "gather: root %d of %d ranks", root, n);
    if (!d_local || (me == root && !d_full)) return fail(c, PTR_ERR_INVALID, "gather: a required buffer is NULL");
    GatherShape g;
    PTR_TRY(gather_shape(c, width, height, strip_height, n_frames, frame_stride_px, me, n, &g));
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    uint32_t* stage = nullptr;
    if (me == root && n > 1) {
        PTR_TRY(grow(c, c->gather_stage, (size_t)(n - 1) * g.count * sizeof(uint32_t)));
        stage = static_cast<uint32_t*>(c->gather_stage.ptr);
    }
    if (n > 1) {
        // a direct gather: every peer sends over its own xGMI link to the root; one group, so the transfers run side by side
        Rccl& r = rccl();
        RCCL_TRY(c, r.GroupStart());
        int rc = 0;
        if (me == root) {
            int slot = 0;
            for (int p = 0; p < n && rc == 0; p++)
                if (p != root) rc = r.Recv(stage + (size_t)(slot++) * g.count, g.count, RCCL_UINT32, p, c->comm, s);
        } else {
            rc = r.Send(d_local, g.count, RCCL_UINT32, root, c->comm, s);
        }
        const int rc2 = r.GroupEnd();
        if (rc != 0 || rc2 != 0) return fail(c, PTR_ERR_HIP, "gather: ncclSend/ncclRecv failed: %s", r.GetErrorString(rc ? rc : rc2));
    }
    if (me == root) {
        int slot = 0;
        for (int p = 0; p < n; p++) {
            const uint32_t* src = p == root ? static_cast<const uint32_t*>(d_local) : stage + (size_t)(slot++) * g.count;
            PTR_TRY(gather_place(c, src, static_cast<uint32_t*>(d_full), width, height, strip_height, n_frames, g, p, n, s));
        }
    }
    return PTR_OK;
}

int ptr_place_strips_device(ptr_ctx* c, const ptr_strips* st, int width, int height, int n_frames,
                            const void* d_part, uint64_t frame_stride_px, void* d_full, void* stream)
{
    if (!c) return PTR_ERR_INVALID;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!st || st->n_parts < 1 || st->part < 0 || st->part >= st->n_parts) return fail(c, PTR_ERR_INVALID, "place: bad strip description");
    if (!d_part || !d_full) return fail(c, PTR_ERR_INVALID, "place: a required buffer is NULL");
    GatherShape g;
    PTR_TRY(gather_shape(c, width, height, st->strip_height, n_frames, 0, st->part, st->n_parts, &g));
    g.stride = frame_stride_px ? frame_stride_px : (uint64_t)g.rows * (uint64_t)width;
    if (g.stride < (uint64_t)g.rows * (uint64_t)width) return fail(c, PTR_ERR_INVALID, "place: frame_stride_px smaller than the part");
    HIP_TRY(c, hipSetDevice(c->device));
    return gather_place(c, static_cast<const uint32_t*>(d_part), static_cast<uint32_t*>(d_full), width, height, st->strip_height,
                        n_frames, g, st->part, st->n_parts, stream ? static_cast<hipStream_t>(stream) : c->stream);
}

int ptr_gather_strips_all(ptr_ctx* const* ctxs, int n, int width, int height, int strip_height, int n_frames,
                          const void* const* d_locals, uint64_t frame_stride_px, void* d_full)
{
    if (!ctxs || n < 1 || n > PTR_COMM_MAX_RANKS || !ctxs[0]) return PTR_ERR_INVALID;
    ptr_ctx* root = ctxs[0];
    if (!d_locals || !d_full) return fail(root, PTR_ERR_INVALID, "gather: a required buffer is NULL");
    for (int i = 0; i < n; i++)
        if (!ctxs[i] || ctxs[i]->comm_n != n || ctxs[i]->comm_rank != i || !d_locals[i])
            return fail(root, PTR_ERR_INVALID, "gather: context %d is not rank %d of a communicator of %d (ptr_comm_init_all), or its buffer is NULL", i, i, n);
    GatherShape g;
    PTR_TRY(gather_shape(root, width, height, strip_height, n_frames, frame_stride_px, 0, n, &g));
    uint32_t* stage = nullptr;
    if (n > 1) {
        HIP_TRY(root, hipSetDevice(root->device));
        {
            std::lock_guard<std::mutex> lk(root->mu);          // (the staging buffer belongs to the root's context)
            PTR_TRY(grow(root, root->gather_stage, (size_t)(n - 1) * g.count * sizeof(uint32_t)));
            stage = static_cast<uint32_t*>(root->gather_stage.ptr);
        }
        Rccl& r = rccl();
        RCCL_TRY(root, r.GroupStart());
        int rc = 0;
        for (int i = 1; i < n && rc == 0; i++) {
            rc = r.Recv(stage + (size_t)(i - 1) * g.count, g.count, RCCL_UINT32, i, root->comm, root->stream);
            if (rc == 0) rc = r.Send(d_locals[i], g.count, RCCL_UINT32, 0, ctxs[i]->comm, ctxs[i]->stream);
        }
        const int rc2 = r.GroupEnd();
        if (rc != 0 || rc2 != 0) return fail(root, PTR_ERR_HIP, "gather: ncclSend/ncclRecv failed: %s", r.GetErrorString(rc ? rc : rc2));
    }
    HIP_TRY(root, hipSetDevice(root->device));
    for (int i = 0; i < n; i++) {
        const uint32_t* src = i == 0 ? static_cast<const uint32_t*>(d_locals[0]) : stage + (size_t)(i - 1) * g.count;
        PTR_TRY(gather_place(root, src, static_cast<uint32_t*>(d_full), width, height, strip_height, n_frames, g, i, n, root->stream));
    }
    return PTR_OK;
}

int ptr_render_multi(ptr_ctx* const* ctxs, int n, const ptr_params* p, uint32_t* out_rgba8)
{
    if (!ctxs || n < 1 || n > PTR_COMM_MAX_RANKS || !ctxs[0]) return PTR_ERR_INVALID;
    ptr_ctx* root = ctxs[0];
    if (!p || !out_rgba8) return fail(root, PTR_ERR_INVALID, "ptr_render_multi: params / out_rgba8 is NULL");
    if (p->start_y != 0 || p->scan_height != p->height || (p->flags & PTR_QUIRK_ROW_SHIFT))
        return fail(root, PTR_ERR_INVALID, "ptr_render_multi renders whole frames: start_y 0, scan_height == height, no PTR_QUIRK_ROW_SHIFT");
    for (int i = 0; i < n; i++)
        if (!ctxs[i] || ctxs[i]->comm_n != n || ctxs[i]->comm_rank != i)
            return fail(root, PTR_ERR_INVALID, "ptr_render_multi: context %d is not rank %d of a communicator of %d (ptr_comm_init_all)", i, i, n);
    constexpr int STRIP = 8;
    const ptr_strips s0{n, 0, STRIP, 0};
    const int rows_max = ptr_strip_rows(&s0, p->height);
    if (rows_max < 1) return fail(root, PTR_ERR_INVALID, "ptr_render_multi: bad height %d", p->height);
    const uint64_t stride = (uint64_t)rows_max * (uint64_t)p->width;
    const void* locals[PTR_COMM_MAX_RANKS];
    for (int i = 0; i < n; i++) {
        ptr_ctx* c = ctxs[i];
        std::lock_guard<std::mutex> lk(c->mu);
        const ptr_strips sp{n, i, STRIP, 0};
        ptr_params q = *p;
        q.scan_height = ptr_strip_rows(&sp, p->height);
        HIP_TRY(c, hipSetDevice(c->device));
        PTR_TRY(grow(c, c->multi_local, (size_t)stride * sizeof(uint32_t)));
        locals[i] = c->multi_local.ptr;
        if (q.scan_height < 1) continue;                       // more GPUs than strips: nothing to render here
        int rc = check_params(c, &q, 0, 1);
        if (rc == PTR_OK) rc = enqueue_render(c, &q, 0, 1, static_cast<uint32_t*>(c->multi_local.ptr), nullptr, c->stream, &sp, stride);
        if (rc != PTR_OK) { if (c != root) fail(root, rc, "ptr_render_multi, GPU %d: %s", c->device, c->err); return rc; }
    }
    const size_t px = (size_t)p->width * (size_t)p->height;
    {
        std::lock_guard<std::mutex> lk(root->mu);
        HIP_TRY(root, hipSetDevice(root->device));
        PTR_TRY(grow(root, root->multi_full, px * sizeof(uint32_t)));
    }
    PTR_TRY(ptr_gather_strips_all(ctxs, n, p->width, p->height, STRIP, 1, locals, stride, root->multi_full.ptr));
    std::lock_guard<std::mutex> lk(root->mu);
    HIP_TRY(root, hipSetDevice(root->device));
    PTR_TRY(copy_to_host(root, out_rgba8, root->multi_full.ptr, px * sizeof(uint32_t), root->stream));
    for (int i = 1; i < n; i++) {                              // the peers' sends are done once the root has everything; leave no work behind
        HIP_TRY(ctxs[i], hipSetDevice(ctxs[i]->device));
        HIP_TRY(ctxs[i], hipStreamSynchronize(ctxs[i]->stream));
    }
    return PTR_OK;
}

