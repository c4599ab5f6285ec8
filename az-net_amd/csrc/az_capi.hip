// az_capi.hip -- the C ABI of libaznet_hip.so (include/aznet_hip.h): context lifecycle, head and feature maps, the two
// lanes and the public launch / fetch entry points, measurement switches, the native exchange.  The forms of a search live
// in az_search.hip, the unit entry points in az_units.hip.  There is no CPU fallback anywhere in this library: without a
// gfx950 device az_create fails with AZ_ERR_NO_DEVICE.
#include "az_ctx.h"

extern "C" {

const char *az_version(void) { return AZ_VERSION_STR; }
int az_abi_sizes(void) { return (int)(sizeof(az_params) & 0xffff) | (int)((sizeof(az_stats) & 0xffff) << 16); }

int az_create(int device, az_ctx **out)
{
    if (!out) return AZ_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return AZ_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return AZ_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return AZ_ERR_NO_DEVICE;   // gfx950 code objects only
    if (hipSetDevice(device) != hipSuccess) return AZ_ERR_NO_DEVICE;
    az_ctx *c = new az_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return AZ_ERR_HIP; }
    if (hipHostMalloc((void **)&c->h_cnt, RES_HDR + (size_t)AZ_TOPK_MAX * 36) != hipSuccess) { delete c; return AZ_ERR_HIP; }
    for (int i = 0; i < 3; ++i)
        if (hipHostMalloc((void **)&c->h_res[i], RES_HDR + (size_t)AZ_TOPK_MAX * 36) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_res[i], hipEventDisableTiming) != hipSuccess) { az_destroy(c); return AZ_ERR_HIP; }
    *out = c;
    return AZ_OK;
}

static void destroy_twin(az_ctx *c);
static void destroy_batch(az_ctx *c);

int az_destroy(az_ctx *c)
{
    if (!c) return AZ_ERR_INVALID;
    destroy_twin(c);
    destroy_batch(c);
    if (c->comm) { if (c->comm_stream) hipStreamSynchronize(c->comm_stream); azk_rccl_destroy(c->comm); c->comm = nullptr; }
    if (c->comm_stream) { hipStreamDestroy(c->comm_stream); c->comm_stream = nullptr; }
    for (auto &e : c->comm_ev) if (e) { hipEventDestroy(e); e = nullptr; }
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->stream2) { hipStreamSynchronize(c->stream2); hipStreamDestroy(c->stream2); c->stream2 = nullptr; }
    if (c->stream3) { hipStreamSynchronize(c->stream3); hipStreamDestroy(c->stream3); c->stream3 = nullptr; }
    for (hipEvent_t *e : {&c->ev_h6, &c->ev_i7, &c->ev_s2, &c->ev_tail, &c->ev_s3, &c->ev_geo[0], &c->ev_geo[1]})
        if (*e) { hipEventDestroy(*e); *e = nullptr; }
    clear_events(c);
    for (hipEvent_t ev : c->event_pool) hipEventDestroy(ev);
    c->event_pool.clear();
    for (auto &g : c->graphs) hipGraphExecDestroy(g.second.exec);
    c->graphs.clear();
    free_all(c);
    for (void *p : c->allocs_geom) hipFree(p);
    c->allocs_geom.clear();
    for (void *p : c->allocs_det) hipFree(p);
    c->allocs_det.clear();
    for (auto *q : c->plans) { free_plan(q); delete q; }
    for (auto &e : c->spec_store) for (void *q2 : {(void *)e.urois, (void *)e.B1, (void *)e.choff, (void *)e.Udev}) if (q2) hipFree(q2);
    c->spec_store.clear();
    c->plans.clear();
    if (c->feat_owned[0]) { for (float *f : c->feat_owned) hipFree(f); hipFree(c->feat_stage); }
    for (void *p : {c->ev_a, c->ev_b, c->ev_c, c->ev_d, c->ev_e, c->ev_f, c->ev_g, c->ev_h, (void *)c->hisB,
                    (void *)c->hisZ, (void *)c->pool, (void *)c->pool_tmp, (void *)c->pool_n, (void *)c->pool_hist})
        if (p) hipFree(p);
    if (c->nms_dets) { hipFree(c->nms_dets); hipFree(c->nms_sdets); hipFree(c->nms_order); hipFree(c->nms_mask); hipFree(c->nms_keep); hipFree(c->nms_rank); }
    if (c->h_cnt) hipHostFree(c->h_cnt);
    for (int i = 0; i < 3; ++i) {
        if (c->h_res[i]) hipHostFree(c->h_res[i]);
        if (c->ev_res[i]) hipEventDestroy(c->ev_res[i]);
    }
    for (auto &q : c->io) {
        if (q.host) hipHostFree(q.host);
        if (q.dev) hipFree(q.dev);
        if (q.ev) hipEventDestroy(q.ev);
    }
    c->io.clear();
    if (c->spare.ev) { hipEventDestroy(c->spare.ev); c->spare.ev = nullptr; }
    if (c->ev_hand) hipEventDestroy(c->ev_hand);
    if (c->ev_copy) hipEventDestroy(c->ev_copy);
    if (c->span_ring) hipFree(c->span_ring);
    if (c->h_nms) hipHostFree(c->h_nms);
    if (c->h_nmsb) hipHostFree(c->h_nmsb);
    if (c->h_nmsg) hipHostFree(c->h_nmsg);
    if (c->nms_done) hipFree(c->nms_done);
    if (c->h_Y) { hipHostFree(c->h_Y); hipHostFree(c->h_S); }
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return AZ_OK;
}

const char *az_last_error(const az_ctx *c) { return c ? c->err.c_str() : "null context"; }

void *az_stream(az_ctx *c) { return c ? (void *)c->stream : nullptr; }

int az_set_limits(az_ctx *c, int max_regions, int max_candidates)
{
    if (!c || max_regions < 64 || max_candidates < max_regions) return fail(c, AZ_ERR_INVALID, "bad limits");
    if (c->head_loaded || c->geom_ready) return fail(c, AZ_ERR_STATE, "az_set_limits must precede the first use of the context");
    c->maxR = max_regions;
    c->maxCand = max_candidates;
    c->maxCh = 4 * max_regions;
    return AZ_OK;
}

int az_set_gemm_mode(az_ctx *c, int parts)
{
    if (!c || !(parts == 0 || parts == 2 || parts == 3)) return fail(c, AZ_ERR_INVALID, "az_set_gemm_mode: 0, 2 or 3");
    if (c->head_loaded) return fail(c, AZ_ERR_STATE, "az_set_gemm_mode must precede az_load_head");
    c->gemm_parts = parts;
    return AZ_OK;
}

int az_load_head(az_ctx *c, int C, int n6, int n71, int n72, const float *W6, const float *b6,
                 const float *W71, const float *b71, const float *W72, const float *b72, const float *Was,
                 const float *bas, const float *Wab, const float *bab, const float *Wz, const float *bz)
{
    if (!c) return AZ_ERR_INVALID;
    if (!W6 || !b6 || !W71 || !b71 || !W72 || !b72 || !Was || !bas || !Wab || !bab || !Wz || !bz)
        return fail(c, AZ_ERR_INVALID, "az_load_head: null weight pointer");
    if (C <= 0 || (C & 3) || n6 <= 0 || (n6 & 3) || n71 <= 0 || (n71 & 3) || n72 <= 0 || (n72 & 3))
        return fail(c, AZ_ERR_INVALID, "az_load_head: C, n6, n71, n72 must be positive multiples of 4");
    if (azk_tail_lds_bytes(n71 + n72) > 64 * 1024)
        return fail(c, AZ_ERR_INVALID, "az_load_head: n71 + n72 too large for the tail kernel's LDS tile");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
    if (c->stream3) HIPCHK(c, hipStreamSynchronize(c->stream3));
    c->s2_live = false; c->i7_live = false; c->part7 = nullptr;
    for (float *&q : c->part7_ring) q = nullptr;          // (freed with the head's other buffers below)
    c->part7_turn = 0;
    c->s3_live = false; c->g_live[0] = c->g_live[1] = false;
    destroy_twin(c);                          // (the second lane reads this head's buffers: rebuilt at the next launch)
    destroy_batch(c);
    free_all(c);
    c->spare.ready = false; c->spare.ev_live = false;      // (its buffers were in the list free_all walked; the slots are gone)
    c->head_loaded = false;
    {
        int rg = ensure_geom(c);
        if (rg) return rg;
    }
    AzHeadDims &d = c->d;
    d.C = C; d.pooled = 7; d.K6 = C * 49; d.n6 = n6; d.n71 = n71; d.n72 = n72; d.n7 = n71 + n72;
    d.H = d.W = 0;
    c->S6 = azk_fc_split(d.K6);
    c->S7 = azk_fc_split(d.n6);
    const size_t R = (size_t)c->maxR;
    int rc;
#define A(p, n) if ((rc = dalloc(c, &c->p, (n))) != AZ_OK) return rc
    A(W6, azk_tiled_elems(n6, d.K6)); A(b6, n6); A(W7, azk_tiled_elems(d.n7, n6)); A(b7, d.n7);
    A(Wt, 64 * azk_tail_weight_rows(d.n7)); A(bt, 64);
    A(pool5, R * d.K6);
    {
        const size_t p6 = (size_t)c->S6 * R * n6, p7 = (size_t)c->S7 * R * d.n7;
        const size_t pm = p6 > p7 ? p6 : p7;
        const size_t pw = (size_t)n6 * d.K6;          // also stages W6 for the column permutation
        A(part, pm > pw ? pm : pw);
    }
    A(h6, R * n6); A(h7, R * d.n7);
    A(part7, (size_t)c->S7 * R * d.n7);
    if (c->gemm_parts) { A(W6p, (size_t)c->gemm_parts * azk_weight_plane_elems(n6, d.K6)); A(pool5p, (size_t)c->gemm_parts * azk_act_plane_elems((int)R, d.K6)); A(gscale, 4);
        HIPCHK(c, hipMemsetAsync(c->pool5p, 0, (size_t)c->gemm_parts * azk_act_plane_elems((int)R, d.K6) * 2, c->stream)); }
#undef A
    // Weights: Caffe [out, in] row-major is already the K-contiguous "B^T" layout the GEMM reads.
    // int6 reads pool5, which this library keeps bin-major ([p][c], see az_head.hip): permute
    // W6's columns to match (c*49 + p  ->  p*C + c).  `part` is big enough to stage it.
    // The GEMM streams weights tile-major (azk_tile_weights): permute / stack in a row-major temporary, then tile.
    // (`tmp` serves both layers: the larger of the two row-major blocks)
    struct TmpGuard { float *p = nullptr; ~TmpGuard() { if (p) hipFree(p); } } tg;
    {
        size_t te = (size_t)n6 * d.K6;
        if ((size_t)d.n7 * n6 > te) te = (size_t)d.n7 * n6;
        HIPCHK(c, hipMalloc((void **)&tg.p, te * 4));
    }
    float *tmp = tg.p;
    HIPCHK(c, hipMemcpy(c->part, W6, (size_t)n6 * d.K6 * 4, hipMemcpyHostToDevice));
    azk_permute_k(c->stream, c->part, tmp, n6, C, 1);
    if (c->gemm_parts) {          // 16-bit terms of the (permuted, row-major) int6 weights
        c->w6_scale = 0.f;
        if (c->gemm_parts == 2) {
            // two fp16 terms: the weights are scaled by the power of two that brings max |w| into [2^14, 2^15)
            float mx = 0.f;
            for (size_t i = 0, n = (size_t)n6 * d.K6; i < n; ++i) { const float a = fabsf(W6[i]); if (a > mx) mx = a; }
            c->w6_scale = 1.f;
            if (mx > 0.f && mx < INFINITY) { int e; (void)frexpf(mx, &e); c->w6_scale = ldexpf(1.f, 15 - e); }
            HIPCHK(c, hipMemsetAsync(c->gscale, 0, 4 * sizeof(float), c->stream));
        }
        azk_split_weight_planes(c->stream, tmp, c->W6p, n6, d.K6, c->gemm_parts, c->w6_scale);
    }
    azk_tile_weights(c->stream, tmp, c->W6, n6, d.K6);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->b6, b6, (size_t)n6 * 4, hipMemcpyHostToDevice));
    // int7_1 and int7_2 both read int6: one GEMM with the two weight blocks stacked along N.
    HIPCHK(c, hipMemcpy(tmp, W71, (size_t)n71 * n6 * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(tmp + (size_t)n71 * n6, W72, (size_t)n72 * n6 * 4, hipMemcpyHostToDevice));
    azk_tile_weights(c->stream, tmp, c->W7, d.n7, n6);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->b7, b71, (size_t)n71 * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->b7 + n71, b72, (size_t)n72 * 4, hipMemcpyHostToDevice));
    // tail weights, k-major [n7][64]: outputs 0..10 adj_score, 11..54 adj_bbox (k < n71), output 55
    // zoom_score (k >= n71); everything else zero
    {
        std::vector<float> wt(64 * azk_tail_weight_rows(d.n7), 0.f);
        for (int o = 0; o < 11; ++o)
            for (int k = 0; k < n71; ++k) wt[(size_t)k * 64 + o] = Was[(size_t)o * n71 + k];
        for (int o = 0; o < 44; ++o)
            for (int k = 0; k < n71; ++k) wt[(size_t)k * 64 + 11 + o] = Wab[(size_t)o * n71 + k];
        for (int k = 0; k < n72; ++k) wt[(size_t)(n71 + k) * 64 + 55] = Wz[k];
        HIPCHK(c, hipMemcpy(c->Wt, wt.data(), wt.size() * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMemset(c->bt, 0, 64 * 4));
    HIPCHK(c, hipMemcpy(c->bt, bas, 11 * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->bt + 11, bab, 44 * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->bt + 55, bz, 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipDeviceSynchronize());
    // the many-row GEMM's LDS opt-in is per device; without it every launch stays on k_fc_splitk
    if (azk_fc_gemm12_prepare() != 0) { (void)hipGetLastError(); c->gemm12_min_rows = 0x7fffffff; c->gemm12_env = 1; }
    if (c->gemm_parts && azk_fc_terms_prepare(c->gemm_parts) != 0) {
        (void)hipGetLastError();
        return fail(c, AZ_ERR_HIP, "az_load_head: the 16-bit-term GEMM's LDS opt-in failed on this device (az_set_gemm_mode)");
    }
    c->head_loaded = true;
    return AZ_OK;
}

}  // extern "C"

// The context's two channel-last copies (and the NCHW staging buffer) hold at least n elements.
static int ensure_feat_copies(az_ctx *c, size_t n)
{
    if (n <= c->feat_owned_elems) return AZ_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->twin) HIPCHK(c, hipStreamSynchronize(c->twin->stream));      // (a lane may be copying out of the old buffers)
    if (c->feat_owned[0]) { for (float *f : c->feat_owned) hipFree(f); hipFree(c->feat_stage); }
    for (float *&f : c->feat_owned) f = nullptr;
    c->feat_stage = nullptr; c->feat_owned_elems = 0;
    ++c->feat_gen;
    for (float *&f : c->feat_owned) HIPCHK(c, hipMalloc((void **)&f, n * 4));
    HIPCHK(c, hipMalloc((void **)&c->feat_stage, n * 4));
    c->feat_owned_elems = n;
    return AZ_OK;
}

int set_feature_map_common(az_ctx *c, const float *src, bool src_is_host, int C, int H, int W, bool wait)
{
    int rc = check_ready(c, false, false);
    if (rc) return rc;
    if (!src || C != c->d.C || H <= 0 || W <= 0)
        return fail(c, AZ_ERR_INVALID, "feature map: channel count must match the loaded head");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)C * H * W;
    if ((rc = ensure_feat_copies(c, n)) != AZ_OK) return rc;
    // a second lane may still have to take its private copy of the map this context held two hand-overs ago (the copy
    // waits behind that lane's earlier search): the buffer is not written before it has
    if (c->twin && c->twin->ev_copy_live) HIPCHK(c, hipStreamWaitEvent(c->stream, c->twin->ev_copy, 0));
    const float *nchw = src;
    if (src_is_host) {
        HIPCHK(c, hipMemcpyAsync(c->feat_stage, src, n * 4, hipMemcpyHostToDevice, c->stream));
        nchw = c->feat_stage;
    }
    // RoIPool reads the map channel-last: one transpose per image, outside the level loop.
    c->feat_turn = (c->feat_turn + 1) % az_ctx::NFEAT;
    azk_nchw_to_nhwc(c->stream, nchw, c->feat_owned[c->feat_turn], C, H * W);
    if (wait) HIPCHK(c, hipStreamSynchronize(c->stream));      // the caller may now reuse / free `src`
    c->feat = c->feat_owned[c->feat_turn];
    c->d.H = H; c->d.W = W;
    return AZ_OK;
}

// The head buffers of a lane: pool5, split-K slabs, h6, h7 (a second lane has them from the start; a batch slot gets them
// when one of its searches has to be run again on its own).
int ensure_lane_head(az_ctx *t)
{
    if (!t || t->head_bufs) return AZ_OK;
    int rc;
    HIPCHK(t, hipSetDevice(t->device));
    const size_t R = (size_t)t->maxR;
    const AzHeadDims &d = t->d;
    if (t->batch_set && t->owner) {
        // a batch slot: the owner's spare set (az_ctx.h: SpareHead), created at its first use
        az_ctx *o = t->owner;
        auto &sp = o->spare;
        if (!sp.ready) {
            auto bad = [&](int code) { t->err = o->err; return code; };
#define A(p, n) if ((rc = dalloc(o, &sp.p, (n))) != AZ_OK) return bad(rc)
            A(pool5, R * d.K6);
            {
                const size_t p6 = (size_t)t->S6 * R * d.n6, p7 = (size_t)t->S7 * R * d.n7;
                A(part, p6 > p7 ? p6 : p7);
            }
            A(h6, R * d.n6); A(h7, R * d.n7);
            A(part7, (size_t)t->S7 * R * d.n7);
            if (t->gemm_parts) {
                A(pool5p, (size_t)t->gemm_parts * azk_act_plane_elems((int)R, d.K6)); A(gscale, 4);
                if (hipMemsetAsync(sp.pool5p, 0, (size_t)t->gemm_parts * azk_act_plane_elems((int)R, d.K6) * 2, o->stream) != hipSuccess ||
                    hipMemsetAsync(sp.gscale, 0, 4 * sizeof(float), o->stream) != hipSuccess || hipStreamSynchronize(o->stream) != hipSuccess)
                    return fail(t, AZ_ERR_HIP, "batch slot: clearing the spare operand planes");
            }
#undef A
            if (!sp.ev && hipEventCreateWithFlags(&sp.ev, hipEventDisableTiming) != hipSuccess)
                return fail(t, AZ_ERR_HIP, "batch slot: the spare head set's event");
            sp.ready = true; sp.ev_live = false;
        }
        t->pool5 = sp.pool5; t->part = sp.part; t->h6 = sp.h6; t->h7 = sp.h7; t->part7 = sp.part7;
        t->pool5p = sp.pool5p; t->gscale = sp.gscale;
        t->head_shared = true;
        t->head_bufs = true;
        return AZ_OK;
    }
#define A(p, n) if ((rc = dalloc(t, &t->p, (n))) != AZ_OK) return rc
    A(pool5, R * d.K6);
    {
        const size_t p6 = (size_t)t->S6 * R * d.n6, p7 = (size_t)t->S7 * R * d.n7;
        A(part, p6 > p7 ? p6 : p7);
    }
    A(h6, R * d.n6); A(h7, R * d.n7);
    A(part7, (size_t)t->S7 * R * d.n7);
    if (t->gemm_parts) {
        A(pool5p, (size_t)t->gemm_parts * azk_act_plane_elems((int)R, d.K6)); A(gscale, 4);
        if (hipMemsetAsync(t->pool5p, 0, (size_t)t->gemm_parts * azk_act_plane_elems((int)R, d.K6) * 2, t->stream) != hipSuccess ||
            hipMemsetAsync(t->gscale, 0, 4 * sizeof(float), t->stream) != hipSuccess || hipStreamSynchronize(t->stream) != hipSuccess)
            return fail(t, AZ_ERR_HIP, "lane: clearing the operand planes");
    }
#undef A
    t->head_bufs = true;
    return AZ_OK;
}

extern "C" {

int az_set_feature_map_dev(az_ctx *c, const float *dev_ptr, int C, int H, int W)
{
    return set_feature_map_common(c, dev_ptr, false, C, H, W);
}

int az_set_feature_map_host(az_ctx *c, const float *host_ptr, int C, int H, int W)
{
    return set_feature_map_common(c, host_ptr, true, C, H, W);
}

int az_set_feature_map_dev_async(az_ctx *c, const float *dev_ptr, int C, int H, int W)
{
    return set_feature_map_common(c, dev_ptr, false, C, H, W, false);
}

int az_set_feature_map_dev_nhwc(az_ctx *c, const float *dev_ptr, int C, int H, int W)
{
    int rc = check_ready(c, false, false);
    if (rc) return rc;
    if (!dev_ptr || C != c->d.C || H <= 0 || W <= 0)
        return fail(c, AZ_ERR_INVALID, "feature map: channel count must match the loaded head");
    c->feat = dev_ptr;                 // already in the layout RoIPool reads: borrowed, no copy
    c->d.H = H; c->d.W = W;
    return AZ_OK;
}

// ---- lanes ----------------------------------------------------------------------------------------------------------
static void destroy_twin(az_ctx *c)
{
    if (!c || !c->twin) return;
    az_ctx *t = c->twin;
    c->twin = nullptr;
    // (the twin's allocation lists hold only its own buffers: the head's weights belong to the owner)
    az_destroy(t);
    c->lane_order.clear();
    c->lane_next = 0;
}

// The image slots and pass buffers of the batches this lane has run (az_batch_launch).
static void destroy_batch(az_ctx *c)
{
    if (!c) return;
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto &B : c->bsets) {
        for (az_ctx *t : B.slots) {
            if (t->stream) hipStreamSynchronize(t->stream);
            if (t->h_res_own0) { t->h_res[0] = t->h_res_own0; t->h_res_own0 = nullptr; }     // (its own block again: az_destroy frees that)
            az_destroy(t);
        }
        B.slots.clear();
        for (void *q : {(void *)B.off, (void *)B.rois_cat, (void *)B.ubox_cat, (void *)B.feats, (void *)B.feat_hw, (void *)B.row_hw, (void *)B.args_dev, (void *)B.res_dev}) if (q) hipFree(q);
        if (B.args_host) hipHostFree(B.args_host);
        if (B.res_host) hipHostFree(B.res_host);
        B = az_ctx::Batch();
    }
    c->bset_turn = 0;
    c->batch_order.clear();
    c->batch_next = 0;
}

// A lane of context c: an az_ctx with its own stream, staging and per-search buffers that READS c's weights.
static int make_lane(az_ctx *c, az_ctx **out, bool with_head, const char *what)
{
    *out = nullptr;
    if (!c->head_loaded) return fail(c, AZ_ERR_STATE, std::string(what) + " needs a loaded head");
    az_ctx *t = nullptr;
    int rc = az_create(c->device, &t);
    if (rc) return fail(c, rc, std::string(what) + ": could not create the lane");
    t->owner = c;
    t->maxR = c->maxR; t->maxCand = c->maxCand; t->maxCh = c->maxCh;
    auto bail = [&](int code, const char *msg) { c->err = t->err.empty() ? std::string(what) + ": " + msg : t->err; az_destroy(t); return code; };
    if ((rc = ensure_geom(t)) != AZ_OK) return bail(rc, "geometry buffers");
    t->d = c->d; t->d.H = t->d.W = 0;
    t->S6 = c->S6; t->S7 = c->S7; t->gemm_parts = c->gemm_parts; t->w6_scale = c->w6_scale; t->spatial_scale = c->spatial_scale;
    t->W6 = c->W6; t->b6 = c->b6; t->W7 = c->W7; t->b7 = c->b7; t->Wt = c->Wt; t->bt = c->bt; t->W6p = c->W6p;
    t->head_bufs = false;
    if (with_head && (rc = ensure_lane_head(t)) != AZ_OK) return bail(rc, "head buffers");
    t->gemm12_env = c->gemm12_env; t->gemm12_min_rows = c->gemm12_min_rows; t->gemm12_dual_rows = c->gemm12_dual_rows;
    t->head_loaded = true;
    t->profiling = c->profiling; t->use_graphs = c->use_graphs; t->cal = c->cal;
    *out = t;
    return AZ_OK;
}

// The second lane.
static int ensure_twin(az_ctx *c)
{
    if (c->twin) return AZ_OK;
    az_ctx *t = nullptr;
    const int rc = make_lane(c, &t, true, "az_set_lanes");
    if (rc) return rc;
    c->twin = t;
    return AZ_OK;
}

int az_set_lanes(az_ctx *c, int lanes)
{
    if (!c || c->owner || (lanes != 1 && lanes != 2)) return fail(c, AZ_ERR_INVALID, "az_set_lanes: 1 or 2");
    if (!c->lane_order.empty() || !c->pend.empty() || !c->batch_order.empty()) return fail(c, AZ_ERR_STATE, "az_set_lanes: searches are still queued");
    c->batch_next = 0;
    c->lanes = lanes;
    c->lane_next = 0;
    return AZ_OK;
}

// The context (lane) the next search launched through the public entry points runs on.
static az_ctx *next_lane(az_ctx *c, const az_params *p, int *lane_out, int *rc_out)
{
    *lane_out = 0; *rc_out = AZ_OK;
    // (only searches that can be queued take turns: fixed proposal count, not the tuner's variant)
    if (!c || c->owner || c->lanes != 2 || !c->head_loaded || (p && (!p->fixed_num || (p->reserved & 4)))) return c;
    if (c->lane_next == 0) return c;
    if ((*rc_out = ensure_twin(c)) != AZ_OK) return nullptr;
    *lane_out = 1;
    return c->twin;
}

void *az_last_stream(az_ctx *c)
{
    if (!c) return nullptr;
    az_ctx *t = (!c->owner && !c->lane_order.empty() && c->lane_order.back() == 1 && c->twin) ? c->twin : c;
    return (void *)(t->last_s ? t->last_s : t->stream);       // (a two-stage search ends on the lane's second stream)
}

void *az_next_stream(az_ctx *c)
{
    if (!c) return nullptr;
    int lane, rc;
    az_ctx *t = next_lane(c, nullptr, &lane, &rc);
    return (void *)(t ? t->stream : c->stream);
}

static int launch_routed(az_ctx *c, const az_params *p, const float *dev_map, int C, int H, int W, int channels_last)
{
    if (!c) return AZ_ERR_INVALID;
    int lane, rc;
    az_ctx *t = next_lane(c, p, &lane, &rc);
    if (!t) return rc;
    if (dev_map) {
        // the map is handed to the lane that runs the search: an NCHW map is transposed on THAT lane's stream into that
        // lane's copies, a channel-last one is borrowed
        rc = channels_last ? az_set_feature_map_dev_nhwc(t, dev_map, C, H, W) : set_feature_map_common(t, dev_map, false, C, H, W, false);
        if (rc) { if (t != c) c->err = t->err; return rc; }
    } else if (t != c) {
        // the map was set on the context itself: the lane takes it behind whatever the context's stream still has to do to it
        // (an un-awaited transpose).  A map the caller holds in the channel-last layout is read where it lies (the caller
        // keeps it untouched until the fetch); one of the CONTEXT's own channel-last copies is copied into the lane's own
        // buffers -- the context writes its two copies in turn, and this lane's search may be fetched (or even start) only
        // after the context has been handed the map after next.
        if (!c->feat) return fail(c, AZ_ERR_STATE, "no feature map set");
        t->d.H = c->d.H; t->d.W = c->d.W;
        if (!t->ev_hand) HIPCHK(c, hipEventCreateWithFlags(&t->ev_hand, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(t->ev_hand, c->stream));
        HIPCHK(c, hipStreamWaitEvent(t->stream, t->ev_hand, 0));
        if (c->feat == c->feat_owned[0] || c->feat == c->feat_owned[1] || c->feat == c->feat_owned[2]) {
            const size_t n = (size_t)c->d.C * c->d.H * c->d.W;
            if ((rc = ensure_feat_copies(t, n)) != AZ_OK) { c->err = t->err; return rc; }
            t->feat_turn = (t->feat_turn + 1) % az_ctx::NFEAT;
            HIPCHK(c, hipMemcpyAsync(t->feat_owned[t->feat_turn], c->feat, n * 4, hipMemcpyDeviceToDevice, t->stream));
            if (!t->ev_copy) HIPCHK(c, hipEventCreateWithFlags(&t->ev_copy, hipEventDisableTiming));
            HIPCHK(c, hipEventRecord(t->ev_copy, t->stream));
            t->ev_copy_live = true;
            t->feat = t->feat_owned[t->feat_turn];
        } else
            t->feat = c->feat;
    }
    if (t != c) {
        if (t->cal.state == 0 && c->cal.state != 0) t->cal = c->cal;
        t->profiling = c->profiling; t->use_graphs = c->use_graphs;
    }
    rc = launch_impl(t, p);
    if (rc) { if (t != c) c->err = t->err; return rc; }
    if (t == c && c->cal.state == 1 && c->twin && c->twin->cal.state == 0) c->twin->cal = c->cal;
    c->lane_order.push_back(lane);
    if (c->lanes == 2 && p->fixed_num && !(p->reserved & 4)) c->lane_next ^= 1;
    return AZ_OK;
}

int az_propose_launch(az_ctx *c, const az_params *p)
{
    if (c && c->owner) return launch_impl(c, p);
    return launch_routed(c, p, nullptr, 0, 0, 0, 0);
}

int az_propose_launch_on(az_ctx *c, const az_params *p, const float *dev_map, int C, int H, int W, int channels_last)
{
    if (!dev_map) return fail(c, AZ_ERR_INVALID, "az_propose_launch_on: null map");
    return launch_routed(c, p, dev_map, C, H, W, channels_last);
}

int az_propose_fetch(az_ctx *c, double *boxes_out, float *scores_out, int cap, int *n_out, az_stats *st)
{
    if (!c || c->lane_order.empty()) return fail(c, AZ_ERR_STATE, "az_propose_fetch without az_propose_launch");
    if (!boxes_out || !n_out || cap < 0) return fail(c, AZ_ERR_INVALID, "az_propose_fetch: bad arguments");
    const int lane = c->lane_order.front();
    c->lane_order.pop_front();
    az_ctx *t = lane ? c->twin : c;
    if (!t || t->pend.empty()) return fail(c, AZ_ERR_STATE, "az_propose_fetch: the lane's queue is empty");
    c->last_fetch_lane = lane;
    const int rc = fetch_entry(t, 0, boxes_out, scores_out, cap, n_out, st);
    if (rc && t != c) c->err = t->err;
    return rc;
}

// ---- a batch of images in lockstep (az_search.hip: batch_launch_impl) --------------------------------------------------
static az_ctx *batch_lane(az_ctx *c, int *lane_out, int *rc_out)
{
    *lane_out = 0; *rc_out = AZ_OK;
    if (c->lanes != 2 || c->batch_next == 0) return c;
    if ((*rc_out = ensure_twin(c)) != AZ_OK) return nullptr;
    *lane_out = 1;
    return c->twin;
}

void *az_batch_next_stream(az_ctx *c)
{
    if (!c || c->owner) return c ? (void *)c->stream : nullptr;
    int lane, rc;
    az_ctx *L = batch_lane(c, &lane, &rc);
    return (void *)(L ? L->stream : c->stream);
}

int az_batch_launch(az_ctx *c, int n, const az_params *p, const float *const *maps, int C, int H, int W)
{
    if (!p || n < 1 || n > AZ_BATCH_MAX) return fail(c, AZ_ERR_INVALID, "az_batch_launch: 1 .. AZ_BATCH_MAX maps of the loaded head's channel count");
    az_params pa[AZ_BATCH_MAX];
    int Hs[AZ_BATCH_MAX], Ws[AZ_BATCH_MAX];
    for (int b = 0; b < n; ++b) { pa[b] = *p; Hs[b] = H; Ws[b] = W; }
    return az_batch_launch_shapes(c, n, pa, maps, C, Hs, Ws);
}

int az_batch_launch_shapes(az_ctx *c, int n, const az_params *pa, const float *const *maps, int C, const int *Hs, const int *Ws)
{
    if (!c || c->owner) return fail(c, AZ_ERR_INVALID, "az_batch_launch: the context itself, not a lane");
    if (!c->head_loaded) return fail(c, AZ_ERR_STATE, "az_load_head has not been called");
    if (!pa || !maps || !Hs || !Ws || n < 1 || n > AZ_BATCH_MAX || C != c->d.C)
        return fail(c, AZ_ERR_INVALID, "az_batch_launch: 1 .. AZ_BATCH_MAX maps of the loaded head's channel count");
    for (int b = 0; b < n; ++b)
        if (Hs[b] <= 0 || Ws[b] <= 0) return fail(c, AZ_ERR_INVALID, "az_batch_launch: 1 .. AZ_BATCH_MAX maps of the loaded head's channel count");
    int lane, rc;
    az_ctx *L = batch_lane(c, &lane, &rc);
    if (!L) return rc;
    // (the batch's pre-pass and passes write the lane's counters and per-roi outputs: a search of the lane whose result block
    //  is not on its way yet -- variable proposal count, the tuner's variant -- would lose them; launch_impl's rule for queueing)
    if (!L->pend.empty() && !L->pend.back().copied)
        return fail(c, AZ_ERR_STATE, "az_batch_launch: a search without a fixed proposal count is still unfetched on this lane");
    const int set = L->bset_turn;
    auto &B = L->bsets[set];
    if (B.n_live) return fail(c, AZ_ERR_STATE, "az_batch_launch: two batches per lane are already in flight, fetch one first");
    HIPCHK(c, hipSetDevice(c->device));
    while ((int)B.slots.size() < n) {
        az_ctx *t = nullptr;
        if ((rc = make_lane(c, &t, false, "az_batch_launch")) != AZ_OK) return rc;
        t->batch_set = &B;
        B.slots.push_back(t);
    }
    for (int b = 0; b < n; ++b) {
        az_ctx *t = B.slots[b];
        if (t->cal.state == 0 && c->cal.state != 0) t->cal = c->cal;
        t->profiling = 0; t->use_graphs = 0;
    }
    if (L != c) { if (L->cal.state == 0 && c->cal.state != 0) L->cal = c->cal; }
    int not_taken = 0;
    rc = batch_launch_impl(L, B, n, B.slots.data(), pa, maps, Hs, Ws, &not_taken);
    if (rc && !not_taken) {
        // (a HIP call failed somewhere in the launch sequence: whatever was enqueued is waited for and dropped, so that the
        //  slots do not sit on half a batch)
        const std::string msg = L->err;
        (void)hipStreamSynchronize(L->stream);
        (void)hipGetLastError();
        for (int b = 0; b < n; ++b) { B.slots[b]->pend.clear(); for (bool &sb : B.slots[b]->slot_busy) sb = false; }
        c->err = msg;
        return rc;
    }
    B.lockstep = !not_taken;
    if (not_taken) {
        // this shape / these settings: one search per image, each on its slot's own stream, behind whatever the caller made
        // the batch's stream wait for
        if (!L->ev_hand) HIPCHK(c, hipEventCreateWithFlags(&L->ev_hand, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(L->ev_hand, L->stream));
        for (int b = 0; b < n; ++b) {
            az_ctx *t = B.slots[b];
            if (!maps[b]) return fail(c, AZ_ERR_INVALID, "az_batch_launch: null map");
            HIPCHK(c, hipStreamWaitEvent(t->stream, L->ev_hand, 0));
            t->feat = maps[b]; t->d.H = Hs[b]; t->d.W = Ws[b];
            if ((rc = launch_impl(t, &pa[b])) != AZ_OK) {
                c->err = t->err;
                // (the images launched so far are dropped: nothing of this batch can be fetched)
                for (int q = 0; q < b; ++q) {
                    az_ctx *u = B.slots[q];
                    (void)hipStreamSynchronize(u->stream);
                    u->pend.clear();
                    for (bool &sb : u->slot_busy) sb = false;
                }
                return rc;
            }
        }
    }
    B.n_live = n; B.next_fetch = 0;
    for (int &r : B.rows_acc) r = 0;
    c->batch_order.push_back(lane | (set << 1));
    L->bset_turn ^= 1;
    if (c->lanes == 2) c->batch_next ^= 1;
    return AZ_OK;
}

int az_batch_fetch(az_ctx *c, int i, double *boxes_out, float *scores_out, int cap, int *n_out, az_stats *st)
{
    if (!c || c->owner || c->batch_order.empty()) return fail(c, AZ_ERR_STATE, "az_batch_fetch without az_batch_launch");
    if (!boxes_out || !n_out || cap < 0) return fail(c, AZ_ERR_INVALID, "az_batch_fetch: bad arguments");
    const int lane = c->batch_order.front() & 1, set = c->batch_order.front() >> 1;
    az_ctx *L = lane ? c->twin : c;
    if (!L) return fail(c, AZ_ERR_STATE, "az_batch_fetch: the lane is gone");
    auto &B = L->bsets[set];
    if (i != B.next_fetch || i >= B.n_live) return fail(c, AZ_ERR_INVALID, "az_batch_fetch: the images of a batch are fetched in order, each once");
    az_ctx *t = B.slots[i];
    int rc = t->pend.empty() ? fail(t, AZ_ERR_STATE, "az_batch_fetch: the image's search is not queued") :
                               fetch_entry(t, 0, boxes_out, scores_out, cap, n_out, st);
    if (rc) c->err = t->err;
    if (++B.next_fetch == B.n_live) {
        if (B.lockstep)       // (what the lane's next batches go by: both sets)
            for (auto &S : L->bsets) { for (int l = 0; l < AZ_MAX_LEVELS; ++l) S.rows_hint[l] = B.rows_acc[l]; S.hint_n = B.n_live; }
        B.n_live = 0; B.next_fetch = 0;
        c->batch_order.pop_front();
    }
    return rc;
}

// Every image of the oldest unfetched batch in one call: image i's boxes at boxes_out + i * cap * 4, its scores at
// scores_out + i * cap (may be NULL), its count in n_out[i], its statistics in st[i] (may be NULL).  Returns the first error
// (the remaining images are still collected, their n_out = -1 on error).
int az_batch_fetch_all(az_ctx *c, double *boxes_out, float *scores_out, int cap, int *n_out, az_stats *st)
{
    if (!c || c->owner || c->batch_order.empty()) return fail(c, AZ_ERR_STATE, "az_batch_fetch_all without az_batch_launch");
    if (!boxes_out || !n_out || cap < 0) return fail(c, AZ_ERR_INVALID, "az_batch_fetch_all: bad arguments");
    const int lane = c->batch_order.front() & 1, set = c->batch_order.front() >> 1;
    az_ctx *L = lane ? c->twin : c;
    if (!L) return fail(c, AZ_ERR_STATE, "az_batch_fetch_all: the lane is gone");
    const int n = L->bsets[set].n_live, i0 = L->bsets[set].next_fetch;
    int first = AZ_OK;
    std::string msg;
    for (int i = 0; i < i0; ++i) n_out[i] = 0;          // (already fetched one by one)
    for (int i = i0; i < n; ++i) {
        const int rc = az_batch_fetch(c, i, boxes_out + (size_t)i * cap * 4, scores_out ? scores_out + (size_t)i * cap : nullptr, cap,
                                      &n_out[i], st ? &st[i] : nullptr);
        if (rc) { n_out[i] = -1; if (!first) { first = rc; msg = c->err; } }
    }
    if (first) c->err = msg;
    return first;
}

int az_propose(az_ctx *c, const az_params *p, double *boxes_out, float *scores_out, int cap, int *n_out,
               az_stats *st)
{
    // (launch + fetch of the SAME search, on the context's own lane: with another search still queued the fetch would
    //  return that one's result)
    if (c && (!c->lane_order.empty() || !c->pend.empty()))
        return fail(c, AZ_ERR_STATE, "az_propose: a search launched with az_propose_launch is still queued, fetch it first");
    if (!boxes_out || !n_out || cap < 0) return fail(c, AZ_ERR_INVALID, "az_propose: bad arguments");
    int rc = launch_impl(c, p);
    if (rc) return rc;
    if (c) c->last_fetch_lane = 0;
    return fetch_entry(c, 0, boxes_out, scores_out, cap, n_out, st);
}

int az_measure_box(az_ctx *c, double *mfma_f32_tflops, double *copy_tb_per_s)
{
    if (!c) return AZ_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int rc = azk_measure_box(c->stream, mfma_f32_tflops, copy_tb_per_s, (size_t)1 << 30);
    if (rc) { (void)hipGetLastError(); return fail(c, AZ_ERR_HIP, std::string("az_measure_box: ") + hipGetErrorString((hipError_t)rc)); }
    return AZ_OK;
}

int az_set_pass_costs(az_ctx *c, int n, const int32_t *rows, const double *us)
{
    if (!c || n < 0 || n == 1 || n > 6 || (n && (!rows || !us))) return fail(c, AZ_ERR_INVALID, "az_set_pass_costs: 0 or 2..6 points");
    for (int i = 0; i < n; ++i)
        if (rows[i] <= 0 || !(us[i] > 0) || (i && (rows[i] <= rows[i - 1] || us[i] < us[i - 1])))
            return fail(c, AZ_ERR_INVALID, "az_set_pass_costs: rows and costs must ascend");
    c->cal = az_ctx::PassCal();
    for (int i = 0; i < n; ++i) { c->cal.rows[i] = rows[i]; c->cal.us[i] = us[i]; }
    c->cal.n = n;
    c->cal.state = n ? 1 : 0;
    if (c->twin) c->twin->cal = c->cal;
    return AZ_OK;
}

int az_get_pass_costs(az_ctx *c, int32_t *rows_out, double *us_out, int cap, int *n_out)
{
    if (!c || !n_out || cap < 0) return fail(c, AZ_ERR_INVALID, "az_get_pass_costs: bad arguments");
    const int n = c->cal.state == 1 ? c->cal.n : 0;
    *n_out = n;
    if (n > cap) return fail(c, AZ_ERR_CAPACITY, "az_get_pass_costs: cap too small");
    for (int i = 0; i < n; ++i) { if (rows_out) rows_out[i] = c->cal.rows[i]; if (us_out) us_out[i] = c->cal.us[i]; }
    return AZ_OK;
}

// ---- the exchange step of an image-sharded run, natively (az_rccl.hip) ------------------------------------------------
int az_rccl_unique_id(void *id_out, size_t cap)
{
    if (!id_out || cap < 128) return AZ_ERR_INVALID;
    std::string why;
    return azk_rccl_unique_id(id_out, &why) ? AZ_ERR_HIP : AZ_OK;
}

int az_rccl_init(az_ctx *c, const void *id, size_t id_bytes, int nranks, int rank)
{
    if (!c || c->owner || !id || id_bytes < 128 || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(c, AZ_ERR_INVALID, "az_rccl_init: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->comm_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    for (auto &e : c->comm_ev) if (!e) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (c->comm) { HIPCHK(c, hipStreamSynchronize(c->comm_stream)); azk_rccl_destroy(c->comm); c->comm = nullptr; }
    std::string why;
    if (azk_rccl_init(id, nranks, rank, &c->comm, &why)) return fail(c, AZ_ERR_HIP, "az_rccl_init: " + why);
    c->comm_ranks = nranks; c->comm_rank = rank;
    return AZ_OK;
}

int az_gather_records(az_ctx *c, const void *send_dev, void *recv_dev, size_t bytes_per_rank)
{
    if (!c || c->owner || !send_dev || !recv_dev || !bytes_per_rank) return fail(c, AZ_ERR_INVALID, "az_gather_records: bad arguments");
    if (!c->comm) return fail(c, AZ_ERR_STATE, "az_gather_records: no communicator (az_rccl_init)");
    HIPCHK(c, hipSetDevice(c->device));
    // the records were staged on the lanes' streams: the collective's stream waits, on the device, for what both have
    // queued so far
    HIPCHK(c, hipEventRecord(c->comm_ev[0], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->comm_ev[0], 0));
    if (c->twin) {
        HIPCHK(c, hipEventRecord(c->comm_ev[1], c->twin->stream));
        HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->comm_ev[1], 0));
    }
    // (two-stage searches stage their records on the lanes' second streams)
    for (az_ctx *l : {c, c->twin})
        if (l && l->s2_live && l->ev_s2) HIPCHK(c, hipStreamWaitEvent(c->comm_stream, l->ev_s2, 0));
    for (az_ctx *l : {c, c->twin})
        if (l && l->s3_live && l->ev_s3) HIPCHK(c, hipStreamWaitEvent(c->comm_stream, l->ev_s3, 0));
    std::string why;
    if (azk_rccl_all_gather(c->comm, c->comm_stream, send_dev, recv_dev, bytes_per_rank, &why))
        return fail(c, AZ_ERR_HIP, "az_gather_records: " + why);
    return AZ_OK;
}

int az_rccl_destroy(az_ctx *c)
{
    if (!c) return AZ_ERR_INVALID;
    if (c->comm) { hipSetDevice(c->device); hipStreamSynchronize(c->comm_stream); azk_rccl_destroy(c->comm); c->comm = nullptr; }
    return AZ_OK;
}

void *az_comm_stream(az_ctx *c) { return c ? (void *)c->comm_stream : nullptr; }

int az_result_record_layout(int k, size_t *bytes, size_t *n_off, size_t *boxes_off, size_t *scores_off)
{
    if (k <= 0 || k > AZ_TOPK_MAX) return AZ_ERR_INVALID;
    if (bytes) *bytes = RES_HDR + (size_t)k * 36;
    if (n_off) *n_off = offsetof(AzCounts, nsel);
    if (boxes_off) *boxes_off = RES_HDR;
    if (scores_off) *scores_off = RES_HDR + (size_t)k * 32;
    return AZ_OK;
}

int az_propose_stage_result_dev(az_ctx *c, void *dst_dev, size_t cap_bytes)
{
    // (applies to the search launched last: call it right behind az_propose_launch)
    if (!c) return AZ_ERR_INVALID;
    az_ctx *t = (!c->owner && !c->lane_order.empty() && c->lane_order.back() == 1) ? c->twin : c;
    if (!t) return fail(c, AZ_ERR_STATE, "az_propose_stage_result_dev without az_propose_launch");
    const int rc = stage_impl(t, dst_dev, cap_bytes);
    if (rc && t != c) c->err = t->err;
    return rc;
}


// The records of the batch launched last, image i to dst_dev + i * pitch: ONE strided device-to-device copy behind the batch
// (its result blocks lie side by side); an image that az_batch_fetch has to search again alone is staged again there.
int az_batch_stage_results_dev(az_ctx *c, void *dst_dev, size_t pitch_bytes, size_t cap_bytes)
{
    if (!c || c->owner || c->batch_order.empty()) return fail(c, AZ_ERR_STATE, "az_batch_stage_results_dev without az_batch_launch");
    az_ctx *L = (c->batch_order.back() & 1) ? c->twin : c;
    if (!L || !L->bsets[c->batch_order.back() >> 1].n_live) return fail(c, AZ_ERR_STATE, "az_batch_stage_results_dev: no batch in flight");
    auto &B = L->bsets[c->batch_order.back() >> 1];
    const int n = B.n_live;
    if (B.next_fetch) return fail(c, AZ_ERR_STATE, "az_batch_stage_results_dev: call it right behind az_batch_launch");
    az_ctx *t0 = B.slots[0];
    if (t0->pend.empty()) return fail(c, AZ_ERR_STATE, "az_batch_stage_results_dev: nothing queued");
    const size_t bytes = RES_HDR + (size_t)t0->pend.back().p.num_proposals * 36;
    if (!dst_dev || pitch_bytes < bytes || cap_bytes < pitch_bytes * (size_t)(n - 1) + bytes)
        return fail(c, AZ_ERR_INVALID, "az_batch_stage_results_dev: destination too small");
    HIPCHK(c, hipSetDevice(c->device));
    if (!B.lockstep) {
        // (the images were launched one by one on their slots: one copy each, on the slot's stream)
        for (int i = 0; i < n; ++i) {
            const int rc = stage_impl(B.slots[i], (unsigned char *)dst_dev + (size_t)i * pitch_bytes, bytes);
            if (rc) { c->err = B.slots[i]->err; return rc; }
        }
        return AZ_OK;
    }
    const size_t src_pitch = (size_t)((unsigned char *)B.slots[n > 1 ? 1 : 0]->cnt - (unsigned char *)B.slots[0]->cnt);
    hipStream_t s = L->stream;
    if (n > 1) HIPCHK(c, hipMemcpy2DAsync(dst_dev, pitch_bytes, B.res_dev, src_pitch, bytes, (size_t)n, hipMemcpyDeviceToDevice, s));
    else HIPCHK(c, hipMemcpyAsync(dst_dev, B.res_dev, bytes, hipMemcpyDeviceToDevice, s));
    for (int i = 0; i < n; ++i) {
        az_ctx *t = B.slots[i];
        auto &q = t->pend.back();
        q.stage_dst = (unsigned char *)dst_dev + (size_t)i * pitch_bytes; q.stage_cap = bytes;
        // ("the record is staged when the fetch returns": the slot's event again, behind the staging copy)
        if (q.copied) HIPCHK(c, hipEventRecord(t->ev_res[q.slot], s));
    }
    return AZ_OK;
}

int az_last_candidates(az_ctx *c, double *boxes_out, float *scores_out, int cap, int *n_out)
{
    if (c && !c->owner && c->last_fetch_lane == 1 && c->twin) {
        // (the search fetched last ran on the second lane: its candidates are in that lane's buffers)
        const int r2 = az_last_candidates(c->twin, boxes_out, scores_out, cap, n_out);
        if (r2) c->err = c->twin->err;
        return r2;
    }
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!n_out) return AZ_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
    if (c->stream3) HIPCHK(c, hipStreamSynchronize(c->stream3));
    if (c->cand_n < 0)
        return fail(c, AZ_ERR_STATE, "az_last_candidates: no fetched search, or a later call reused the candidate buffers");
    const int n = c->cand_n;
    *n_out = n;
    if (n > cap) return fail(c, AZ_ERR_CAPACITY, "az_last_candidates: cap too small");
    if (boxes_out) HIPCHK(c, hipMemcpy(boxes_out, c->Yall, (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost));
    if (scores_out) HIPCHK(c, hipMemcpy(scores_out, c->Sall, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return AZ_OK;
}

int az_set_graphs(az_ctx *c, int on)
{
    if (!c) return AZ_ERR_INVALID;
    c->use_graphs = on ? 1 : 0;
    return AZ_OK;
}

int az_set_profiling(az_ctx *c, int on)
{
    if (!c) return AZ_ERR_INVALID;
    const int asked = on;
    if ((on & 8) && (on & 16) && c->span_ring) {
        // bit 4 with bit 3: forget the spans recorded so far WITHOUT touching the GPU -- the ring keeps handing out slots it
        // has not used yet (each is used once), so nothing has to be cleared: no stream synchronisation, no copy, no idle
        // gap in front of the region the caller is about to time
        std::vector<AzEventRec> keep;
        for (auto &e : c->events) if (e.slot < 0) keep.push_back(e);
        c->events.swap(keep);
        on &= ~16;
    } else if (on & 8) {
        on &= ~16;
        // every slot starts as (first-in = ~0, last-out = 0); a slot is used once between two calls of this function
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
        if (c->stream3) HIPCHK(c, hipStreamSynchronize(c->stream3));
        if (!c->span_ring) HIPCHK(c, hipMalloc((void **)&c->span_ring, (size_t)az_ctx::SPAN_SLOTS * 16));
        std::vector<unsigned long long> init((size_t)az_ctx::SPAN_SLOTS * 2);
        for (size_t i = 0; i < init.size(); i += 2) { init[i] = ~0ull; init[i + 1] = 0ull; }
        HIPCHK(c, hipMemcpy(c->span_ring, init.data(), init.size() * 8, hipMemcpyHostToDevice));
        // (spans recorded so far are gone with the slots)
        std::vector<AzEventRec> keep;
        for (auto &e : c->events) if (e.slot < 0) keep.push_back(e);
        c->events.swap(keep);
        c->span_next = 0;
    }
    c->profiling = on;
    if (!on) clear_events(c);
    if (c->twin) az_set_profiling(c->twin, asked);
    return AZ_OK;
}

int az_last_kernel_times(az_ctx *c, char *names_out, float *ms_out, int32_t *level_out, int cap, int *n_out)
{
    if (!c || !n_out) return AZ_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
    if (c->stream3) HIPCHK(c, hipStreamSynchronize(c->stream3));
    const int n = (int)c->events.size();
    *n_out = n;
    if (c->event_errors) {
        const int ne = c->event_errors;
        c->event_errors = 0;
        return fail(c, AZ_ERR_HIP, "az_last_kernel_times: " + std::to_string(ne) + " hipEvent call(s) failed while profiling");
    }
    std::vector<unsigned long long> spans;
    if (c->span_ring && c->span_next > 0) {
        spans.resize((size_t)c->span_next * 2);
        HIPCHK(c, hipMemcpy(spans.data(), c->span_ring, spans.size() * 8, hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < n && i < cap; ++i) {
        float ms = 0.f;
        const int sl = c->events[i].slot;
        if (sl >= 0) {
            // 100 MHz ticks; a launch that left at once (another kernel owned it) or has not run reads as 0
            const unsigned long long t0 = (size_t)sl * 2 + 1 < spans.size() ? spans[(size_t)sl * 2] : ~0ull;
            const unsigned long long t1 = (size_t)sl * 2 + 1 < spans.size() ? spans[(size_t)sl * 2 + 1] : 0ull;
            ms = (t0 != ~0ull && t1 > t0) ? (float)((double)(t1 - t0) * 1e-5) : 0.f;
        } else
        if (hipEventElapsedTime(&ms, c->events[i].a, c->events[i].b) != hipSuccess)
            return fail(c, AZ_ERR_HIP, "az_last_kernel_times: hipEventElapsedTime failed");
        if (ms_out) ms_out[i] = ms;
        if (level_out) level_out[i] = c->events[i].level;
        if (names_out) {
            std::memset(names_out + 32 * i, 0, 32);
            std::strncpy(names_out + 32 * i, c->events[i].name.c_str(), 31);
        }
    }
    if (c->twin) {
        // the second lane's launches behind this lane's
        const int n0 = n < cap ? n : cap;
        int n2 = 0;
        const int rc = az_last_kernel_times(c->twin, names_out ? names_out + 32 * (size_t)n0 : nullptr, ms_out ? ms_out + n0 : nullptr,
                                            level_out ? level_out + n0 : nullptr, cap - n0, &n2);
        if (rc) { c->err = c->twin->err; return rc; }
        *n_out = n + n2;
    }
    return AZ_OK;
}

}  // extern "C"
