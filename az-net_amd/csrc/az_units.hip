// az_units.hip -- the C ABI's unit entry points (one stage of the path at a time: host in, host out, same kernels), the
// Fast R-CNN head on the shared map, NMS, the zoom-threshold tuner, recall evaluation and the image front-end.
#include "az_ctx.h"

extern "C" {

// --------------------------------------------------------------------------------------
// Unit entry points: host in, host out, same kernels.
static int sift_common(az_ctx *c, int C, double min_side, double *out, int cap, int *n_out)
{
    hipStream_t s = c->stream;
    int *Nptr = &c->cnt->scratch[0], *Pn = &c->cnt->scratch[1], *err = &c->cnt->scratch[2];
    HIPCHK(c, hipMemsetAsync(c->cnt, 0, sizeof(AzCounts), s));
    int rc = set_count(c, Nptr, C);
    if (rc) return rc;
    azk_region_keys(s, c->child, Nptr, c->maxCh, min_side, c->ckey);
    azk_dedup_regions(s, c->ckey, Nptr, c->maxCh, c->maxR, c->first, c->child, c->B[1], Pn, err, nullptr, nullptr);
    HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->cnt, sizeof(AzCounts), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (c->h_cnt->scratch[2]) return fail(c, AZ_ERR_CAPACITY, "sift_dup: region capacity exceeded");
    const int n = c->h_cnt->scratch[1];
    *n_out = n;
    if (n > cap) return fail(c, AZ_ERR_CAPACITY, "sift_dup: output cap too small");
    if (n) HIPCHK(c, hipMemcpy(out, c->B[1], (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost));
    return AZ_OK;
}

int az_sift_dup(az_ctx *c, const double *regions, int C, double min_side, double *out, int cap, int *n_out)
{
    int rc = check_geom(c);
    if (rc) return rc;
    if (C < 0 || (C && !regions) || !n_out || !(min_side > 0)) return fail(c, AZ_ERR_INVALID, "az_sift_dup: bad arguments");
    if (C > c->maxCh) return fail(c, AZ_ERR_CAPACITY, "az_sift_dup: too many regions");
    HIPCHK(c, hipSetDevice(c->device));
    if (C) HIPCHK(c, hipMemcpyAsync(c->child, regions, (size_t)C * 4 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return sift_common(c, C, min_side, out, cap, n_out);
}

int az_divide_region(az_ctx *c, const double *regions, int P, double min_side, double *out, int cap, int *n_out)
{
    int rc = check_geom(c);
    if (rc) return rc;
    if (P < 0 || (P && !regions) || !n_out || !(min_side > 0)) return fail(c, AZ_ERR_INVALID, "az_divide_region: bad arguments");
    if (P > c->maxR) return fail(c, AZ_ERR_CAPACITY, "az_divide_region: too many regions");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemsetAsync(c->cnt, 0, sizeof(AzCounts), s));
    if (P) HIPCHK(c, hipMemcpyAsync(c->Z, regions, (size_t)P * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    if ((rc = set_count(c, &c->cnt->PZ[0], P)) != AZ_OK) return rc;
    azk_divide(s, &c->cnt->PZ[0], &c->cnt->CH[0], &c->cnt->err, c->maxR, c->maxCh, c->Z, min_side, c->choff, c->child,
               c->ckey, nullptr, nullptr, nullptr, 0, nullptr);
    azk_dedup_regions(s, c->ckey, &c->cnt->CH[0], c->maxCh, c->maxR, c->first, c->child, c->B[1], &c->cnt->P[1],
                      &c->cnt->err, nullptr, nullptr);
    HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->cnt, sizeof(AzCounts), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (c->h_cnt->err) return fail(c, AZ_ERR_CAPACITY, "az_divide_region: ctx capacity exceeded");
    const int n = c->h_cnt->P[1];
    *n_out = n;
    if (n > cap) return fail(c, AZ_ERR_CAPACITY, "az_divide_region: output cap too small");
    if (n) HIPCHK(c, hipMemcpy(out, c->B[1], (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost));
    return AZ_OK;
}

int az_roi_dedup(az_ctx *c, const double *boxes, int P, double scale, double dedup, int batch_size,
                 float *rois_out, int32_t *index_out, int32_t *inv_index_out, int *n_unique)
{
    int rc = check_geom(c);
    if (rc) return rc;
    if (P < 0 || (P && !boxes) || !n_unique || batch_size <= 0) return fail(c, AZ_ERR_INVALID, "az_roi_dedup: bad arguments");
    if (P > c->maxR) return fail(c, AZ_ERR_CAPACITY, "az_roi_dedup: too many regions");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemsetAsync(c->cnt, 0, sizeof(AzCounts), s));
    if (P) HIPCHK(c, hipMemcpyAsync(c->B[0], boxes, (size_t)P * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    if ((rc = set_count(c, &c->cnt->P[0], P)) != AZ_OK) return rc;
    azk_rois_dedup(s, c->B[0], &c->cnt->P[0], c->maxR, scale, (float)dedup, batch_size, c->rois, c->key, c->grp,
                   c->first, c->index, c->inv, c->urois, c->ubox, &c->cnt->U[0]);
    HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->cnt, sizeof(AzCounts), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const int U = c->h_cnt->U[0];
    *n_unique = U;
    if (P && rois_out) HIPCHK(c, hipMemcpy(rois_out, c->rois, (size_t)P * 5 * 4, hipMemcpyDeviceToHost));
    if (U && index_out) HIPCHK(c, hipMemcpy(index_out, c->index, (size_t)U * 4, hipMemcpyDeviceToHost));
    if (P && inv_index_out) HIPCHK(c, hipMemcpy(inv_index_out, c->inv, (size_t)P * 4, hipMemcpyDeviceToHost));
    return AZ_OK;
}

static int stage_rois(az_ctx *c, const float *rois, int R)
{
    if (R < 0 || (R && !rois)) return fail(c, AZ_ERR_INVALID, "bad rois");
    if (R > c->maxR) return fail(c, AZ_ERR_CAPACITY, "too many rois");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->cnt, 0, sizeof(AzCounts), c->stream));
    if (R) HIPCHK(c, hipMemcpyAsync(c->urois, rois, (size_t)R * 5 * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->ubox, 0, (size_t)(R > 0 ? R : 1) * 4 * sizeof(double), c->stream));
    return set_count(c, &c->cnt->U[0], R);
}

int az_roi_pool(az_ctx *c, const float *rois, int R, float *out)
{
    int rc = check_ready(c, true);
    if (rc) return rc;
    if ((rc = stage_rois(c, rois, R)) != AZ_OK) return rc;
    if (!out) return fail(c, AZ_ERR_INVALID, "az_roi_pool: null output");
    azk_roi_pool(c->stream, c->feat, c->d, c->spatial_scale, c->urois, &c->cnt->U[0], c->maxR, c->pool5, nullptr, 0, 0,
                 0);
    // the ABI returns Caffe's [R, C, 7, 7] flattening; HBM holds [R, 49, C]
    if (R) azk_permute_k(c->stream, c->pool5, c->part, R, c->d.C, 0);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (R) HIPCHK(c, hipMemcpy(out, c->part, (size_t)R * c->d.K6 * 4, hipMemcpyDeviceToHost));
    return AZ_OK;
}

int az_head_forward(az_ctx *c, const float *rois, int R, float *zoom_prob, float *adj_prob, float *adj_bbox)
{
    int rc = check_ready(c, true);
    if (rc) return rc;
    if ((rc = stage_rois(c, rois, R)) != AZ_OK) return rc;
    if (!(c->profiling & 4)) clear_events(c);
    prep_scale(c);
    // (the row count is known on the host here: many rows take the many-row GEMM, as a one-pass search does)
    launch_head(c, &c->cnt->U[0], 0, 1, 1, 0.0, c->zoom_u, c->score_u, c->delta_u, 0.0, false, 0, nullptr, nullptr, R);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    if (R && zoom_prob) HIPCHK(c, hipMemcpy(zoom_prob, c->zoom_u, (size_t)R * 4, hipMemcpyDeviceToHost));
    if (R && adj_prob) HIPCHK(c, hipMemcpy(adj_prob, c->score_u, (size_t)R * AZ_NSUB * 4, hipMemcpyDeviceToHost));
    if (R && adj_bbox) HIPCHK(c, hipMemcpy(adj_bbox, c->delta_u, (size_t)R * 4 * AZ_NSUB * 4, hipMemcpyDeviceToHost));
    return AZ_OK;
}

int az_decode_filter(az_ctx *c, const double *anchors, const float *deltas, const float *scores, int R,
                     int im_h, int im_w, double eps, double min_side, double *boxes_out, float *scores_out,
                     int cap, int *n_out)
{
    int rc = check_geom(c);
    if (rc) return rc;
    if (R < 0 || (R && (!anchors || !deltas || !scores)) || !n_out) return fail(c, AZ_ERR_INVALID, "az_decode_filter: bad arguments");
    if (R > c->maxR) return fail(c, AZ_ERR_CAPACITY, "az_decode_filter: too many regions");
    c->cand_n = -1;                              // Yall / Sall are reused below
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemsetAsync(c->cnt, 0, sizeof(AzCounts), s));
    // stage: anchors -> ubox, deltas -> delta_u, scores -> Sout (scratch); inv = identity
    std::vector<int> ident(R);
    for (int i = 0; i < R; ++i) ident[i] = i;
    if (R) {
        HIPCHK(c, hipMemcpyAsync(c->ubox, anchors, (size_t)R * 4 * sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->delta_u, deltas, (size_t)R * 4 * AZ_NSUB * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->Sout, scores, (size_t)R * AZ_NSUB * 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->inv, ident.data(), (size_t)R * 4, hipMemcpyHostToDevice, s));
    }
    if ((rc = set_count(c, &c->cnt->P[0], R)) != AZ_OK) return rc;
    HIPCHK(c, hipMemsetAsync(c->zoom_u, 0, (size_t)(R > 0 ? R : 1) * 4, s));
    azk_decode_unit(s, c->ubox, c->delta_u, c->Sout, R, im_h, im_w, eps, c->pred_u, c->score_u);
    azk_flags_compact(s, c->cnt, 0, c->maxR, c->maxCand, c->ubox, c->inv, c->pred_u, c->score_u, c->zoom_u, 2.0,
                      min_side, 0, c->cflag, c->zflag, c->bc_c, c->bc_z, c->Yall, c->Sall, c->Z, c->zr);
    HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->cnt, sizeof(AzCounts), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const int n = c->h_cnt->NC[0];
    *n_out = n;
    if (n > cap) return fail(c, AZ_ERR_CAPACITY, "az_decode_filter: output cap too small");
    if (n && boxes_out) HIPCHK(c, hipMemcpy(boxes_out, c->Yall, (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost));
    if (n && scores_out) HIPCHK(c, hipMemcpy(scores_out, c->Sall, (size_t)n * 4, hipMemcpyDeviceToHost));
    return AZ_OK;
}

int az_topk(az_ctx *c, const float *scores, int n, int k, int32_t *idx_out, int *n_out)
{
    int rc = check_geom(c);
    if (rc) return rc;
    if (n < 0 || (n && !scores) || k <= 0 || !idx_out || !n_out) return fail(c, AZ_ERR_INVALID, "az_topk: bad arguments");
    if (n > c->maxCand || k > AZ_TOPK_MAX) return fail(c, AZ_ERR_CAPACITY, "az_topk: n or k too large");
    c->cand_n = -1;                              // Sall is reused below
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemsetAsync(c->cnt, 0, sizeof(AzCounts), s));
    if (n) HIPCHK(c, hipMemcpyAsync(c->Sall, scores, (size_t)n * 4, hipMemcpyHostToDevice, s));
    if ((rc = set_count(c, &c->cnt->scratch[0], n)) != AZ_OK) return rc;
    azk_topk(s, c->Sall, &c->cnt->scratch[0], c->maxCand, k, c->sel_idx, &c->cnt->nsel, c->rank_part);
    HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->cnt, sizeof(AzCounts), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const int m = c->h_cnt->nsel;
    *n_out = m;
    if (m) HIPCHK(c, hipMemcpy(idx_out, c->sel_idx, (size_t)m * 4, hipMemcpyDeviceToHost));
    return AZ_OK;
}

int az_nms(az_ctx *c, const float *dets, int n, double thresh, int64_t *keep, int *n_keep)
{
    if (!c) return AZ_ERR_INVALID;
    if (n < 0 || (n && (!dets || !keep)) || !n_keep) return fail(c, AZ_ERR_INVALID, "az_nms: bad arguments");
    *n_keep = 0;
    if (n == 0) return AZ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (n <= azk_nms_small_max()) {
        // the reference's own call-site size (apply_nms, test.py:467-484: <= 100 boxes per class): ONE launch, no copy
        // commands -- the workgroup reads the boxes from and writes the keep list to host-mapped memory
        if (!c->h_nms) HIPCHK(c, hipHostMalloc((void **)&c->h_nms, 8192, hipHostMallocMapped));
        float *hd = (float *)c->h_nms;                                  // [256][5] f32 = 5120 B
        long long *hk = (long long *)(c->h_nms + 5120);                 // [256] i64 = 2048 B, then the count
        int *hn = (int *)(c->h_nms + 5120 + 2048);
        std::memcpy(hd, dets, (size_t)n * 5 * sizeof(float));
        const unsigned tag = nms_next_tag(c);
        *hn = 0;
        if (!(c->profiling & 4)) clear_events(c);
        { Timed t(c, "nms", n);
          azk_nms_one_small(s, hd, n, thresh, hk, hn, tag); }
        // Poll the result in the mapped block -- a stream synchronisation costs an interrupt round trip (~10-15 us) on top of
        // a kernel of about that length.  Count and keep entries carry the call's tag (words may land out of order); the
        // stream's own completion is picked up by whatever uses it next (same stream: ordered).  (AZ_NMS_POLL=0, profiling,
        // or no answer within a millisecond: the plain wait, after which everything is visible.)
        static const bool poll = !(getenv("AZ_NMS_POLL") && !atoi(getenv("AZ_NMS_POLL")));
        const unsigned want = tag & 0x3FFFFFu;
        bool got = false;
        if (poll && !c->profiling) {
            const volatile int *vn = hn;
            for (int spin = 0; spin < 200000 && !got; ++spin) got = ((unsigned)*vn >> 9) == want;
            if (got) got = nms_keep_tagged(hk, (int)((unsigned)*vn & 0x1FFu), tag, 200000);
        }
        if (!got) HIPCHK(c, hipStreamSynchronize(s));
        HIPCHK(c, hipGetLastError());
        const unsigned word = (unsigned)*(const volatile int *)hn;
        const int nk = (int)(word & 0x1FFu);
        if ((word >> 9) != want || nk > n || !nms_keep_tagged(hk, nk, tag, 0))
            return fail(c, AZ_ERR_HIP, "az_nms: the kernel left no result");
        *n_keep = nk;
        for (int i = 0; i < nk; ++i) keep[i] = (long long)(unsigned)(hk[i] & 0xFFFFFFFFll);
        return AZ_OK;
    }
    if (n > c->nms_cap) {
        HIPCHK(c, hipStreamSynchronize(s));
        if (c->nms_dets) { hipFree(c->nms_dets); hipFree(c->nms_sdets); hipFree(c->nms_order); hipFree(c->nms_mask); hipFree(c->nms_keep); hipFree(c->nms_rank); }
        c->nms_dets = nullptr; c->nms_cap = 0;
        int cap = 1024;
        while (cap < n) cap *= 2;
        const size_t W = (size_t)(cap + 63) / 64;
        if (azk_nms_scan_lds_bytes(cap) > 150000) return fail(c, AZ_ERR_CAPACITY, "az_nms: n too large");
        HIPCHK(c, hipMalloc((void **)&c->nms_dets, (size_t)cap * 5 * 4));
        HIPCHK(c, hipMalloc((void **)&c->nms_sdets, (size_t)cap * 5 * 4));
        HIPCHK(c, hipMalloc((void **)&c->nms_order, (size_t)cap * 4 + 16));
        HIPCHK(c, hipMalloc((void **)&c->nms_mask, ((size_t)cap * W + azk_nms_band_words(cap)) * 8));      // mask, then band
        HIPCHK(c, hipMalloc((void **)&c->nms_rank, (size_t)cap * 4));
        HIPCHK(c, hipMemset(c->nms_rank, 0, (size_t)cap * 4));
        HIPCHK(c, hipMalloc((void **)&c->nms_keep, (size_t)cap * 8 + 16));
        c->nms_cap = cap;
    }
    int *nk = c->nms_order + c->nms_cap;      // spare int after the order array
    HIPCHK(c, hipMemcpyAsync(c->nms_dets, dets, (size_t)n * 5 * 4, hipMemcpyHostToDevice, s));
    if (!(c->profiling & 4)) clear_events(c);
    static const bool poll_g = !(getenv("AZ_NMS_POLL") && !atoi(getenv("AZ_NMS_POLL")));
    if (poll_g && !c->profiling) {
        // keep list and count straight into host-mapped memory, the count last (k_nms_scan): no copy-back commands, no
        // stream synchronisation -- the host polls the count
        const size_t need = (size_t)n * 8 + 64;
        if (need > c->h_nmsg_cap) {
            HIPCHK(c, hipStreamSynchronize(s));
            if (c->h_nmsg) hipHostFree(c->h_nmsg);
            c->h_nmsg = nullptr; c->h_nmsg_cap = 0;
            HIPCHK(c, hipHostMalloc((void **)&c->h_nmsg, need * 2, hipHostMallocMapped));
            c->h_nmsg_cap = need * 2;
        }
        volatile long long *hn = (volatile long long *)c->h_nmsg;      // (tag << 32) | count
        long long *hk = (long long *)(c->h_nmsg + 64);                 // (tag << 32) | index
        const unsigned tag = nms_next_tag(c);
        *hn = 0;
        azk_nms(s, c->nms_dets, n, thresh, c->nms_order, c->nms_sdets, c->nms_mask, c->nms_mask + (size_t)c->nms_cap * ((c->nms_cap + 63) / 64), (unsigned long long *)c->nms_rank, hk, (int *)c->h_nmsg, tag);
        bool got = false;
        for (long spin = 0; spin < 4000000 && !got; ++spin) got = (unsigned)((unsigned long long)*hn >> 32) == tag;
        if (got) got = nms_keep_tagged(hk, (int)(*hn & 0xFFFFFFFFll), tag, 200000);
        if (!got) HIPCHK(c, hipStreamSynchronize(s));
        HIPCHK(c, hipGetLastError());
        const long long word = *hn;
        const int h_nk2 = (int)(word & 0xFFFFFFFFll);
        if ((unsigned)((unsigned long long)word >> 32) != tag || h_nk2 < 0 || h_nk2 > n || !nms_keep_tagged(hk, h_nk2, tag, 0))
            return fail(c, AZ_ERR_HIP, "az_nms: the kernels left no result");
        *n_keep = h_nk2;
        for (int i = 0; i < h_nk2; ++i) keep[i] = (long long)(unsigned)(hk[i] & 0xFFFFFFFFll);
        return AZ_OK;
    }
    { Timed t(c, "nms", n);
      azk_nms(s, c->nms_dets, n, thresh, c->nms_order, c->nms_sdets, c->nms_mask, c->nms_mask + (size_t)c->nms_cap * ((c->nms_cap + 63) / 64), (unsigned long long *)c->nms_rank, c->nms_keep, nk); }
    int h_nk = 0;
    HIPCHK(c, hipMemcpyAsync(&h_nk, nk, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, hipGetLastError());
    *n_keep = h_nk;
    if (h_nk) HIPCHK(c, hipMemcpy(keep, c->nms_keep, (size_t)h_nk * 8, hipMemcpyDeviceToHost));
    return AZ_OK;
}


// --------------------------------------------------------------------------------------
// Fast R-CNN head on the shared conv map (lib/detect/test.py:259-318,432-445).
int az_load_det_head(az_ctx *c, int C, int n6, int n7, int ncls, const float *W6, const float *b6,
                     const float *W7, const float *b7, const float *Wc, const float *bc, const float *Wb,
                     const float *bb)
{
    if (!c) return AZ_ERR_INVALID;
    if (!W6 || !b6 || !W7 || !b7 || !Wc || !bc || !Wb || !bb) return fail(c, AZ_ERR_INVALID, "az_load_det_head: null pointer");
    if (C <= 0 || (C & 3) || n6 <= 0 || (n6 & 3) || n7 <= 0 || (n7 & 3) || ncls < 2 || ncls > 256)
        return fail(c, AZ_ERR_INVALID, "az_load_det_head: C, n6, n7 multiples of 4; 2 <= ncls <= 256");
    if (c->head_loaded && C != c->d.C) return fail(c, AZ_ERR_INVALID, "az_load_det_head: C differs from the AZ head's");
    int rc = ensure_geom(c);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (void *p : c->allocs_det) hipFree(p);
    c->allocs_det.clear();
    c->det_loaded = false;
    const size_t R = (size_t)c->maxR, K6 = (size_t)C * 49, NO = (size_t)5 * ncls;
    c->det_n6 = n6; c->det_n7 = n7; c->det_ncls = ncls;
    c->det_S6 = azk_fc_split((int)K6); c->det_S7 = azk_fc_split(n6);
#define A(p, n) if ((rc = dalloc_det(c, &c->p, (n))) != AZ_OK) return rc
    A(dW6, azk_tiled_elems(n6, (int)K6)); A(db6, n6); A(dW7, azk_tiled_elems(n7, n6)); A(db7, n7);
    A(dWt, azk_tiled_elems((int)NO, n7)); A(dbt, NO);
    A(dh6, R * n6); A(dh7, R * n7);
    {
        size_t pm = (size_t)c->det_S6 * R * n6;
        const size_t p7 = (size_t)c->det_S7 * R * n7, pt = (size_t)AZK_TAIL_SPLIT * R * NO, pw = (size_t)n6 * K6;
        pm = pm > p7 ? pm : p7; pm = pm > pt ? pm : pt; pm = pm > pw ? pm : pw;
        A(dpart, pm);
    }
    A(dprob_u, R * ncls); A(ddelta_u, R * 4 * ncls); A(dpred_u, R * ncls * 4); A(dprob, R * ncls); A(dpred, R * ncls * 4);
    if (!c->pool5) { A(pool5, R * K6); }     // normally the AZ head's buffer is shared
    c->dW6p = nullptr; c->dgscale = nullptr;
    if (c->gemm_parts && c->pool5p && azk_fc_terms_prepare(c->gemm_parts) == 0) {
        A(dW6p, (size_t)c->gemm_parts * azk_weight_plane_elems(n6, (int)K6)); A(dgscale, 4);
        HIPCHK(c, hipMemsetAsync(c->dgscale, 0, 4 * sizeof(float), c->stream));
    }
#undef A
    if (!c->head_loaded) { c->d.C = C; c->d.pooled = 7; c->d.K6 = (int)K6; }
    {
        struct TmpGuard { float *p = nullptr; ~TmpGuard() { if (p) hipFree(p); } } tg;
        size_t te = (size_t)n6 * K6;
        if ((size_t)n7 * n6 > te) te = (size_t)n7 * n6;
        if (NO * n7 > te) te = NO * n7;
        HIPCHK(c, hipMalloc((void **)&tg.p, te * 4));
        float *tmp = tg.p;
        HIPCHK(c, hipMemcpy(c->dpart, W6, (size_t)n6 * K6 * 4, hipMemcpyHostToDevice));
        azk_permute_k(c->stream, c->dpart, tmp, n6, C, 1);          // bin-major columns, like the AZ head
        azk_tile_weights(c->stream, tmp, c->dW6, n6, (int)K6);
        if (c->dW6p) {
            c->det_w6_scale = 0.f;
            if (c->gemm_parts == 2) {
                float mx = 0.f;
                for (size_t i = 0, n = (size_t)n6 * K6; i < n; ++i) { const float a = fabsf(W6[i]); if (a > mx) mx = a; }
                c->det_w6_scale = 1.f;
                if (mx > 0.f && mx < INFINITY) { int e; (void)frexpf(mx, &e); c->det_w6_scale = ldexpf(1.f, 15 - e); }
            }
            azk_split_weight_planes(c->stream, tmp, c->dW6p, n6, (int)K6, c->gemm_parts, c->det_w6_scale);
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(tmp, W7, (size_t)n7 * n6 * 4, hipMemcpyHostToDevice));
        azk_tile_weights(c->stream, tmp, c->dW7, n7, n6);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        // rows 0..ncls-1 cls_score, ncls..5*ncls-1 bbox_pred
        HIPCHK(c, hipMemcpy(tmp, Wc, (size_t)ncls * n7 * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(tmp + (size_t)ncls * n7, Wb, (size_t)4 * ncls * n7 * 4, hipMemcpyHostToDevice));
        azk_tile_weights(c->stream, tmp, c->dWt, (int)NO, n7);
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    HIPCHK(c, hipMemcpy(c->db6, b6, (size_t)n6 * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->db7, b7, (size_t)n7 * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->dbt, bc, (size_t)ncls * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->dbt + ncls, bb, (size_t)4 * ncls * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipDeviceSynchronize());
    c->det_loaded = true;
    return AZ_OK;
}

// the detection head on the `U` rois in ctx->urois / ctx->ubox
// (rows_bound: what the host knows about the row count -- the number of boxes before the 1/16 dedup)
static void launch_det_head(az_ctx *c, const int *Uptr, int im_h, int im_w, double eps, int rows_bound)
{
    AzHeadDims d = c->d;
    const int K6 = d.C * 49, NO = 5 * c->det_ncls;
    d.K6 = K6;
    // many rows (the reference's 300 proposals per image): fc6 / fc7 on the many-row GEMM, as int6 (same bits either way)
    auto gemm = [&](const float *x, int ldx, const float *W, int N, int K, int S, float *part) {
        const bool can12 = (N / 128) * S >= 256 && N % 128 == 0 && K % 32 == 0 && azk_fc_chunk(K, S) * S == K &&
                           azk_fc_chunk(K, S) >= 64 && c->gemm12_min_rows < 0x7fffffff && rows_bound >= c->gemm12_min_rows;
        if (can12) azk_fc_gemm12(c->stream, x, ldx, W, K, Uptr, c->maxR, N, K, S, azk_fc_chunk(K, S), part);
        else azk_fc_gemm(c->stream, x, ldx, W, K, Uptr, c->maxR, N, K, S, part);
    };
    const bool terms = c->gemm_parts && c->dW6p;             // (16-bit-term modes: fc6, 86 % of this head's FLOPs, as int6)
    if (terms && c->gemm_parts == 2)
        azk_feat_scale(c->stream, c->feat, (long long)d.C * d.H * d.W, c->dgscale, c->det_w6_scale);
    { Timed t(c, "det_roi_pool", 0);
      azk_roi_pool(c->stream, c->feat, d, c->spatial_scale, c->urois, Uptr, c->maxR, c->pool5, terms ? c->pool5p : nullptr,
                   terms ? azk_act_plane_elems(c->maxR, K6) : 0, terms ? c->gemm_parts : 0, 0, 0,
                   (terms && c->gemm_parts == 2) ? c->dgscale : nullptr); }
    { Timed t(c, "det_fc6_gemm", 0, 1);
      if (terms)
          azk_fc_gemm_terms(c->stream, c->pool5p, K6, azk_act_plane_elems(c->maxR, K6), c->dW6p, K6,
                            azk_weight_plane_elems(c->det_n6, K6), Uptr, c->maxR, c->det_n6, K6, c->det_S6,
                            azk_fc_chunk(K6, c->det_S6), c->dpart, c->gemm_parts, c->dgscale);
      else
          gemm(c->pool5, K6, c->dW6, c->det_n6, K6, c->det_S6, c->dpart); }
    { Timed t(c, "det_fc6_reduce", 0);
      azk_fc_reduce(c->stream, c->dpart, c->db6, Uptr, c->maxR, c->det_n6, c->det_S6, c->dh6, c->det_n6, 1); }
    { Timed t(c, "det_fc7_gemm", 0, 1);
      gemm(c->dh6, c->det_n6, c->dW7, c->det_n7, c->det_n6, c->det_S7, c->dpart); }
    { Timed t(c, "det_fc7_reduce", 0);
      azk_fc_reduce(c->stream, c->dpart, c->db7, Uptr, c->maxR, c->det_n7, c->det_S7, c->dh7, c->det_n7, 1); }
    { Timed t(c, "det_tail_gemm", 0, 1);
      azk_fc_gemm(c->stream, c->dh7, c->det_n7, c->dWt, c->det_n7, Uptr, c->maxR, NO, c->det_n7, AZK_TAIL_SPLIT,
                  c->dpart); }
    { Timed t(c, "det_epilogue", 0);
      azk_det_epilogue(c->stream, c->dpart, AZK_TAIL_SPLIT, c->det_ncls, c->dbt, c->ubox, Uptr, c->maxR, im_h, im_w,
                       eps, c->dprob_u, c->ddelta_u, c->dpred_u); }
}

static int check_det(az_ctx *c)
{
    if (!c) return AZ_ERR_INVALID;
    if (!c->det_loaded) return fail(c, AZ_ERR_STATE, "az_load_det_head has not been called");
    if (!c->feat) return fail(c, AZ_ERR_STATE, "no feature map set");
    return AZ_OK;
}

int az_det_forward(az_ctx *c, const float *rois, int R, float *cls_prob, float *bbox_pred)
{
    int rc = check_det(c);
    if (rc) return rc;
    if ((rc = stage_rois(c, rois, R)) != AZ_OK) return rc;
    if (!(c->profiling & 4)) clear_events(c);
    launch_det_head(c, &c->cnt->U[0], 1, 1, 0.0, R);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    const size_t nc = (size_t)c->det_ncls;
    if (R && cls_prob) HIPCHK(c, hipMemcpy(cls_prob, c->dprob_u, (size_t)R * nc * 4, hipMemcpyDeviceToHost));
    if (R && bbox_pred) HIPCHK(c, hipMemcpy(bbox_pred, c->ddelta_u, (size_t)R * 4 * nc * 4, hipMemcpyDeviceToHost));
    return AZ_OK;
}

int az_detect(az_ctx *c, const double *boxes, int P, double scale, double dedup, int batch_size, int im_h,
              int im_w, double eps, float *scores_out, double *boxes_out)
{
    int rc = check_det(c);
    if (rc) return rc;
    if (P < 0 || (P && !boxes) || batch_size <= 0 || !(scale > 0)) return fail(c, AZ_ERR_INVALID, "az_detect: bad arguments");
    if (P > c->maxR) return fail(c, AZ_ERR_CAPACITY, "az_detect: too many boxes");
    if (P == 0) return AZ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (!(c->profiling & 4)) clear_events(c);
    HIPCHK(c, hipMemsetAsync(c->cnt, 0, sizeof(AzCounts), s));
    HIPCHK(c, hipMemcpyAsync(c->B[0], boxes, (size_t)P * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    if ((rc = set_count(c, &c->cnt->P[0], P)) != AZ_OK) return rc;
    azk_rois_dedup(s, c->B[0], &c->cnt->P[0], c->maxR, scale, (float)dedup, batch_size, c->rois, c->key, c->grp,
                   c->first, c->index, c->inv, c->urois, c->ubox, &c->cnt->U[0]);
    launch_det_head(c, &c->cnt->U[0], im_h, im_w, eps, P);
    azk_det_gather(s, &c->cnt->P[0], c->inv, c->det_ncls, c->dprob_u, c->dpred_u, c->dprob, c->dpred);
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, hipGetLastError());
    const size_t nc = (size_t)c->det_ncls;
    if (scores_out) HIPCHK(c, hipMemcpy(scores_out, c->dprob, (size_t)P * nc * 4, hipMemcpyDeviceToHost));
    if (boxes_out) HIPCHK(c, hipMemcpy(boxes_out, c->dpred, (size_t)P * nc * 4 * sizeof(double), hipMemcpyDeviceToHost));
    return AZ_OK;
}

// --------------------------------------------------------------------------------------
// apply_nms (lib/detect/test.py:467-484) calls nms once per class per image: n_groups independent
// problems, here in one call.  Groups of up to 256 boxes (all of them, at that call site) share ONE
// launch, a workgroup each; larger groups go through az_nms one by one.
int az_nms_batched(az_ctx *c, const float *dets, const int32_t *offsets, int n_groups, double thresh,
                   int64_t *keep, int32_t *n_keep)
{
    if (!c || n_groups < 0 || (n_groups && (!offsets || !n_keep)))
        return fail(c, AZ_ERR_INVALID, "az_nms_batched: bad arguments");
    if (n_groups == 0) return AZ_OK;
    const int total = offsets[n_groups];
    std::vector<int> small, large;
    for (int g = 0; g < n_groups; ++g) {
        const int n = offsets[g + 1] - offsets[g];
        if (n < 0) return fail(c, AZ_ERR_INVALID, "az_nms_batched: offsets must ascend");
        n_keep[g] = 0;
        if (n == 0) continue;
        (n <= azk_nms_small_max() ? small : large).push_back(g);
    }
    if (total > 0 && (!dets || !keep)) return fail(c, AZ_ERR_INVALID, "az_nms_batched: NULL array");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    static const bool poll = !(getenv("AZ_NMS_POLL") && !atoi(getenv("AZ_NMS_POLL")));
    if (!small.empty() && poll && !c->profiling && total <= 16384) {
        // The reference's call site (apply_nms: 20 classes x <= 100 boxes per image): everything -- boxes, offsets, group
        // list, keep lists, counts -- lives in ONE host-mapped block; one launch, no copy commands, and the host polls a
        // flag that the last workgroup to finish raises (a stream synchronisation plus five copies cost 70 of 96 us).
        const size_t o_off = ((size_t)total * 5 * sizeof(float) + 15) & ~(size_t)15;
        const size_t o_sel = o_off + (((size_t)n_groups + 1) * sizeof(int) + 15 & ~(size_t)15);
        const size_t o_keep = o_sel + ((small.size() * sizeof(int) + 15) & ~(size_t)15);
        const size_t o_nk = o_keep + (size_t)total * sizeof(long long);
        const size_t o_flag = o_nk + (((size_t)n_groups * sizeof(int) + 15) & ~(size_t)15);
        const size_t need = o_flag + 64;
        if (need > c->h_nmsb_cap) {
            HIPCHK(c, hipStreamSynchronize(s));
            if (c->h_nmsb) hipHostFree(c->h_nmsb);
            c->h_nmsb = nullptr; c->h_nmsb_cap = 0;
            HIPCHK(c, hipHostMalloc((void **)&c->h_nmsb, need + need / 2, hipHostMallocMapped));
            c->h_nmsb_cap = need + need / 2;
        }
        if (!c->nms_done) {
            HIPCHK(c, hipMalloc((void **)&c->nms_done, 16));
            HIPCHK(c, hipMemsetAsync(c->nms_done, 0, 16, s));
        }
        unsigned char *b = c->h_nmsb;
        std::memcpy(b, dets, (size_t)total * 5 * sizeof(float));
        std::memcpy(b + o_off, offsets, ((size_t)n_groups + 1) * sizeof(int));
        std::memcpy(b + o_sel, small.data(), small.size() * sizeof(int));
        std::memset(b + o_nk, 0, (size_t)n_groups * sizeof(int));
        volatile int *flag = (volatile int *)(b + o_flag);
        const int seq = ++c->nms_seq;
        const unsigned tag = nms_next_tag(c), want = tag & 0x3FFFFFu;
        *flag = 0;
        azk_nms_small(s, (const float *)b, (const int *)(b + o_off), (const int *)(b + o_sel), (int)small.size(), thresh,
                      (long long *)(b + o_keep), (int *)(b + o_nk), c->nms_done, (int *)(b + o_flag), seq, tag);
        // the flag says "all workgroups are done"; each count and keep word is still taken by its own tag (see nms_next_tag)
        const long long *hk = (const long long *)(b + o_keep);
        const volatile int *hn = (const volatile int *)(b + o_nk);
        auto all_tagged = [&](long spins) {
            for (int g : small) {
                long k = 0;
                while (((unsigned)hn[g] >> 9) != want) if (++k > spins) return false;
                if (!nms_keep_tagged(hk + offsets[g], (int)((unsigned)hn[g] & 0x1FFu), tag, spins)) return false;
            }
            return true;
        };
        bool got = false;
        for (int spin = 0; spin < 400000 && !got; ++spin) got = *flag == seq;
        if (got) got = all_tagged(200000);
        if (!got) HIPCHK(c, hipStreamSynchronize(s));
        HIPCHK(c, hipGetLastError());
        if (!all_tagged(0)) return fail(c, AZ_ERR_HIP, "az_nms_batched: the kernel left no result");
        for (int g : small) {
            const int nk = (int)((unsigned)hn[g] & 0x1FFu);
            if (nk > offsets[g + 1] - offsets[g]) return fail(c, AZ_ERR_HIP, "az_nms_batched: bad count");
            n_keep[g] = nk;
            for (int k = 0; k < nk; ++k) keep[offsets[g] + k] = (long long)(unsigned)(hk[(size_t)offsets[g] + k] & 0xFFFFFFFFll);
        }
    } else
    if (!small.empty()) {
        int rc;
        if ((rc = ev_grow(c, 0, &c->ev_a, (size_t)total * 5 * sizeof(float))) != AZ_OK) return rc;
        if ((rc = ev_grow(c, 1, &c->ev_b, ((size_t)n_groups + 1) * sizeof(int))) != AZ_OK) return rc;
        if ((rc = ev_grow(c, 2, &c->ev_c, small.size() * sizeof(int))) != AZ_OK) return rc;
        if ((rc = ev_grow(c, 3, &c->ev_d, (size_t)total * sizeof(long long))) != AZ_OK) return rc;
        if ((rc = ev_grow(c, 4, &c->ev_e, (size_t)n_groups * sizeof(int))) != AZ_OK) return rc;
        HIPCHK(c, hipMemcpyAsync(c->ev_a, dets, (size_t)total * 5 * sizeof(float), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->ev_b, offsets, ((size_t)n_groups + 1) * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->ev_c, small.data(), small.size() * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemsetAsync(c->ev_e, 0, (size_t)n_groups * sizeof(int), s));
        if (!(c->profiling & 4)) clear_events(c);
        { Timed t(c, "nms_batched", (int)small.size());
          azk_nms_small(s, (const float *)c->ev_a, (const int *)c->ev_b, (const int *)c->ev_c, (int)small.size(), thresh,
                        (long long *)c->ev_d, (int *)c->ev_e); }
        std::vector<long long> hk((size_t)total);
        HIPCHK(c, hipMemcpyAsync(hk.data(), c->ev_d, (size_t)total * sizeof(long long), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(n_keep, c->ev_e, (size_t)n_groups * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));           // `small`, `hk` live on this frame
        HIPCHK(c, hipGetLastError());
        for (int g : small)
            for (int k = 0; k < n_keep[g]; ++k) keep[offsets[g] + k] = hk[(size_t)offsets[g] + k];
    }
    for (int g : large) {
        int nk = 0;
        int rc = az_nms(c, dets + 5 * (size_t)offsets[g], offsets[g + 1] - offsets[g], thresh, keep + offsets[g], &nk);
        if (rc) return rc;
        n_keep[g] = nk;
    }
    return AZ_OK;
}

// --------------------------------------------------------------------------------------
// Tuner (lib/detect/tune.py): anchor history and the global k-th largest zoom score.
int az_last_anchors(az_ctx *c, double *regions_out, float *zoom_out, int cap, int *n_out)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!n_out) return AZ_ERR_INVALID;
    if (!(c->last.reserved & 4) || !c->hisB)
        return fail(c, AZ_ERR_STATE, "az_last_anchors: the last az_propose was not a tuner search (reserved bit 2)");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int n = c->his_n;
    *n_out = n;
    if (n > cap) return fail(c, AZ_ERR_CAPACITY, "az_last_anchors: cap too small");
    if (regions_out) HIPCHK(c, hipMemcpy(regions_out, c->hisB, (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost));
    if (zoom_out) HIPCHK(c, hipMemcpy(zoom_out, c->hisZ, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return AZ_OK;
}

int az_tune_begin(az_ctx *c, long long capacity)
{
    if (!c || capacity <= 0) return fail(c, AZ_ERR_INVALID, "az_tune_begin: bad capacity");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (capacity > c->pool_cap) {
        if (c->pool) { hipFree(c->pool); hipFree(c->pool_tmp); c->pool = c->pool_tmp = nullptr; c->pool_cap = 0; }
        HIPCHK(c, hipMalloc((void **)&c->pool, (size_t)capacity * sizeof(float)));
        HIPCHK(c, hipMalloc((void **)&c->pool_tmp, (size_t)capacity * sizeof(float)));
        c->pool_cap = capacity;
    }
    if (!c->pool_n) {
        HIPCHK(c, hipMalloc((void **)&c->pool_n, 4 * sizeof(unsigned long long)));
        HIPCHK(c, hipMalloc((void **)&c->pool_hist, 256 * sizeof(unsigned long long)));
    }
    HIPCHK(c, hipMemsetAsync(c->pool_n, 0, 4 * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AZ_OK;
}

int az_tune_end(az_ctx *c)
{
    if (!c) return AZ_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->pool) { hipFree(c->pool); hipFree(c->pool_tmp); }
    c->pool = c->pool_tmp = nullptr;
    c->pool_cap = 0;
    return AZ_OK;
}

static int pool_size(az_ctx *c, long long *n, long long *dropped)
{
    if (!c->pool) return fail(c, AZ_ERR_STATE, "az_tune_begin has not been called");
    unsigned long long h[2];
    HIPCHK(c, hipMemcpyAsync(h, c->pool_n, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n = (long long)h[0];
    *dropped = (long long)h[1];
    return AZ_OK;
}

int az_tune_push(az_ctx *c, const float *scores, long long n)
{
    if (!c || n < 0 || (n && !scores)) return fail(c, AZ_ERR_INVALID, "az_tune_push: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    long long have, dropped;
    int rc = pool_size(c, &have, &dropped);
    if (rc) return rc;
    if (have + n > c->pool_cap) return fail(c, AZ_ERR_CAPACITY, "az_tune_push: pool capacity exceeded");
    if (n) HIPCHK(c, hipMemcpyAsync(c->pool + have, scores, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const unsigned long long nn = (unsigned long long)(have + n);
    HIPCHK(c, hipMemcpyAsync(c->pool_n, &nn, sizeof(nn), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AZ_OK;
}

// MSB-first radix select over order-preserving keys: the key of the k-th largest score.
static int pool_kth_key(az_ctx *c, long long n, long long k, unsigned int *key_out)
{
    unsigned int prefix = 0;
    long long want = k;                       // rank (1-based, from the top) inside the current bucket
    unsigned long long h[256];
    for (int shift = 24; shift >= 0; shift -= 8) {
        HIPCHK(c, hipMemsetAsync(c->pool_hist, 0, sizeof(h), c->stream));
        azk_pool_hist(c->stream, c->pool, n, prefix, shift, c->pool_hist);
        HIPCHK(c, hipMemcpyAsync(h, c->pool_hist, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        int b = 255;
        for (; b > 0; --b) {
            if ((long long)h[b] >= want) break;
            want -= (long long)h[b];
        }
        prefix |= (unsigned int)b << shift;
    }
    *key_out = prefix;
    return AZ_OK;
}

static float key_to_float(unsigned int k)
{
    const unsigned int u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    std::memcpy(&f, &u, sizeof(f));
    return f;
}

int az_tune_kth_largest(az_ctx *c, long long k, float *value_out, long long *n_total)
{
    if (!c || k <= 0 || !value_out) return fail(c, AZ_ERR_INVALID, "az_tune_kth_largest: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    long long n, dropped;
    int rc = pool_size(c, &n, &dropped);
    if (rc) return rc;
    if (n_total) *n_total = n;
    if (dropped) return fail(c, AZ_ERR_CAPACITY, "az_tune: score pool overflowed; raise az_tune_begin's capacity");
    if (n <= k) { *value_out = -INFINITY; return AZ_OK; }     // the heap of tune.py:343-350 never overflowed
    unsigned int key;
    if ((rc = pool_kth_key(c, n, k, &key)) != AZ_OK) return rc;
    *value_out = key_to_float(key);
    return AZ_OK;
}

int az_tune_top(az_ctx *c, long long k, float *scores_out, long long cap, long long *n_out)
{
    if (!c || k <= 0 || !n_out) return fail(c, AZ_ERR_INVALID, "az_tune_top: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    long long n, dropped;
    int rc = pool_size(c, &n, &dropped);
    if (rc) return rc;
    if (dropped) return fail(c, AZ_ERR_CAPACITY, "az_tune: score pool overflowed; raise az_tune_begin's capacity");
    unsigned int key = 0;
    if (n > k && (rc = pool_kth_key(c, n, k, &key)) != AZ_OK) return rc;
    HIPCHK(c, hipMemsetAsync(&c->pool_n[2], 0, sizeof(unsigned long long), c->stream));
    azk_pool_keep(c->stream, c->pool, n, key, c->pool_tmp, &c->pool_n[2]);
    unsigned long long m = 0;
    HIPCHK(c, hipMemcpyAsync(&m, &c->pool_n[2], sizeof(m), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *n_out = (long long)m;
    if ((long long)m > cap) return fail(c, AZ_ERR_CAPACITY, "az_tune_top: cap too small");
    if (m && scores_out) HIPCHK(c, hipMemcpy(scores_out, c->pool_tmp, (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
    return AZ_OK;
}

// --------------------------------------------------------------------------------------
// Recall evaluation (lib/datasets/imdb.py:120-159) and utils.cython_bbox.bbox_overlaps.
int az_bbox_overlaps(az_ctx *c, const double *boxes, int N, const double *query, int K, double *overlaps_out)
{
    if (!c || N < 0 || K < 0 || ((N && !boxes) || (K && !query)) || (N && K && !overlaps_out))
        return fail(c, AZ_ERR_INVALID, "az_bbox_overlaps: bad arguments");
    if (N == 0 || K == 0) return AZ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ev_grow(c, 0, &c->ev_a, (size_t)N * 4 * sizeof(double))) != AZ_OK) return rc;
    if ((rc = ev_grow(c, 1, &c->ev_b, (size_t)K * 4 * sizeof(double))) != AZ_OK) return rc;
    if ((rc = ev_grow(c, 2, &c->ev_c, (size_t)N * K * sizeof(double))) != AZ_OK) return rc;
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->ev_a, boxes, (size_t)N * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->ev_b, query, (size_t)K * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    azk_bbox_overlaps(s, (const double *)c->ev_a, N, (const double *)c->ev_b, K, (double *)c->ev_c);
    HIPCHK(c, hipMemcpyAsync(overlaps_out, c->ev_c, (size_t)N * K * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return AZ_OK;
}

int az_recall_match(az_ctx *c, int n_images, const double *boxes, const int32_t *box_off, const double *gt,
                    const int32_t *gt_off, double *gt_overlaps_out)
{
    if (!c || n_images < 0 || (n_images && (!box_off || !gt_off)))
        return fail(c, AZ_ERR_INVALID, "az_recall_match: bad arguments");
    if (n_images == 0) return AZ_OK;
    const int NB = box_off[n_images], NG = gt_off[n_images];
    std::vector<long long> ov_off((size_t)n_images + 1, 0);
    for (int i = 0; i < n_images; ++i) {
        const long long n = box_off[i + 1] - box_off[i], k = gt_off[i + 1] - gt_off[i];
        if (n < 0 || k < 0) return fail(c, AZ_ERR_INVALID, "az_recall_match: offsets must ascend");
        if (n == 0 && k > 0)
            return fail(c, AZ_ERR_INVALID, "az_recall_match: an image without boxes (imdb.py:128-129 skips those)");
        ov_off[i + 1] = ov_off[i] + n * k;
    }
    if (NG == 0) return AZ_OK;
    if (!boxes || !gt || !gt_overlaps_out) return fail(c, AZ_ERR_INVALID, "az_recall_match: NULL array");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    const size_t offb = ((size_t)n_images + 1) * sizeof(int32_t);
    if ((rc = ev_grow(c, 0, &c->ev_a, (size_t)NB * 4 * sizeof(double))) != AZ_OK) return rc;
    if ((rc = ev_grow(c, 1, &c->ev_b, (size_t)NG * 4 * sizeof(double))) != AZ_OK) return rc;
    if ((rc = ev_grow(c, 2, &c->ev_c, (size_t)ov_off[n_images] * sizeof(double) + 8)) != AZ_OK) return rc;
    if ((rc = ev_grow(c, 3, &c->ev_d, offb)) != AZ_OK) return rc;
    if ((rc = ev_grow(c, 4, &c->ev_e, offb)) != AZ_OK) return rc;
    if ((rc = ev_grow(c, 5, &c->ev_f, ((size_t)n_images + 1) * sizeof(long long))) != AZ_OK) return rc;
    if ((rc = ev_grow(c, 6, &c->ev_g, (size_t)NG * sizeof(double))) != AZ_OK) return rc;
    if ((rc = ev_grow(c, 7, &c->ev_h, (size_t)n_images * sizeof(int))) != AZ_OK) return rc;
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->ev_a, boxes, (size_t)NB * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->ev_b, gt, (size_t)NG * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->ev_d, box_off, offb, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->ev_e, gt_off, offb, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->ev_f, ov_off.data(), ((size_t)n_images + 1) * sizeof(long long), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemsetAsync(c->ev_h, 0, (size_t)n_images * sizeof(int), s));
    azk_recall_match(s, n_images, (const double *)c->ev_a, (const int *)c->ev_d, (const double *)c->ev_b,
                     (const int *)c->ev_e, (const long long *)c->ev_f, (double *)c->ev_c, (double *)c->ev_g,
                     (int *)c->ev_h);
    std::vector<int> bad((size_t)n_images);
    HIPCHK(c, hipMemcpyAsync(gt_overlaps_out, c->ev_g, (size_t)NG * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(bad.data(), c->ev_h, (size_t)n_images * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));     // ov_off / bad live on this frame
    for (int i = 0; i < n_images; ++i)
        if (bad[i])
            return fail(c, AZ_ERR_INVALID,
                        "az_recall_match: image " + std::to_string(i) +
                            " has more ground-truth boxes than candidates (assert gt_ovr >= 0, imdb.py:139)");
    return AZ_OK;
}

// --------------------------------------------------------------------------------------
// Image front-end (_get_image_blob, lib/detect/test.py:27-59).
int az_image_blob_size(int h, int w, double scale, int *oh, int *ow)
{
    if (h <= 0 || w <= 0 || !(scale > 0) || !oh || !ow) return AZ_ERR_INVALID;
    *oh = (int)std::nearbyint((double)h * scale);      // cv2: saturate_cast<int>(rows * fy), ties to even
    *ow = (int)std::nearbyint((double)w * scale);
    return (*oh > 0 && *ow > 0) ? AZ_OK : AZ_ERR_INVALID;
}

static int image_blob_common(az_ctx *c, const uint8_t *im, int h, int w, const float *means, double scale,
                             float *out, bool out_is_dev, int oh, int ow)
{
    int eh, ew;
    if (!c || !im || !means || !out || az_image_blob_size(h, w, scale, &eh, &ew) != AZ_OK || eh != oh || ew != ow)
        return fail(c, AZ_ERR_INVALID, "az_image_blob: bad arguments (output size must come from az_image_blob_size)");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    const size_t nin = (size_t)h * w * 3, nout = (size_t)oh * ow * 3;
    if ((rc = ev_grow(c, 0, &c->ev_a, nin)) != AZ_OK) return rc;
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(c->ev_a, im, nin, hipMemcpyHostToDevice, s));
    float *dst = out;
    if (!out_is_dev) {
        if ((rc = ev_grow(c, 1, &c->ev_b, nout * sizeof(float))) != AZ_OK) return rc;
        dst = (float *)c->ev_b;
    }
    azk_image_blob(s, (const unsigned char *)c->ev_a, h, w, means, 1.0 / scale, 1.0 / scale, oh, ow, dst);
    if (!out_is_dev) HIPCHK(c, hipMemcpyAsync(out, dst, nout * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return AZ_OK;
}

int az_image_blob_host(az_ctx *c, const uint8_t *im, int h, int w, const float *means, double scale, float *blob_out,
                       int oh, int ow)
{
    return image_blob_common(c, im, h, w, means, scale, blob_out, false, oh, ow);
}

int az_image_blob_dev(az_ctx *c, const uint8_t *im, int h, int w, const float *means, double scale, float *blob_dev,
                      int oh, int ow)
{
    return image_blob_common(c, im, h, w, means, scale, blob_dev, true, oh, ow);
}


int az_image_blob_dev_on(az_ctx *c, const uint8_t *im, int h, int w, const float *means, double scale, float *blob_dev,
                         int oh, int ow, void *stream)
{
    int eh, ew;
    if (!c || !im || !means || !blob_dev || az_image_blob_size(h, w, scale, &eh, &ew) != AZ_OK || eh != oh || ew != ow)
        return fail(c, AZ_ERR_INVALID, "az_image_blob_dev_on: bad arguments (output size must come from az_image_blob_size)");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const size_t nin = (size_t)h * w * 3;
    if (nin > c->io_cap) {
        // (larger images: nothing may still be reading the old slots)
        for (auto &q : c->io) {
            if (q.ev) HIPCHK(c, hipEventSynchronize(q.ev));
            if (q.host) hipHostFree(q.host);
            if (q.dev) hipFree(q.dev);
            if (q.ev) hipEventDestroy(q.ev);
        }
        c->io.clear();
        c->io_cap = nin + nin / 4 + 256;
        c->io_turn = 0;
    }
    auto add_slot = [&](size_t at) {
        az_ctx::IoSlot q;
        if (hipHostMalloc((void **)&q.host, c->io_cap) != hipSuccess || hipMalloc((void **)&q.dev, c->io_cap) != hipSuccess ||
            hipEventCreateWithFlags(&q.ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (q.host) hipHostFree(q.host);
            if (q.dev) hipFree(q.dev);
            if (q.ev) hipEventDestroy(q.ev);
            return false;
        }
        c->io.insert(c->io.begin() + (long)at, q);
        return true;
    };
    while (c->io.size() < 2)
        if (!add_slot(c->io.size())) return fail(c, AZ_ERR_HIP, "az_image_blob_dev_on: upload slots");
    if (c->io_turn >= (int)c->io.size()) c->io_turn = 0;
    // the slot in turn holds the oldest upload: done long ago in a loop that fetches what it launches; still QUEUED when the
    // caller enqueues a whole batch behind the previous batch's search -- then a fresh slot takes this image (it goes in
    // front of the busy one: the ring stays oldest-first) rather than the host waiting for the GPU
    if (hipEventQuery(c->io[c->io_turn].ev) == hipErrorNotReady && (int)c->io.size() < az_ctx::IO_SLOTS_MAX)
        (void)add_slot((size_t)c->io_turn);
    (void)hipGetLastError();
    az_ctx::IoSlot &q = c->io[c->io_turn];
    c->io_turn = (c->io_turn + 1) % (int)c->io.size();
    HIPCHK(c, hipEventSynchronize(q.ev));                   // (a fresh slot's event was never recorded: returns at once)
    std::memcpy(q.host, im, nin);                           // the caller's array may go away as soon as this returns
    HIPCHK(c, hipMemcpyAsync(q.dev, q.host, nin, hipMemcpyHostToDevice, s));
    azk_image_blob(s, q.dev, h, w, means, 1.0 / scale, 1.0 / scale, oh, ow, blob_dev);
    HIPCHK(c, hipEventRecord(q.ev, s));
    HIPCHK(c, hipGetLastError());
    return AZ_OK;
}


}  // extern "C"
