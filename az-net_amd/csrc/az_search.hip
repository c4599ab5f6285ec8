// az_search.hip -- the forms a search takes and the launch sequence of each: head-pass cost model (measured on the device),
// pair / whole-tree / closure / one-pass plans and their per-shape caches, the level loop as one stream-ordered launch
// sequence (optionally a hipGraph), and collecting a result -- including running a search again in another form.
#include "az_ctx.h"

// Which form of the search a call takes.
struct SearchPlan { int n_spec; bool fused, fused_lv, defer_root; int pair_mask; int lv_limit; int full; /* 0 / 1 tree rows / 2 closure */
                    int cut; /* > 0: nothing is enqueued from this level on (the tree is expected to end before it) */ };

// Cost of one head pass (RoIPool, int6, reduce, int7, heads) at `rows` rois, in us: measured on this device at a few row
// counts the first time the context launches a search (calibrate_passes) and interpolated; until then (or with
// AZ_PASS_CAL=0) the figures of the round-3 profiles: weight-streaming bound up to ~40 rows, then ~1.4 us per row.
// What a level costs besides its head pass (its geometry kernel and the kernel boundaries) is GEOM_US; a window lookup
// stage LOOKUP_US.
static double pass_us(const az_ctx *c, double rows)
{
    const auto &k = c->cal;
    if (k.state == 1 && k.n >= 2) {
        if (rows <= k.rows[0]) return k.us[0];
        for (int i = 1; i < k.n; ++i)
            if (rows <= k.rows[i] || i == k.n - 1)
                return k.us[i - 1] + (k.us[i] - k.us[i - 1]) * (rows - k.rows[i - 1]) / (double)(k.rows[i] - k.rows[i - 1]);
    }
    // (int6 on the 16-bit matrix cores, az_set_gemm_mode 2 / 3: a row costs a fraction of that, a launch somewhat more.
    //  Measured: two terms 100-113 us at 48 rows, 365 us at 670; three terms 125 us and 630 us -- int6 alone)
    double t;
    if (c->gemm_parts == 2) { t = 85.0 + 0.42 * rows; t = t < 100.0 ? 100.0 : t; }
    else if (c->gemm_parts == 3) { t = 110.0 + 0.78 * rows; t = t < 130.0 ? 130.0 : t; }
    else { t = 60.0 + 1.4 * rows; t = t < 92.0 ? 92.0 : t; }
    return t + 50.0;
}
static inline hipStream_t geom_stream(const az_ctx *c) { return c->gs ? c->gs : c->stream; }   // (az_ctx.h: two stages)
constexpr double PASS_OVERHEAD_US = 40.0, LOOKUP_US = 8.0;     // (PASS_OVERHEAD_US: the level's geometry kernel + boundaries)
constexpr unsigned AZ_TAB_ROOT_HOST = 0x1FFFu;      // (az_geom_dev.h: AZ_TAB_ROOT)

// Measure pass_us on this device: whole head passes over synthetic rois (a grid of ~64-px boxes on the current map) at a
// few row counts, HIP events on the ctx stream, best of three each; ~10 ms, once per context, outside any capture and with
// no search queued.  The forms' costs differ by tens of us per image and boxes of one pool differ by 5-10 %: literals tuned
// on one box pick the wrong form on another.  AZ_PASS_CAL=0 keeps the literals.
static int calibrate_passes(az_ctx *c)
{
    auto &k = c->cal;
    if (k.state != 0) return AZ_OK;
    { const char *e = getenv("AZ_PASS_CAL"); if (e && !atoi(e)) { k.state = -1; return AZ_OK; } }
    if (!c->feat || !c->pend.empty() || c->d.H <= 0 || c->d.W <= 0) return AZ_OK;       // (next time)
    join_s2(c);
    k.state = -1;                                                                      // (any failure below: literals)
    hipStream_t s = c->stream;
    const int sizes[] = {48, 112, 176, 352, 704, 1408};
    int nsz = 0;
    for (int v : sizes) if (v + 1 < c->maxR) ++nsz;
    if (nsz < 2) return AZ_OK;
    const int maxrows = sizes[nsz - 1];
    {   // rois: boxes of ~4 x 4 map cells walking over the map (what the deep levels look like)
        std::vector<float> r((size_t)maxrows * 5);
        const float fw = (float)c->d.W / c->spatial_scale, fh = (float)c->d.H / c->spatial_scale;
        for (int i = 0; i < maxrows; ++i) {
            const float x = fmodf(37.0f * i, fw > 80.f ? fw - 72.f : 1.f), y = fmodf(53.0f * i, fh > 80.f ? fh - 72.f : 1.f);
            r[5 * (size_t)i] = 0.f; r[5 * (size_t)i + 1] = x; r[5 * (size_t)i + 2] = y;
            r[5 * (size_t)i + 3] = x + 63.f; r[5 * (size_t)i + 4] = y + 63.f;
        }
        HIPCHK(c, hipMemcpyAsync(c->urois, r.data(), r.size() * sizeof(float), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    hipEvent_t ea = nullptr, eb = nullptr;
    if (hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) {
        if (ea) hipEventDestroy(ea);
        (void)hipGetLastError();
        return AZ_OK;
    }
    const int prof = c->profiling;
    c->profiling = 0;
    c->cand_n = -1;
    bool ok = true;
    for (int i = 0; i < nsz && ok; ++i) {
        HIPCHK(c, hipMemsetAsync(c->cnt, 0, sizeof(AzCounts), s));
        ok = set_count(c, &c->cnt->U[0], sizes[i]) == AZ_OK;
        double best = 1e30;
        for (int rep = 0; rep < 4 && ok; ++rep) {
            prep_scale(c);
            ok = hipEventRecord(ea, s) == hipSuccess;
            launch_head(c, &c->cnt->U[0], 0, 1, 1, 0.0, c->zoom_u, c->score_u, c->delta_u, 0.0, false, 0, nullptr, nullptr, sizes[i]);
            ok = ok && hipEventRecord(eb, s) == hipSuccess && hipEventSynchronize(eb) == hipSuccess;
            float ms = 0.f;
            ok = ok && hipEventElapsedTime(&ms, ea, eb) == hipSuccess;
            if (rep > 0 && ms * 1e3 < best) best = ms * 1e3;
        }
        k.rows[i] = sizes[i]; k.us[i] = best;
    }
    hipEventDestroy(ea); hipEventDestroy(eb);
    c->profiling = prof;
    c->npass = 0;
    (void)hipGetLastError();
    if (!ok) return AZ_OK;
    for (int i = 1; i < nsz; ++i) if (!(k.us[i] > k.us[i - 1])) k.us[i] = k.us[i - 1] + 1.0;    // (monotone)
    k.n = nsz;
    k.state = 1;
    if (getenv("AZ_FULL_DEBUG")) {
        fprintf(stderr, "az: head-pass cost on this device (rows: us):");
        for (int i = 0; i < nsz; ++i) fprintf(stderr, " %d: %.1f", k.rows[i], k.us[i]);
        fprintf(stderr, "\n");
    }
    return AZ_OK;
}

// Pair speculation: the head pass of level l also evaluates one row per distinct RoIPool window among ALL children of
// its regions, so that level l+1 needs no pass of its own (az_level.hip).  Worth it when most regions zoom: the extra
// rows are then few more than level l+1 would have forwarded anyway, and a whole pass (one stream of the 411 MB int6
// weights for small levels, the reduce / int7 / heads / geometry chain always) disappears.  The decision comes from
// the previous search of this context on the same image shape (what a dataset run looks like); without history
// nothing is speculated.  params.reserved bit 6 / AZ_PAIR_SPEC=0: never; bit 7 / AZ_PAIR_SPEC=2: at every eligible
// level (tests).  Results are bit-identical either way.
// The records of the shape's history: r = 0 the last search (the hint_* fields), r = 1.. the ones before it.
struct HintView { const int *rows, *P, *PZ, *U, *SPN; };
static HintView hint_rec(const az_ctx *c, int r)
{
    if (r == 0) return {c->hint_rows, c->hint_P, c->hint_PZ, c->hint_U, c->hint_SPN};
    const auto &o = c->hint_old[r - 1];
    return {o.rows, o.P, o.PZ, o.U, o.SPN};
}
constexpr double EMPTY_LEVEL_US = 35.0;    // an enqueued level whose row count turns out to be zero: five launches + a geometry kernel that leave at once

// rows a pair-speculating pass of level l carries for level l+1, for one recorded tree: what it carried then, else level l+1's
// unique rois scaled by parents / zoomed parents, else (the tree ended at level l) ~4.5 windows per region
static double pair_rows(const HintView &v, int l)
{
    if (v.SPN[l] >= 0) return (double)v.SPN[l];
    if (v.U[l + 1] > 0) return (double)v.U[l + 1] * v.P[l] / (v.PZ[l] > 0 ? v.PZ[l] : 1);
    return 4.5 * v.P[l];
}

static int pair_plan(az_ctx *c, const az_params *p, int nlev, int n_spec, bool fused_lv, int lv_limit)
{
    if (c->pair_env < 0) { const char *e = getenv("AZ_PAIR_SPEC"); c->pair_env = e ? atoi(e) : 1; }
    if (!fused_lv || (p->reserved & 64) || c->pair_env == 0) return 0;
    for (const auto &hw : c->nopair)
        if (hw.first == p->im_h && hw.second == p->im_w) return 0;
    const bool force = (p->reserved & 128) || c->pair_env == 2;
    const bool hist = c->hint_h == p->im_h && c->hint_w == p->im_w && c->hint_nlev == nlev && c->hint_n > 0;
    int mask = 0;
    for (int l = n_spec; l + 1 < nlev && l < lv_limit; ++l) {      // (the lookup runs in level l's fused geometry kernel)
        bool want = force;
        if (!want && hist) {
            // expected cost over the shape's recorded trees that reached level l (the others pay nothing here either way)
            double with = 0.0, without = 0.0;
            int n = 0;
            bool fits = true;
            for (int r = 0; r < c->hint_n; ++r) {
                const HintView v = hint_rec(c, r);
                if (v.P[l] <= 0) continue;
                const double S = pair_rows(v, l);
                with += pass_us(c, v.U[l] + S) + PASS_OVERHEAD_US + LOOKUP_US;
                without += pass_us(c, v.U[l]) + PASS_OVERHEAD_US +
                           (v.U[l + 1] > 0 ? pass_us(c, v.U[l + 1]) + PASS_OVERHEAD_US : EMPTY_LEVEL_US);
                fits = fits && v.U[l] + S + 2 < c->maxR;
                ++n;
            }
            want = n > 0 && with < without && fits;
        }
        if (want) { mask |= 1 << l; ++l; }          // level l+1 is looked up: it has no pass to carry rows
    }
    return mask;
}

static bool plan_is_for(const az_ctx::StaticPlan &k, const az_params *p, int nlev);

static SearchPlan plan_search(az_ctx *c, const az_params *p, int nlev, bool tune)
{
    SearchPlan q;
    q.n_spec = (nlev >= 3 && !(p->reserved & 1) && !tune) ? 3 : 0;
    // The geometry of those three levels is a few dozen elements per stage: by default it runs
    // inside single-workgroup kernels (az_fused.hip) instead of ~40 tiny launches.
    // (params.reserved bit 1 keeps the multi-launch form; same bits, for tests.)
    q.fused = q.n_spec && !(p->reserved & 2) && !(p->im_h == c->nofuse_h && p->im_w == c->nofuse_w);
    // Levels after the speculative ones: one single-workgroup kernel per mid-tree level (az_level.hip) instead of
    // ten launches (params.reserved bit 4 / AZ_LEVEL_FUSED=0 keep the multi-launch form; same bits).
    if (c->level_fused_env < 0) { const char *e = getenv("AZ_LEVEL_FUSED"); c->level_fused_env = (e && !atoi(e)) ? 0 : 1; }
    q.fused_lv = q.fused && nlev > q.n_spec && !(p->reserved & 16) && c->level_fused_env &&
                 !(p->im_h == c->nofuse_lv_h && p->im_w == c->nofuse_lv_w);
    // The root's row (zoom forced, candidates only needed by the final selection) moves from the speculative
    // pass to the first fused level's head pass: 48 rows = 1.5 strips instead of 49 = 2 for a 600x1000 image
    // (AZ_DEFER_ROOT=0 keeps it in the speculative pass; same bits).  That level must be a mid-tree one.
    if (c->defer_root_env < 0) { const char *e = getenv("AZ_DEFER_ROOT"); c->defer_root_env = (e && !atoi(e)) ? 0 : 1; }
    q.defer_root = q.fused_lv && q.n_spec == 3 && nlev >= q.n_spec + 2 && c->defer_root_env;
    // ... and must exist: a tree that ends before it would pay a whole head pass for the root's one row (measured: a
    // [1, 8, 0, 0, 0] tree 0.43 ms deferred against 0.32).  The previous search of this image shape tells.
    if (q.defer_root && c->hint_h == p->im_h && c->hint_w == p->im_w && c->hint_nlev == nlev && c->hint_P[q.n_spec] == 0)
        q.defer_root = false;
    // (round 5: a stream of different images -- deferring gains 16 us when the tree reaches that level and costs a whole
    //  one-row head pass, ~110 us, when it does not: only when every one of the context's last four searches got there)
    if (q.defer_root && c->n_hist < 4) q.defer_root = false;
    for (int i = 0; i < 4 && q.defer_root; ++i)
        if ((int)((c->early_hist >> (4 * i)) & 15u) <= q.n_spec) q.defer_root = false;
    q.lv_limit = AZ_MAX_LEVELS + 1;
    for (const auto &e : c->lv_limits)
        if (e.h == p->im_h && e.w == p->im_w) q.lv_limit = e.limit;
    q.pair_mask = pair_plan(c, p, nlev, q.n_spec, q.fused_lv, q.lv_limit);
    // whole-tree speculation (decided and prepared by az_propose_launch: full_prepare): one head pass over the rows of
    // the image shape's full tree, every level's outputs by window lookup -- no deferred root, no pair rows
    q.full = (c->full_now && q.fused && q.fused_lv && q.n_spec == 3 && q.lv_limit >= q.n_spec && c->plan &&
              c->plan->fs[c->full_now - 1].full_state == 1 && plan_is_for(*c->plan, p, nlev)) ? c->full_now : 0;
    if (q.full) { q.defer_root = false; q.pair_mask = 0; }
    // early end: recent searches of this context had no regions from level `cut` on (a level the fused kernels hand over
    // to: the one before it carries the check).  Two rules, by what a miss costs (round 5; az_ctx.h: early_hist):
    //   cut == 2 (the tree is the root and its children): a hit saves the third level's 40 rows and two empty levels
    //            (~70 us of ~170), a miss wastes the 9-row pass (~100 us) -- taken when at least 7 of the context's last 8
    //            searches ended there, whatever the very last one did;
    //   cut >= 3: a miss repeats a search that has already run most of its passes -- taken only when the last four all
    //            ended at or before that level.
    q.cut = 0;
    if (c->cut_env < 0) { const char *e = getenv("AZ_EARLY_END"); c->cut_env = (e && !atoi(e)) ? 0 : 1; }
    if (c->cut_env && !(p->reserved & 4096) && !q.full && !tune && q.fused && q.fused_lv) {
        auto ended_by = [&](int i, int l) { return (int)((c->early_hist >> (4 * i)) & 15u) <= l; };
        if (q.n_spec == 3 && nlev > 2) {
            int n2 = 0;
            for (int i = 0; i < 8; ++i) n2 += ended_by(i, 2) ? 1 : 0;
            if (n2 >= 7) q.cut = 2;
        }
        for (int l = q.n_spec; !q.cut && l < nlev; ++l) {
            bool all = true;
            for (int i = 0; i < 4 && all; ++i) all = ended_by(i, l);
            if (all) q.cut = l;
        }
        if (q.cut > q.n_spec && q.cut - 1 >= q.lv_limit) q.cut = 0;      // (the level before it runs on the multi-launch kernels)
        if (q.cut && q.cut < q.n_spec && q.defer_root) q.defer_root = false;   // (a deferred root needs level 4 to exist)
    }
    return q;
}

// The speculative pre-pass (B1 = divide_region(root), all children of B1, the rois of the speculative rows) is a
// function of the image shape alone: run once per shape, outside any graph capture, its outputs kept in
// dedicated buffers and its three counters on the host; k_spec_levels restores them for every search.
static int ensure_spec_cache(az_ctx *c, const az_params *p, const SearchPlan &q)
{
    if (!q.fused) return AZ_OK;
    const int defer = q.defer_root ? 1 : 0;
    auto &k = c->spc[defer];
    if (k.h == p->im_h && k.w == p->im_w && k.scale == p->scale && k.min_side == p->min_side)
        return AZ_OK;
    auto use = [&](az_ctx::SpecEntry &e) {
        c->spec_urois[defer] = e.urois; c->specB1[defer] = e.B1; c->spec_choff[defer] = e.choff; c->spec_U[defer] = e.Udev;
        k.h = e.h; k.w = e.w; k.scale = e.scale; k.min_side = e.min_side; k.P1 = e.P1; k.CH = e.CH; k.U = e.U;
        e.use = ++c->spec_clock;
    };
    for (auto &e : c->spec_store)
        if (e.h == p->im_h && e.w == p->im_w && e.defer == defer && e.scale == p->scale && e.min_side == p->min_side) {
            use(e);
            return AZ_OK;
        }
    join_s2(c);                        // (the pre-pass works in the per-search buffers)
    hipStream_t s = c->stream;
    azk_spec_prepass(s, c->cnt, c->B[0], c->spec_scr_B1[defer], c->child, c->spec_scr_choff[defer], c->spec_scr_urois[defer],
                     p->scale, p->min_side, c->maxR, c->maxCh, p->im_h, p->im_w, defer);
    HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->cnt, sizeof(AzCounts), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (c->h_cnt->err) {               // the speculative rows outgrow the context: take the multi-launch path
        c->nofuse_h = p->im_h; c->nofuse_w = p->im_w;
        k.h = -1;
        return AZ_OK;
    }
    az_ctx::SpecEntry e;
    e.h = p->im_h; e.w = p->im_w; e.defer = defer; e.scale = p->scale; e.min_side = p->min_side;
    e.P1 = c->h_cnt->specP1; e.CH = c->h_cnt->specCH; e.U = c->h_cnt->specU;
    if (hipMalloc((void **)&e.urois, (size_t)(e.U + 1) * 5 * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&e.B1, (size_t)(e.P1 + 1) * 4 * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&e.choff, (size_t)(e.P1 + 1) * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&e.Udev, 16) != hipSuccess) {
        for (void *q2 : {(void *)e.urois, (void *)e.B1, (void *)e.choff, (void *)e.Udev}) if (q2) hipFree(q2);
        return fail(c, AZ_ERR_HIP, "hipMalloc failed for a speculative pre-pass entry");
    }
    HIPCHK(c, hipMemcpyAsync(e.urois, c->spec_scr_urois[defer], (size_t)e.U * 5 * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(e.B1, c->spec_scr_B1[defer], (size_t)e.P1 * 4 * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(e.choff, c->spec_scr_choff[defer], (size_t)e.P1 * sizeof(int), hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(e.Udev, &c->cnt->specU, sizeof(int), hipMemcpyDeviceToDevice, s));
    const int rows_short = (defer ? 0 : 1) + e.P1;        // the pass without the third level's rows (early end: plan.cut == 2)
    HIPCHK(c, hipMemcpyAsync(e.Udev + 1, &rows_short, sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (c->spec_store.size() >= 128) {
        // drop the least recently used entry; captured launch sequences may hold its pointers: drop those too
        size_t lru = 0;
        for (size_t i = 1; i < c->spec_store.size(); ++i) if (c->spec_store[i].use < c->spec_store[lru].use) lru = i;
        for (auto &g : c->graphs) hipGraphExecDestroy(g.second.exec);
        c->graphs.clear();
        auto &d = c->spec_store[lru];
        for (int i = 0; i < 2; ++i) if (c->spec_urois[i] == d.urois) { c->spc[i].h = -1; }
        for (void *q2 : {(void *)d.urois, (void *)d.B1, (void *)d.choff, (void *)d.Udev}) hipFree(q2);
        c->spec_store.erase(c->spec_store.begin() + (long)lru);
    }
    c->spec_store.push_back(e);
    use(c->spec_store.back());
    return AZ_OK;
}

// Final selection (test.py:392-400): top-k by score, or everything with score >= Tc.
static void enqueue_select(az_ctx *c, const az_params *p, int nlev, int k)
{
    hipStream_t s = geom_stream(c);
    Timed t(c, "select", nlev);
    if (p->fixed_num)
        azk_topk_full(s, c->Sall, &c->cnt->ytot[nlev], c->maxCand, k, c->sel_idx, &c->cnt->nsel, c->Yall,
                      c->Sall, (double *)((unsigned char *)c->cnt + RES_HDR),
                      (float *)((unsigned char *)c->cnt + RES_HDR + (size_t)k * 32),
                      (p->reserved & 8) ? nullptr : c->rank_part);
    else
        azk_thresh_select_full(s, c->Sall, &c->cnt->ytot[nlev], c->maxCand, p->Tc, c->maxCand, c->sel_idx,
                               &c->cnt->nsel, c->Yall, c->Sall, c->Yout, c->Sout);
}

// ---- Tz <= 0: the tree is known before any score is (az_static.hip) -----------------------------------------------
// (params.reserved bits 0, 1, 2, 4 ask for one of the level-loop forms; bit 5 / AZ_STATIC_TREE=0 turn the plan off)
static bool static_wanted(az_ctx *c, const az_params *p, bool tune)
{
    if (c->static_env < 0) {
        const char *e = getenv("AZ_STATIC_TREE"), *f = getenv("AZ_FINAL_FUSED"), *g = getenv("AZ_PLAN_CACHE");
        c->static_env = (e && !atoi(e)) ? 0 : 1;
        c->final_env = (f && !atoi(f)) ? 0 : 1;
        if (g && atoi(g) > 0) c->plan_cache_max = atoi(g);
    }
    if (tune || !(p->Tz <= 0.0) || (p->reserved & (1 | 2 | 16 | 32)) || !c->static_env) return false;
    for (const auto &hw : c->nostatic)
        if (hw.first == p->im_h && hw.second == p->im_w) return false;
    return true;
}

static bool plan_is_for(const az_ctx::StaticPlan &k, const az_params *p, int nlev)
{
    return k.h == p->im_h && k.w == p->im_w && k.scale == p->scale && k.min_side == p->min_side &&
           k.dedup == p->dedup && k.batch == p->batch_size && k.nlev == nlev;
}

static bool static_plan_matches(const az_ctx *c, const az_params *p, int nlev)
{
    return c->plan && plan_is_for(*c->plan, p, nlev);
}

// All levels' regions with every region zoomed: the level loop's own geometry kernels (roi projection + dedup,
// divide_region + _sift_dup), run once per image shape, outside any graph capture.
static int ensure_static_plan(az_ctx *c, const az_params *p, int nlev)
{
    for (auto *q : c->plans)
        if (plan_is_for(*q, p, nlev)) { c->plan = q; q->last_use = ++c->plan_clock; return AZ_OK; }
    c->plan = nullptr;
    join_s2(c);                        // (the plan is built in the per-search buffers)
    hipStream_t s = c->stream;
    auto give_up = [&]() {
        if (c->nostatic.size() >= 32) c->nostatic.erase(c->nostatic.begin());
        c->nostatic.emplace_back(p->im_h, p->im_w);
        return (int)AZ_OK;
    };
    // (the plan under construction owns five device buffers until it is handed to the cache: freed on every other exit)
    struct PlanGuard { az_ctx::StaticPlan k; bool keep = false; ~PlanGuard() { if (!keep) free_plan(&k); } } pg;
    az_ctx::StaticPlan &k = pg.k;
    // Two passes over the tree: sizes first, then placement.  Rows of the one head pass: levels 2, 3, ... in order, the
    // root last (RoIPool treats that one whole-image roi cooperatively: a workgroup per bin instead of a wave.
    // Deepest level first with levels 1-3 cooperative was measured too: 26.2 us against 24.5).
    int uoff[AZ_MAX_LEVELS] = {0};
    int roff = 0;
    for (int pass = 0; pass < 2; ++pass) {
        azk_init_root(s, c->cnt, c->B[0], p->im_h, p->im_w);
        roff = 0;
        for (int l = 0; l < nlev; ++l) {
            const int cur = l & 1;
            azk_rois_dedup(s, c->B[cur], &c->cnt->P[l], c->maxR, p->scale, (float)p->dedup, p->batch_size, c->rois,
                           c->key, c->grp, c->first, c->index, c->inv, c->urois, c->ubox, &c->cnt->U[l]);
            if (l + 1 < nlev) {
                azk_divide(s, &c->cnt->P[l], &c->cnt->CH[l], &c->cnt->err, c->maxR, c->maxCh, c->B[cur], p->min_side,
                           c->choff, c->child, c->ckey, nullptr, nullptr, nullptr, 0, nullptr);
                azk_dedup_regions(s, c->ckey, &c->cnt->CH[l], c->maxCh, c->maxR, c->first, c->child, c->B[cur ^ 1],
                                  &c->cnt->P[l + 1], &c->cnt->err, nullptr, nullptr);
            }
            if (pass == 0) {
                HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->cnt, sizeof(AzCounts), hipMemcpyDeviceToHost, s));
                HIPCHK(c, hipStreamSynchronize(s));
                if (c->h_cnt->err) return give_up();
                k.roff[l] = roff; k.U[l] = c->h_cnt->U[l]; k.CH[l] = (l + 1 < nlev) ? c->h_cnt->CH[l] : 0;
                roff += c->h_cnt->P[l];
                if (l == 0 && (c->h_cnt->P[0] != 1 || k.U[0] != 1)) return give_up();
            } else {
                const int P = k.roff[l + 1] - k.roff[l], U = k.U[l];
                if (P > 0) {
                    HIPCHK(c, hipMemcpyAsync(k.urois + (size_t)uoff[l] * 5, c->urois, (size_t)U * 5 * sizeof(float),
                                             hipMemcpyDeviceToDevice, s));
                    HIPCHK(c, hipMemcpyAsync(k.ubox + (size_t)uoff[l] * 4, c->ubox, (size_t)U * 4 * sizeof(double),
                                             hipMemcpyDeviceToDevice, s));
                    azk_plan_rows(s, c->inv, &c->cnt->P[l], c->maxR, k.roff[l], uoff[l], k.reg_u);
                }
            }
        }
        if (pass == 0) {
            k.roff[nlev] = roff;
            int tot = 0;
            for (int l = 1; l < nlev; ++l) { uoff[l] = tot; tot += k.U[l]; }
            uoff[0] = tot;
            k.Utot = tot + 1;
            if (k.Utot > c->maxR || roff > c->maxR) return give_up();
            k.coop = 1;
            // exact-size buffers of this shape's plan
            auto grab = [&](void **q, size_t bytes) { return hipMalloc(q, bytes + 256) == hipSuccess; };
            if (!grab((void **)&k.urois, (size_t)k.Utot * 5 * sizeof(float)) ||
                !grab((void **)&k.ubox, (size_t)k.Utot * 4 * sizeof(double)) ||
                !grab((void **)&k.reg_u, (size_t)roff * sizeof(int)) ||
                !grab((void **)&k.cand_src, (size_t)roff * AZ_NSUB * sizeof(int)) || !grab((void **)&k.meta, 16))
                return fail(c, AZ_ERR_HIP, "hipMalloc failed for a static plan");
        }
    }
    if (hipMemcpyAsync(k.meta, &k.Utot, sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess ||
        (azk_plan_cands(s, k.reg_u, k.roff[nlev], k.cand_src), hipStreamSynchronize(s)) != hipSuccess)
        return fail(c, AZ_ERR_HIP, "static plan: copy failed");
    k.h = p->im_h; k.w = p->im_w; k.scale = p->scale; k.min_side = p->min_side; k.dedup = p->dedup;
    k.batch = p->batch_size; k.nlev = nlev;
    k.last_use = ++c->plan_clock;
    if (c->plan_cache_max < 1) c->plan_cache_max = 1;
    if ((int)c->plans.size() >= c->plan_cache_max) {
        // drop the least recently used shape; captured launch sequences may hold its pointers: drop those too
        size_t lru = 0;
        for (size_t i = 1; i < c->plans.size(); ++i) if (c->plans[i]->last_use < c->plans[lru]->last_use) lru = i;
        for (auto &g : c->graphs) hipGraphExecDestroy(g.second.exec);
        c->graphs.clear();
        free_plan(c->plans[lru]);
        delete c->plans[lru];
        c->plans.erase(c->plans.begin() + (long)lru);
    }
    c->plans.push_back(new az_ctx::StaticPlan(k));
    pg.keep = true;
    c->plan = c->plans.back();
    return AZ_OK;
}

// The history of an image shape's last level-loop search: into / out of the context's working fields.
static void hint_load(az_ctx *c, int h, int w, int nlev)
{
    if (c->hint_h == h && c->hint_w == w && c->hint_nlev == nlev) return;
    for (auto &e : c->hints)
        if (e.h == h && e.w == w && e.nlev == nlev) {
            std::memcpy(c->hint_rows, e.rows, sizeof(e.rows)); std::memcpy(c->hint_P, e.P, sizeof(e.P));
            std::memcpy(c->hint_PZ, e.PZ, sizeof(e.PZ)); std::memcpy(c->hint_U, e.U, sizeof(e.U));
            std::memcpy(c->hint_SPN, e.SPN, sizeof(e.SPN));
            std::memcpy(c->hint_old, e.old, sizeof(e.old)); c->hint_n = e.n; c->hint_full_streak = e.full_streak;
            c->hint_h = h; c->hint_w = w; c->hint_nlev = nlev;
            e.use = ++c->hint_clock;
            return;
        }
    c->hint_h = -1; c->hint_w = -1; c->hint_nlev = 0;          // no search of this shape seen (yet)
    c->hint_n = 0; c->hint_full_streak = 0;
    std::memset(c->hint_rows, 0, sizeof(c->hint_rows));
}

static void hint_store(az_ctx *c)
{
    if (c->hint_h < 0) return;
    az_ctx::ShapeHint *slot = nullptr;
    for (auto &e : c->hints) if (e.h == c->hint_h && e.w == c->hint_w && e.nlev == c->hint_nlev) slot = &e;
    if (!slot) {
        if (c->hints.size() >= 64) {
            size_t lru = 0;
            for (size_t i = 1; i < c->hints.size(); ++i) if (c->hints[i].use < c->hints[lru].use) lru = i;
            c->hints.erase(c->hints.begin() + (long)lru);
        }
        c->hints.emplace_back();
        slot = &c->hints.back();
        slot->h = c->hint_h; slot->w = c->hint_w; slot->nlev = c->hint_nlev;
    }
    std::memcpy(slot->rows, c->hint_rows, sizeof(slot->rows)); std::memcpy(slot->P, c->hint_P, sizeof(slot->P));
    std::memcpy(slot->PZ, c->hint_PZ, sizeof(slot->PZ)); std::memcpy(slot->U, c->hint_U, sizeof(slot->U));
    std::memcpy(slot->SPN, c->hint_SPN, sizeof(slot->SPN));
    std::memcpy(slot->old, c->hint_old, sizeof(slot->old)); slot->n = c->hint_n; slot->full_streak = c->hint_full_streak;
    slot->use = ++c->hint_clock;
}

// Whole-tree speculation: should this search evaluate, in ONE head pass, a shape-static superset of the rows its tree can
// need and find every level's outputs by window lookup?  Two supersets (StaticPlan::fs): the unique rois of the shape's FULL
// tree (fewest rows; right only if the tree turns out full -- a pruned tree may keep another _sift_dup survivor, err bit
// 256 -> the search is repeated level by level) and the CLOSURE over all survivor choices (~12 % more rows at 600x1000;
// right for every tree).  It pays when the tree is dense: the level-by-level forms stream the int6 weights once per pass
// and pay each pass's fixed cost (RoIPool, reduce, int7, heads, a geometry kernel), the whole-tree pass pays the rows the
// tree does not have.  The decision is by ROW COUNTS: what the shape's previous search would have cost in the
// level-by-level form the context would pick for it (pair_plan) against one pass of the superset's rows, with the pass
// costs measured on this device (pass_us).  A full-tree history takes the tree rows, anything else the closure.
// Builds what the form needs (the shape's plan, the non-deferred speculative pre-pass, the window table, the row map)
// outside any graph capture; sets c->full_now.  params.reserved bit 8: never; bit 9: whenever the shape allows (tests) --
// the tree rows, or with bit 10 the closure; AZ_FULL_SPEC=0 / 2 / 3 likewise (3 = closure whenever possible).
static int build_full_set(az_ctx *c, const az_params *p, int nlev, int variant)
{
    az_ctx::StaticPlan &k = *c->plan;
    az_ctx::StaticPlan::FullSet &f = k.fs[variant];
    const auto &sp = c->spc[0];
    join_s2(c);
    hipStream_t s = c->stream;
    auto grab = [&](void **q, size_t bytes) { return hipMalloc(q, bytes + 256) == hipSuccess; };
    auto give_up = [&]() {
        (void)hipGetLastError();
        for (void *q : {(void *)f.htab, (void *)f.spec_map, (void *)f.full_meta, (void *)f.full_urois, (void *)f.full_ubox}) if (q) hipFree(q);
        f = az_ctx::StaticPlan::FullSet();
        f.full_state = -1;
        return (int)AZ_OK;
    };
    if (sp.U > 64) return give_up();
    const int root = k.Utot - 1;                   // the plan's last row
    int base_rows = 0;                             // rows of the pass before the extra rows
    struct Tmp { float *all = nullptr; int *newrow = nullptr; ~Tmp() { if (all) hipFree(all); if (newrow) hipFree(newrow); } } tmp;
    int N = 0;
    if (variant == 1) {
        // every region any pruning can produce, level by level (no _sift_dup: whichever duplicate survives is among them)
        const int capAll = (int)AZ_TAB_ROOT_HOST - 2;
        if (!grab((void **)&tmp.all, (size_t)capAll * 5 * sizeof(float)) || !grab((void **)&tmp.newrow, (size_t)capAll * sizeof(int)))
            return give_up();
        const double rootb[4] = {0.0, 0.0, p->im_w - 1.0, p->im_h - 1.0};           // test.py:355
        HIPCHK(c, hipMemsetAsync(c->cnt, 0, sizeof(AzCounts), s));
        HIPCHK(c, hipMemcpyAsync(c->Z, rootb, sizeof(rootb), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipStreamSynchronize(s));                                          // (`rootb` lives on this frame)
        int n_cur = 1;
        for (int l = 0; l < nlev; ++l) {
            if (N + n_cur > capAll) return give_up();
            azk_closure_rois(s, c->Z, n_cur, p->scale, tmp.all + (size_t)N * 5);
            N += n_cur;
            if (l + 1 == nlev) break;
            int rc = set_count(c, &c->cnt->PZ[0], n_cur);
            if (rc) return rc;
            azk_divide(s, &c->cnt->PZ[0], &c->cnt->CH[0], &c->cnt->err, c->maxR, c->maxCh, c->Z, p->min_side, c->choff,
                       c->child, c->ckey, nullptr, nullptr, nullptr, 0, nullptr);
            HIPCHK(c, hipMemcpyAsync(c->h_cnt, c->cnt, sizeof(AzCounts), hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            const int n_next = c->h_cnt->CH[0];
            if (c->h_cnt->err || n_next > c->maxR) {
                HIPCHK(c, hipMemsetAsync(c->cnt, 0, sizeof(AzCounts), s));
                return give_up();
            }
            if (n_next == 0) break;
            HIPCHK(c, hipMemcpyAsync(c->Z, c->child, (size_t)n_next * 4 * sizeof(double), hipMemcpyDeviceToDevice, s));
            n_cur = n_next;
        }
    }
    const int cap = (variant == 1 ? N : k.Utot) + sp.U + 1;
    unsigned T = 64; while (T < 2u * (unsigned)cap) T <<= 1;
    if (cap > c->maxR || cap >= (int)AZ_TAB_ROOT_HOST ||
        !grab((void **)&f.htab, (size_t)T * 8) || !grab((void **)&f.spec_map, (size_t)sp.U * sizeof(int)) ||
        !grab((void **)&f.full_meta, 16) || !grab((void **)&f.full_urois, (size_t)cap * 5 * sizeof(float)) ||
        !grab((void **)&f.full_ubox, (size_t)cap * 4 * sizeof(double)))
        return give_up();
    f.hT = T;
    HIPCHK(c, hipMemsetAsync(f.full_meta, 0, 16, s));
    int h[4] = {0, 0, 0, 0};
    if (variant == 1) {
        azk_full_tab_build(s, tmp.all, N, 0, c->spatial_scale, f.htab, T, f.full_meta + 2);
        azk_closure_compact(s, tmp.all, N, c->spatial_scale, f.htab, T, tmp.newrow, f.full_urois, f.full_ubox, f.full_meta + 3,
                            f.full_meta + 2);
        HIPCHK(c, hipMemcpyAsync(h, f.full_meta, 16, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (h[2]) return give_up();
        base_rows = h[3];
    } else {
        HIPCHK(c, hipMemcpyAsync(f.full_urois, k.urois, (size_t)root * 5 * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync(f.full_ubox, k.ubox, (size_t)root * 4 * sizeof(double), hipMemcpyDeviceToDevice, s));
        azk_full_tab_build(s, k.urois, k.Utot, root, c->spatial_scale, f.htab, T, f.full_meta + 2);
        base_rows = root;
    }
    // every row of the speculative layout (levels 1-3) -> its row in this pass; windows the rows above lack become extra rows
    azk_full_map(s, c->spec_urois[0], sp.U, c->spatial_scale, f.htab, T, base_rows, cap, f.full_urois, f.full_ubox, f.spec_map,
                 f.full_meta + 1, f.full_meta + 2);
    HIPCHK(c, hipMemcpyAsync(h, f.full_meta, 16, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (h[2] || (variant == 1 && h[1] != 0)) return give_up();       // (the closure holds every speculative row by construction)
    f.Ufull = base_rows + h[1] + 1;
    // the root: the pass's last row (RoIPool treats the tail of a launch cooperatively)
    HIPCHK(c, hipMemcpyAsync(f.full_urois + (size_t)(f.Ufull - 1) * 5, k.urois + (size_t)root * 5, 5 * sizeof(float),
                             hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(f.full_ubox + (size_t)(f.Ufull - 1) * 4, k.ubox + (size_t)root * 4, 4 * sizeof(double),
                             hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(f.full_meta, &f.Ufull, sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipStreamSynchronize(s));
    f.full_state = 1;
    if (getenv("AZ_FULL_DEBUG")) fprintf(stderr, "az: whole-tree rows for (%dx%d), %s: %d (full tree %d, closure regions %d)\n",
                                         p->im_h, p->im_w, variant ? "closure" : "tree", f.Ufull, k.Utot, N);
    return AZ_OK;
}

// What the level-by-level form the context would pick for this shape (pair_plan on the same history) costs for ONE of the
// shape's recorded trees, in us.
static double level_forms_cost(az_ctx *c, const HintView &v, int nlev, int n_spec, int specU, int pair_mask)
{
    double t = pass_us(c, specU) + PASS_OVERHEAD_US;
    for (int l = n_spec; l < nlev; ++l) {
        if (v.U[l] <= 0) {              // the tree had ended: the level's pass is enqueued all the same and finds no rows
            t += EMPTY_LEVEL_US;
            if ((pair_mask >> l) & 1) ++l;
            continue;
        }
        if ((pair_mask >> l) & 1) {
            t += pass_us(c, v.U[l] + pair_rows(v, l)) + PASS_OVERHEAD_US + LOOKUP_US;
            ++l;
        } else
            t += pass_us(c, v.U[l]) + PASS_OVERHEAD_US;
    }
    return t;
}

static int full_prepare(az_ctx *c, const az_params *p, int nlev, bool tune)
{
    c->full_now = 0;
    if (tune || (p->reserved & (1 | 2 | 16 | 256)) || !p->fixed_num) return AZ_OK;
    if (c->full_env < 0) { const char *e = getenv("AZ_FULL_SPEC"); c->full_env = e ? atoi(e) : 1; }
    const bool forced = (p->reserved & 512) || c->full_env >= 2;
    if (!forced && c->full_env == 0) return AZ_OK;
    const SearchPlan q0 = plan_search(c, p, nlev, tune);        // (full_now is 0: the other form's plan)
    if (!(q0.fused && q0.fused_lv && q0.n_spec == 3 && q0.lv_limit >= q0.n_spec && nlev > q0.n_spec)) return AZ_OK;
    const bool have_hist = c->hint_h == p->im_h && c->hint_w == p->im_w && c->hint_nlev == nlev && c->hint_n > 0;
    // the last TWO searches of this shape walked the FULL tree (every region zoomed at every level but the last)?  One full
    // tree in a stream of different images says little about the next, and a tree-rows pass that misses a window costs a
    // second search; a context that keeps seeing full trees (Tz <= 0, or a threshold every region passes) gets there at its
    // third search.
    const bool full_hist = have_hist && c->hint_full_streak >= 2;
    if (!forced && !have_hist) return AZ_OK;
    int variant = forced ? (((p->reserved & 1024) || c->full_env == 3) ? 1 : 0) : (full_hist ? 0 : 1);
    int rc;
    if ((rc = ensure_static_plan(c, p, nlev)) != AZ_OK) return rc;
    if (!static_plan_matches(c, p, nlev)) return AZ_OK;
    az_ctx::StaticPlan &k = *c->plan;
    if (k.fs[variant].full_state < 0) return AZ_OK;
    double now = 0.0;
    if (!forced) {
        // cheapest the superset can be: the full tree's rows.  Not worth building anything if even that loses.
        // (expected over the shape's recorded trees)
        const int specU = c->spc[q0.defer_root ? 1 : 0].h == p->im_h ? c->spc[q0.defer_root ? 1 : 0].U : 48;
        // (the tree-rows pass presumes the tree is full again: priced against the full trees of the streak)
        const int nrec = variant == 0 ? (c->hint_full_streak < c->hint_n ? c->hint_full_streak : c->hint_n) : c->hint_n;
        for (int r = 0; r < nrec; ++r) now += level_forms_cost(c, hint_rec(c, r), nlev, q0.n_spec, specU, q0.pair_mask);
        now /= nrec;
        const double best = pass_us(c, k.Utot) + PASS_OVERHEAD_US + LOOKUP_US * (nlev - q0.n_spec);
        if (!(best + 10.0 < now)) return AZ_OK;
    }
    // the non-deferred layout of the speculative rows (the root is row 0 there; here it maps to the pass's last row)
    SearchPlan q1 = q0; q1.defer_root = false;
    if ((rc = ensure_spec_cache(c, p, q1)) != AZ_OK) return rc;
    const auto &sp = c->spc[0];
    if (!(sp.h == p->im_h && sp.w == p->im_w && sp.scale == p->scale && sp.min_side == p->min_side)) return AZ_OK;
    if (k.fs[variant].full_state == 0 && (rc = build_full_set(c, p, nlev, variant)) != AZ_OK) return rc;
    if (k.fs[variant].full_state != 1) return AZ_OK;
    if (!forced) {
        const double full = pass_us(c, k.fs[variant].Ufull) + PASS_OVERHEAD_US + LOOKUP_US * (nlev - q0.n_spec);
        if (!(full + 10.0 < now)) return AZ_OK;
    }
    c->full_now = variant + 1;
    if (getenv("AZ_FULL_DEBUG")) fprintf(stderr, "az: whole-tree pass on (%dx%d): %d rows (%s; plan %d)\n", p->im_h, p->im_w,
                                         k.fs[variant].Ufull, variant ? "closure" : "tree rows", k.Utot);
    return AZ_OK;
}

static int enqueue_static(az_ctx *c, const az_params *p, int nlev, int k)
{
    const auto &q = *c->plan;
    launch_head(c, q.meta, -1, p->im_h, p->im_w, p->eps, c->zoom_u, c->score_u, c->delta_u, p->min_side, true,
                q.coop, q.urois, q.ubox, q.Utot, true);
    { Timed t(c, "static_candidates", nlev - 1);
      AzStaticArgs a;
      a.cnt = c->cnt; a.reg_u = q.reg_u; a.cand_src = q.cand_src; a.key_u = c->key_u; a.pred_u = c->pred_u;
      a.score_u = c->score_u;
      a.zoom_u = c->zoom_u; a.Yall = c->Yall; a.Sall = c->Sall; a.Tz = p->Tz;
      a.nlev = nlev; a.Utot = q.Utot; a.capCand = c->maxCand;
      for (int l = 0; l <= nlev; ++l) a.roff[l] = q.roff[l];
      for (int l = 0; l < nlev; ++l) { a.U[l] = q.U[l]; a.CH[l] = q.CH[l]; }
      a.k = k; a.Yout = (double *)((unsigned char *)c->cnt + RES_HDR);
      a.Sout = (float *)((unsigned char *)c->cnt + RES_HDR + (size_t)k * 32);
      // fixed proposal count: the same launch ranks the candidates and writes the top k (params.reserved bit 3
      // keeps the separate selection kernels, for tests)
      if (p->fixed_num && !(p->reserved & 8) && azk_static_select(geom_stream(c), a)) return AZ_OK;
      azk_static_candidates(geom_stream(c), a); }
    enqueue_select(c, p, nlev, k);
    return AZ_OK;
}

// In the level loop only the device knows a level's row count.  If the previous search on this context forwarded many
// rois at level l, the next one probably does too: its int6 is then sent to both GEMM kernels (rows_hint -1, see
// launch_head).  A wrong guess costs an idle launch, never a result.
static int many_rows_expected(const az_ctx *c, int l)
{
    // (hint rows: rows of the PASS at that level, speculative rows included; the mean over the shape's recorded searches)
    if (l < 0 || l >= AZ_MAX_LEVELS || c->gemm12_min_rows == 0x7fffffff || c->hint_n <= 0) return 0;
    long sum = 0;
    for (int r = 0; r < c->hint_n; ++r) sum += hint_rec(c, r).rows[l];
    return sum >= (long)c->gemm12_dual_rows * c->hint_n ? -1 : 0;
}

// --------------------------------------------------------------------------------------
// Everything az_propose enqueues on the ctx stream (no host synchronisation, no host-dependent sizes:
// every count is read on the device), so the same sequence can also be captured into a hipGraph.
static int enqueue_search(az_ctx *c, const az_params *p, int K, int nlev, int k, bool tune)
{
    hipStream_t s = c->stream;

    // Speculative evaluation of levels 1-3.  The root is always divided (test.py:383-384), so
    // level 2's regions are known up front, and level 3's regions are a subset of the children
    // of ALL level-2 regions.  These few dozen rows cost one pass over the 411 MB int6 weights
    // instead of three (each of those levels is weight-streaming-bound).  Head outputs are a
    // fixed function of the roi, so the levels below just look their rows up: bit-identical
    // results.  (params.reserved bit 0 turns this off.)
    const SearchPlan plan = plan_search(c, p, nlev, tune);
    const int n_spec = plan.n_spec;
    const bool fused = plan.fused, fused_lv = plan.fused_lv, defer_root = plan.defer_root;
    if (tune && !c->hisB) {
        c->capHis = 2 * c->maxR;
        HIPCHK(c, hipMalloc((void **)&c->hisB, (size_t)c->capHis * 4 * sizeof(double)));
        HIPCHK(c, hipMalloc((void **)&c->hisZ, (size_t)c->capHis * sizeof(float)));
    }
    if (!fused) azk_init_root(s, c->cnt, c->B[0], p->im_h, p->im_w);       // also zeroes the counters
    if (fused) {
        // (the pre-pass -- B1, all children of B1, the rois of the speculative rows -- depends on the image shape
        //  only: az_propose_launch ran it for this shape, k_spec_levels restores its counters)
    } else if (n_spec) {
        Timed t(c, "spec_geometry", -1);
        // children of the root -> B1 (with _sift_dup), exactly what level 1's divide will produce
        azk_divide(s, &c->cnt->P[0], &c->cnt->scratch[3], &c->cnt->err, c->maxR, c->maxCh, c->B[0], p->min_side,
                   c->choff, c->child, c->ckey, nullptr, nullptr, nullptr, 0, nullptr);
        azk_dedup_regions(s, c->ckey, &c->cnt->scratch[3], c->maxCh, c->maxR, c->first, c->child, c->B[1],
                          &c->cnt->specP1, &c->cnt->err, nullptr, nullptr);
        // children of ALL of B1, before _sift_dup; their offsets identify (parent, child) later
        azk_divide(s, &c->cnt->specP1, &c->cnt->specCH, &c->cnt->err, c->maxR, c->maxCh, c->B[1], p->min_side,
                   c->choff_all, c->child, c->ckey, nullptr, nullptr, nullptr, 0, nullptr);
        azk_spec_rois(s, c->B[0], c->B[1], c->child, c->cnt, c->maxR, p->scale, c->urois);
    }
    const bool full = plan.full != 0;
    const az_ctx::StaticPlan::FullSet *fp = full ? &c->plan->fs[plan.full - 1] : nullptr;
    // the head outputs of the speculative / whole-tree pass (three stages: two sets used in turn, az_ctx.h)
    float *zs = c->zoom_s, *ss = c->score_s, *ds = c->delta_s;
    if (full && c->three_now) {
        c->out_par ^= 1;
        if (c->out_par) { zs = c->zoom_s2; ss = c->score_s2; ds = c->delta_s2; }
    }
    // inv_index of level l (two buffers by level parity: k_level_geom's candidate-copy workgroup reads level l's while
    // its chain workgroup writes level l+1's)
    auto INV = [&](int l) { return (l & 1) ? c->inv_odd : c->inv; };
    // (whole-tree speculation: the *_v sets alternate by level -- a level's geometry kernel reads its own set while it
    //  writes the next level's)
    auto Vp = [&](int l) { return (full && (l & 1)) ? c->pred_w : c->pred_v; };
    auto Vs = [&](int l) { return (full && (l & 1)) ? c->score_w : c->score_v; };
    auto Vz = [&](int l) { return (full && (l & 1)) ? c->zoom_w : c->zoom_v; };
    auto Vk = [&](int l) { return (full && (l & 1)) ? c->keep_w : c->keep_v; };
    auto Vy = [&](int l) { return (full && (l & 1)) ? c->key_w : c->key_v; };
    if (full)
        // the search's ONE head pass: the unique rois of the image shape's full tree (+ the speculative rows the plan
        // lacks), the root last; outputs by row in zoom_s / score_s / delta_s
    {
        launch_head(c, fp->full_meta, -1, p->im_h, p->im_w, p->eps, zs, ss, ds, 0.0, false, 1,
                    fp->full_urois, fp->full_ubox, fp->Ufull);
        s = geom_stream(c);            // (two stages: everything behind the one head pass goes where its int7 went)
    }
    else if (fused && plan.cut == 2 && !defer_root)
        // early end before the third level: the first 1 + P1 rows of S = [root ; B1 ; children of all of B1]
        launch_head(c, c->spec_U[0] + 1, -1, p->im_h, p->im_w, p->eps, zs, ss, ds, 0.0, false,
                    0, c->spec_urois[0], nullptr, 1 + c->spc[0].P1);
    else if (fused)
        launch_head(c, c->spec_U[defer_root ? 1 : 0], -1, p->im_h, p->im_w, p->eps, zs, ss, ds, 0.0, false,
                    0, c->spec_urois[defer_root ? 1 : 0], nullptr, c->spc[defer_root ? 1 : 0].U);
    else if (n_spec)
        launch_head(c, &c->cnt->specU, -1, p->im_h, p->im_w, p->eps, zs, ss, ds);
    if (fused) {
        Timed t(c, "spec_levels", 0);
        AzFusedArgs a;
        a.cnt = c->cnt;
        a.B[0] = c->B[0]; a.B[1] = c->B[1]; a.srcB[0] = c->srcB[0]; a.srcB[1] = c->srcB[1];
        a.index = c->index; a.inv = INV(n_spec); a.zr = c->zr; a.choff = c->choff; a.csrc = c->csrc;
        const int dslot = defer_root ? 1 : 0;
        a.choff_all = c->spec_choff[dslot]; a.specB1 = c->specB1[dslot];
        a.reset = 1; a.specP1 = c->spc[dslot].P1; a.specCH = c->spc[dslot].CH; a.specU = c->spc[dslot].U;
        a.ubox = c->ubox; a.pred_u = c->pred_u; a.Yall = c->Yall; a.Z = c->Z; a.child = c->child;
        a.zoom_u = c->zoom_u; a.score_u = c->score_u; a.delta_u = c->delta_u; a.Sall = c->Sall;
        a.zoom_s = zs; a.score_s = ss; a.delta_s = ds;
        a.scale = p->scale; a.Tz = p->Tz; a.min_side = p->min_side; a.eps = p->eps; a.dedup = (float)p->dedup;
        a.batch = p->batch_size; a.im_h = p->im_h; a.im_w = p->im_w; a.nlev = nlev; a.n_fused = n_spec;
        a.capR = c->maxR; a.capCh = c->maxCh; a.capCand = c->maxCand;
        a.rois = c->rois; a.urois = c->urois; a.next_dedup = fused_lv ? 1 : 0; a.defer_root = defer_root ? 1 : 0;
        a.cut_next = (plan.cut && plan.cut <= n_spec) ? 1 : 0;
        a.cut_short = (plan.cut == 2 && !defer_root) ? 1 : 0;
        a.spec_next = (plan.pair_mask >> n_spec) & 1; a.choff_next = c->choff_pair; a.crow = c->crow;
        a.spatial_scale = c->spatial_scale;
        a.row_map = full ? fp->spec_map : nullptr; a.root_row = full ? fp->Ufull - 1 : 0;
        a.stab = full ? fp->htab : nullptr; a.stabT = full ? fp->hT : 0;
        a.pred_v = Vp(n_spec); a.score_v = Vs(n_spec); a.zoom_v = Vz(n_spec); a.keep_v = Vk(n_spec); a.key_v = Vy(n_spec);
        azk_spec_levels(s, a);
    }
    bool have_v = full;               // this level's head outputs were looked up among the previous pass's rows (*_v arrays)
    const int nlev_run = plan.cut ? plan.cut : nlev;     // (early end: the levels from plan.cut on are not enqueued)
    for (int l = fused ? n_spec : 0; l < nlev_run; ++l) {
        const int cur = l & 1;
        const int *Pptr = &c->cnt->P[l];
        int *Uptr = &c->cnt->U[l];
        // (the last level's copy + top-k stay chip-wide; from plan.lv_limit on the levels outgrow the fused kernel)
        const bool lv_here = fused_lv && l + 1 < nlev && l < plan.lv_limit;
        const bool pair_here = !full && fused_lv && ((plan.pair_mask >> l) & 1) && !have_v;   // this pass carries level l+1's rows
        if (lv_here) {
            // this level's rois were projected and deduplicated by the previous geometry kernel, which also left the
            // pass's row count (its unique rois + pair-speculation rows + the deferred root's) in cnt->PR[l]
            if (!have_v)
                launch_head(c, &c->cnt->PR[l], l, p->im_h, p->im_w, p->eps, c->zoom_u, c->score_u,
                            c->delta_u, p->min_side, true, (defer_root && l == n_spec) ? 1 : 0, nullptr, nullptr,
                            many_rows_expected(c, l));
            Timed t(c, "level_geom", l);
            AzLevelArgs a;
            a.cnt = c->cnt; a.level = l; a.nlev = nlev;
            a.cut_next = (plan.cut == l + 1) ? 1 : 0;
            a.B = c->B[cur]; a.Bnext = c->B[cur ^ 1];
            a.pred_u = have_v ? Vp(l) : c->pred_u; a.score_u = have_v ? Vs(l) : c->score_u;
            a.zoom_u = have_v ? Vz(l) : c->zoom_u; a.keep_u = have_v ? Vk(l) : c->keep_u; a.Uptr = Uptr;
            a.urois = c->urois; a.index = c->index; a.inv = INV(l); a.inv_next = INV(l + 1); a.ubox = c->ubox;
            a.Yall = c->Yall; a.Sall = c->Sall;
            a.scale = p->scale; a.Tz = p->Tz; a.min_side = p->min_side; a.dedup = (float)p->dedup;
            a.batch = p->batch_size; a.capR = c->maxR; a.capCh = c->maxCh; a.capCand = c->maxCand;
            a.force_root = 1; a.root_row = (defer_root && l == n_spec && !have_v) ? 1 : 0;
            a.lookup_next = full ? 2 : (pair_here ? 1 : 0);
            a.spec_next = (!full && !pair_here && ((plan.pair_mask >> (l + 1)) & 1)) ? 1 : 0;
            a.delta_u = full ? ds : c->delta_u; a.choff_all = c->choff_pair; a.choff_next = c->choff_pair; a.crow = c->crow;
            a.stab = full ? fp->htab : nullptr; a.stabT = full ? fp->hT : 0; a.root_row_full = full ? fp->Ufull - 1 : 0;
            a.score_all = ss; a.zoom_all = zs;
            a.pred_v = Vp(l + 1); a.score_v = Vs(l + 1); a.zoom_v = Vz(l + 1); a.keep_v = Vk(l + 1); a.key_v = Vy(l + 1);
            a.im_h = p->im_h; a.im_w = p->im_w; a.eps = p->eps; a.spatial_scale = c->spatial_scale;
            azk_level_geom(s, a);
            have_v = full || pair_here;
            continue;
        }
        if (!fused_lv || l > plan.lv_limit) {   // (otherwise the fused predecessor -- spec_levels or level_geom -- has done this already)
          Timed t(c, "rois_dedup", l);
          azk_rois_dedup(s, c->B[cur], Pptr, c->maxR, p->scale, (float)p->dedup, p->batch_size, c->rois, c->key,
                         c->grp, c->first, c->index, INV(l), c->urois, c->ubox, Uptr); }
        // The last level of a default search with a fixed proposal count: its candidates, its counters and the final
        // top-k come from ONE launch (az_static.hip: k_final_select) instead of k_flags, k_compact, k_rank_count and
        // k_rank_scatter; the tail kernel emits the selection keys.  (params.reserved bits 1 / 3 keep the separate
        // kernels: same bits.)
        const bool final_fused = fused && !tune && l + 1 == nlev && l >= n_spec && p->fixed_num && !(p->reserved & 8) &&
                                 c->final_env;
        if (full && !have_v && l >= n_spec) {
            // whole-tree speculation, a level on the multi-launch kernels: its outputs by window lookup, chip-wide
            Timed t(c, "full_lookup", l);
            azk_full_lookup(s, Uptr, c->urois, c->ubox, fp->htab, fp->hT, fp->Ufull - 1, c->spatial_scale, ds, ss,
                            zs, p->im_h, p->im_w, p->eps, p->min_side, Vp(l), Vs(l), Vz(l), Vk(l), Vy(l), &c->cnt->err);
            have_v = true;
        }
        if (l < n_spec) {
            Timed t(c, "spec_lookup", l);
            azk_spec_lookup(s, l, Uptr, c->index, c->srcB[cur], c->ubox, zs, ss, ds,
                            p->im_h, p->im_w, p->eps, c->zoom_u, c->score_u, c->delta_u, c->pred_u);
        } else if (!have_v) {
            launch_head(c, (fused_lv && l <= plan.lv_limit) ? &c->cnt->PR[l] : Uptr, l, p->im_h, p->im_w, p->eps, c->zoom_u, c->score_u, c->delta_u,
                        p->min_side, final_fused, 0, nullptr, nullptr, many_rows_expected(c, l), final_fused);
        }
        if (final_fused) {
            Timed t(c, "final_select", l);
            AzFinalArgs a;
            a.cnt = c->cnt; a.level = l; a.inv = INV(l); a.key_u = have_v ? Vy(l) : c->key_u;
            a.pred_u = have_v ? Vp(l) : c->pred_u;
            a.score_u = have_v ? Vs(l) : c->score_u; a.zoom_u = have_v ? Vz(l) : c->zoom_u;
            a.Yall = c->Yall; a.Sall = c->Sall; a.Tz = p->Tz;
            a.force_root = (l == 0) ? 1 : 0; a.capCand = c->maxCand; a.k = k;
            a.Yout = (double *)((unsigned char *)c->cnt + RES_HDR);
            a.Sout = (float *)((unsigned char *)c->cnt + RES_HDR + (size_t)k * 32);
            azk_final_select(s, a);
            return AZ_OK;
        }
        if (tune) {
            Timed t(c, "record_anchors", l);
            azk_record_anchors(s, c->cnt, l, c->maxR, c->capHis, c->B[cur], INV(l), c->zoom_u, c->hisB, c->hisZ,
                               &c->cnt->nhis, &c->cnt->err);
        }
        { Timed t(c, "flags_compact", l);
          azk_flags_compact(s, c->cnt, l, c->maxR, c->maxCand, c->B[cur], INV(l), have_v ? Vp(l) : c->pred_u,
                            have_v ? Vs(l) : c->score_u,
                            have_v ? Vz(l) : c->zoom_u, (tune && l == 0) ? 0.0 : p->Tz, p->min_side, l == 0 && !tune, c->cflag,
                            c->zflag, c->bc_c, c->bc_z, c->Yall, c->Sall, c->Z, c->zr); }
        if (l + 1 < nlev) {      // the reference also divides after the last level but never uses it
            const bool track = (n_spec && l == 1);       // level-3 regions remember their speculative row
            { Timed t(c, "divide", l);
              azk_divide(s, &c->cnt->PZ[l], &c->cnt->CH[l], &c->cnt->err, c->maxR, c->maxCh, c->Z, p->min_side,
                         c->choff, c->child, c->ckey, track ? c->choff_all : nullptr, c->zr, &c->cnt->specP1, 1,
                         track ? c->csrc : nullptr); }
            { Timed t(c, "sift_dup", l);
              azk_dedup_regions(s, c->ckey, &c->cnt->CH[l], c->maxCh, c->maxR, c->first, c->child,
                                c->B[cur ^ 1], &c->cnt->P[l + 1], &c->cnt->err, track ? c->csrc : nullptr,
                                c->srcB[cur ^ 1]); }
        }
        have_v = false;           // (a level on the multi-launch kernels never looks the next one's outputs up)
    }
    enqueue_select(c, p, nlev_run, k);
    if (tune && c->pool) {
        Timed t(c, "pool_append", nlev);
        azk_pool_append(s, c->hisZ, &c->cnt->nhis, c->capHis, c->pool, c->pool_n, c->pool_cap);
    }
    return AZ_OK;
}


static int launch_impl_body(az_ctx *c, const az_params *p);

// One search enqueued on THIS context's stream (the public az_propose_launch picks the lane first).
int launch_impl(az_ctx *c, const az_params *p)
{
    const int rc = launch_impl_body(c, p);
    // a batch slot on the owner's spare head set: the next slot to take the set waits for what this one enqueued (whatever
    // became of the launch -- a failed one may have enqueued its first kernels)
    if (c && c->head_shared && c->owner && c->owner->spare.ev) {
        if (hipEventRecord(c->owner->spare.ev, c->stream) == hipSuccess) c->owner->spare.ev_live = true;
        else { (void)hipGetLastError(); c->async_err = 1; }
    }
    return rc;
}

static int launch_impl_body(az_ctx *c, const az_params *p)
{
    int rc = check_ready(c, true, false);          // (whether `stream` waits for the second stream is decided below)
    if (rc) return rc;
    if (!p || p->im_h <= 0 || p->im_w <= 0 || !(p->scale > 0) || p->batch_size <= 0 || !(p->min_side > 0))
        return fail(c, AZ_ERR_INVALID, "az_propose: bad parameters");
    const int K = num_levels(p->im_h, p->im_w, p->min_side);
    // The tuner's variant of the search (lib/detect/tune.py:256-316, params.reserved bit 2) runs
    // `for k in xrange(K)` -- one level more than test.py:373 --, applies Tz from the second level
    // on (the first compares against 0), never forces the root, and keeps the anchor history Bhis.
    const bool tune = (p->reserved & 4) != 0;
    const int nlev = tune ? K : K - 1;
    if (nlev < 1)
        return fail(c, AZ_ERR_INVALID,
                    "az_propose: image too small for one search level (the reference's loop at "
                    "lib/detect/test.py:373 would not execute)");
    if (nlev > AZ_MAX_LEVELS) return fail(c, AZ_ERR_CAPACITY, "az_propose: too many levels");
    int k = p->num_proposals;
    if (p->fixed_num) {
        if (k <= 0) return fail(c, AZ_ERR_INVALID, "az_propose: num_proposals must be positive");
        if (k > AZ_TOPK_MAX) return fail(c, AZ_ERR_CAPACITY, "az_propose: num_proposals > 4096");
    }
    if ((int)c->pend.size() >= az_ctx::AZ_QUEUE_MAX)
        return fail(c, AZ_ERR_STATE, "az_propose_launch: three searches are already queued on this lane, fetch one first");
    if (!c->head_bufs && (rc = ensure_lane_head(c)) != AZ_OK) return rc;   // (a batch slot searching on its own for the first time)
    if (c->head_shared && c->owner && c->owner->spare.ev_live)             // (... behind the slot that had the spare set before it)
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->owner->spare.ev, 0));
    if (!c->pend.empty() && !(p->fixed_num && c->pend.back().copied))
        return fail(c, AZ_ERR_STATE, "az_propose_launch: queueing a search behind another needs a fixed proposal count for both");
    HIPCHK(c, hipSetDevice(c->device));
    if (!(c->profiling & 4)) clear_events(c);
    c->cand_n = -1;
    if (c->cal.state == 0 && (rc = calibrate_passes(c)) != AZ_OK) return rc;
    hint_load(c, p->im_h, p->im_w, nlev);          // what this shape's last search looked like (decides the form below)
    bool stat = static_wanted(c, p, tune);
    if (stat) {
        if ((rc = ensure_static_plan(c, p, nlev)) != AZ_OK) return rc;
        stat = static_plan_matches(c, p, nlev);          // (a tree that outgrows the plan buffers: level loop)
    }
    c->last_static = stat ? 1 : 0;
    c->full_now = 0;
    if (!stat && (rc = full_prepare(c, p, nlev, tune)) != AZ_OK) return rc;
    c->last_full = !stat ? plan_search(c, p, nlev, tune).full : 0;
    if (!stat && (rc = ensure_spec_cache(c, p, plan_search(c, p, nlev, tune))) != AZ_OK) return rc;
    c->last_defer = (!stat && plan_search(c, p, nlev, tune).defer_root) ? 1 : 0;
    c->last_pair_mask = stat ? 0 : plan_search(c, p, nlev, tune).pair_mask;
    c->last_cut = stat ? 0 : plan_search(c, p, nlev, tune).cut;
    hipStream_t s = c->stream;
    if (c->use_graphs < 0) { const char *e = getenv("AZ_GRAPH"); c->use_graphs = (e && atoi(e)) ? 1 : 0; }
    // Stages on streams of their own (az_ctx.h): a search of ONE head pass whose rows do not depend on its own geometry, on a
    // context that runs its searches on ONE lane -- measured (round 5, 600x1000 at Tz = 0): one lane 1.20 -> 1.15-1.17 ms per
    // image (the next image's RoIPool + int6 no longer wait for this one's heads and three single-workgroup geometry kernels);
    // with two lanes the lanes already give that overlap and the split only makes the steps burstier (1.119 -> 1.122 ms), so
    // it is left off there.  AZ_TWO_STAGE=0: never; 2: on two lanes as well; 4: two stages, never three (measurements).
    if (c->split_env < 0) { const char *e = getenv("AZ_TWO_STAGE"); c->split_env = e ? atoi(e) : 1; }
    const bool one_lane = !c->owner && c->lanes == 1;
    c->split_now = (c->split_env && (one_lane || c->split_env == 2) && (stat || c->last_full) && p->fixed_num && !c->use_graphs &&
                    !tune && !Timed::trace() && !c->head_shared) ? 1 : 0;
    if (c->split_now && !c->stream2) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, lo) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_h6, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_i7, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_s2, hipEventDisableTiming) != hipSuccess ||
            hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, lo) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_s3, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_geo[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_geo[1], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            c->split_now = 0; c->split_env = 0;          // (this device / runtime does not give the extra streams: one stage)
        }
    }
    // three stages for the whole-tree / closure form (AZ_TWO_STAGE=4: two stages there as well: measurements)
    c->three_now = (c->split_now && !stat && c->last_full && c->split_env != 4) ? 1 : 0;
    // (a two-stage search's second stage works in the buffers a three-stage search's geometry may still be using)
    if (c->split_now && !c->three_now && c->s3_live && hipStreamWaitEvent(c->stream2, c->ev_s3, 0) != hipSuccess) c->async_err = 1;
    if (!c->split_now) join_s2(c);                       // every kernel of this search goes to `stream`, into the per-search buffers
    c->gs = nullptr; c->ts = nullptr; c->async_err = 0;
    auto enqueue = [&]() { c->npass = 0; prep_scale(c); return stat ? enqueue_static(c, p, nlev, k) : enqueue_search(c, p, K, nlev, k, tune); };
    // az_set_graphs / AZ_GRAPH=1: capture the launch sequence once per (parameters, feature map) and replay it
    // as a hipGraph.  Every size is read on the device, so the sequence never changes for given parameters.
    if (c->use_graphs && !c->profiling && !(tune && c->pool)) {
        // key = the fields themselves (never the struct's bytes: padding is the caller's garbage)
        std::string key;
        auto put = [&key](const void *v, size_t n) { key.append((const char *)v, n); };
        put(&p->im_h, sizeof p->im_h); put(&p->im_w, sizeof p->im_w); put(&p->scale, sizeof p->scale);
        put(&p->Tz, sizeof p->Tz); put(&p->Tc, sizeof p->Tc); put(&p->dedup, sizeof p->dedup);
        put(&p->eps, sizeof p->eps); put(&p->min_side, sizeof p->min_side); put(&p->batch_size, sizeof p->batch_size);
        put(&p->num_proposals, sizeof p->num_proposals); put(&p->fixed_num, sizeof p->fixed_num);
        put(&p->reserved, sizeof p->reserved);
        const void *fp = c->feat;
        key.append((const char *)&fp, sizeof(fp));
        key.append((const char *)&c->d, sizeof(c->d));
        key.append((const char *)&c->nofuse_h, sizeof(int));
        key.append((const char *)&c->nofuse_w, sizeof(int));
        key.append((const char *)&c->nofuse_lv_h, sizeof(int));
        key.append((const char *)&c->nofuse_lv_w, sizeof(int));
        { const int lim = plan_search(c, p, nlev, tune).lv_limit; key.append((const char *)&lim, sizeof(int)); }
        key.append((const char *)&c->last_static, sizeof(int));
        key.append((const char *)&c->last_pair_mask, sizeof(int));
        key.append((const char *)&c->last_defer, sizeof(int));
        key.append((const char *)&c->last_full, sizeof(int));
        key.append((const char *)&c->last_cut, sizeof(int));
        for (int l = 0; l < nlev; ++l) { const int mr = many_rows_expected(c, l); key.append((const char *)&mr, sizeof(int)); }
        const void *pp = (stat || c->last_full) ? (const void *)c->plan : nullptr;
        key.append((const char *)&pp, sizeof(pp));
        auto it = c->graphs.find(key);
        if (it == c->graphs.end()) {
            // (the first search of a shape also runs once un-captured: one-time attribute calls happen there)
            if ((rc = enqueue()) != AZ_OK) return rc;
            HIPCHK(c, hipStreamSynchronize(s));
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            HIPCHK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            rc = enqueue();
            // (whatever enqueue() returned, the capture ends here: the stream must never be left capturing)
            const hipError_t ec = hipStreamEndCapture(s, &g);
            if (rc || ec != hipSuccess) {
                if (g) hipGraphDestroy(g);
                (void)hipGetLastError();
                return rc ? rc : fail(c, AZ_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ec));
            }
            const hipError_t ei = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            hipGraphDestroy(g);
            if (ei != hipSuccess) return fail(c, AZ_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei));
            az_ctx::GraphEntry ent;
            ent.exec = ge; ent.npass = c->npass;
            std::memcpy(ent.pass_src, c->pass_src, sizeof(ent.pass_src));
            std::memcpy(ent.pass_lv, c->pass_lv, sizeof(ent.pass_lv));
            it = c->graphs.emplace(key, ent).first;
        }
        c->npass = it->second.npass;
        std::memcpy(c->pass_src, it->second.pass_src, sizeof(c->pass_src));
        std::memcpy(c->pass_lv, it->second.pass_lv, sizeof(c->pass_lv));
        HIPCHK(c, hipGraphLaunch(it->second.exec, s));
    } else {
        if ((rc = enqueue()) != AZ_OK) return rc;
    }
    HIPCHK(c, hipGetLastError());
    if (c->async_err) { c->gs = nullptr; c->ts = nullptr; c->split_now = 0; c->three_now = 0; return fail(c, AZ_ERR_HIP, "az_propose: a stream / event call of the two-stage search failed"); }
    s = geom_stream(c);                                  // where the search ends: its result copy follows there
    az_ctx::PendingSearch q;
    q.p = *p; q.nlev = nlev; q.is_static = c->last_static; q.defer = c->last_defer; q.pair_mask = c->last_pair_mask;
    q.full = c->last_full;
    q.cut = c->last_cut;
    q.npass = c->npass;
    q.feat = c->feat; q.fH = c->d.H; q.fW = c->d.W; q.feat_gen = c->feat_gen;
    q.feat_is_copy = c->feat && (c->feat == c->feat_owned[0] || c->feat == c->feat_owned[1] || c->feat == c->feat_owned[2]);
    std::memcpy(q.pass_src, c->pass_src, sizeof(q.pass_src));
    std::memcpy(q.pass_lv, c->pass_lv, sizeof(q.pass_lv));
    for (q.slot = 0; q.slot < 3 && c->slot_busy[q.slot]; ++q.slot) { }
    if (q.slot >= 3) return fail(c, AZ_ERR_STATE, "az_propose_launch: no free result slot");
    if (p->fixed_num) {
        // the result block follows the search's kernels in stream order: whatever is enqueued next (the next image's
        // search, a unit call) finds it already on its way to the host
        HIPCHK(c, hipMemcpyAsync(c->h_res[q.slot], c->cnt, RES_HDR + (size_t)k * 36, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipEventRecord(c->ev_res[q.slot], s));
        q.copied = true;
    }
    q.last_s = s;
    c->last_s = s;
    if (c->gs) { HIPCHK(c, hipEventRecord(c->ev_s2, c->stream2)); c->s2_live = true; }
    if (c->gs && c->gs == c->stream3) {
        HIPCHK(c, hipEventRecord(c->ev_s3, c->stream3));
        HIPCHK(c, hipEventRecord(c->ev_geo[c->out_par], c->stream3));
        c->s3_live = true; c->g_live[c->out_par] = true;
    }
    c->gs = nullptr; c->ts = nullptr; c->split_now = 0; c->three_now = 0;
    c->slot_busy[q.slot] = true;
    c->pend.push_back(q);
    return AZ_OK;
}


// Collect the result of the search at position `idx` of the pending queue (0 = the oldest; a fallback rerun sits at
// the back) and remove it from the queue.
int fetch_entry(az_ctx *c, size_t idx, double *boxes_out, float *scores_out, int cap, int *n_out, az_stats *st)
{
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const az_ctx::PendingSearch q = c->pend[idx];
    const int nlev = q.nlev;
    // With a fixed proposal count the output size is bounded up front: one batched D2H (enqueued by the launch), one wait.
    const int want = q.p.fixed_num ? q.p.num_proposals : -1;
    int rc;
    const double *hY = nullptr;
    const float *hS = nullptr;
    unsigned char *blk = c->h_res[q.slot];
    auto drop = [&]() { c->pend.erase(c->pend.begin() + (long)idx); c->slot_busy[q.slot] = false; };
    if (q.copied) {
        const hipError_t e = hipEventSynchronize(c->ev_res[q.slot]);
        if (e != hipSuccess) { drop(); return fail(c, AZ_ERR_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e)); }
        hY = (const double *)(blk + RES_HDR);
        hS = (const float *)(blk + RES_HDR + (size_t)want * 32);
    } else {
        // (variable proposal count: nothing is queued behind this search)
        drop();
        HIPCHK(c, hipMemcpyAsync(blk, c->cnt, sizeof(AzCounts), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        int n = ((const AzCounts *)blk)->nsel;
        if (n > c->maxCand) n = c->maxCand;
        if ((rc = ensure_host(c, n > 0 ? n : 1)) != AZ_OK) return rc;
        if (n > 0) {
            HIPCHK(c, hipMemcpyAsync(c->h_Y, c->Yout, (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(c->h_S, c->Sout, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
        }
        hY = c->h_Y;              // (ensure_host may have moved them)
        hS = c->h_S;
    }
    if (q.copied) drop();
    c->last = q.p;
    const AzCounts &h = *(const AzCounts *)blk;
    if (st) {
        std::memset(st, 0, sizeof(*st));
        st->n_levels = nlev;
        st->n_candidates = h.ytot[q.cut ? q.cut : nlev];
        st->spec_rows = h.specU;
        st->root_deferred = q.defer;
        st->static_plan = q.is_static;
        st->search_form = q.batch ? 5 : (q.is_static ? 4 : (q.full == 2 ? 3 : (q.full == 1 ? 2 : (q.pair_mask ? 1 : 0))));
        st->n_reruns = q.reruns;
        const int *hc = reinterpret_cast<const int *>(&h);
        for (int i = 0; i < q.npass && i < AZ_MAX_LEVELS; ++i) {
            const int r = q.pass_src[i] >= 0 ? hc[q.pass_src[i]] : -q.pass_src[i] - 1;
            if (r <= 0) continue;
            // the tree levels whose rois the pass evaluated
            const int lv = q.pass_lv[i];
            const bool spec3 = nlev >= 3 && !(q.p.reserved & (1 | 4));         // (plan_search: n_spec == 3)
            int mask;
            if (lv < 0) {
                if (q.is_static || q.full) mask = (1 << nlev) - 1;
                else if (q.batch) mask = 3;                                    // (the root and its children)
                else mask = (q.cut == 2 ? 3 : 7) & ~(q.defer ? 1 : 0);
            } else {
                mask = 1 << lv;
                if ((q.pair_mask >> lv) & 1) mask |= 1 << (lv + 1);
                if (q.defer && spec3 && lv == 3) mask |= 1;
            }
            st->pass_levels[st->n_passes] = mask;
            st->pass_rows[st->n_passes++] = r;
        }
        for (int l = 0; l < nlev; ++l) {
            st->level_regions[l] = h.P[l];
            st->level_unique[l] = h.U[l];
            st->level_zoomed[l] = h.PZ[l];
            st->num_eval += h.P[l];
            if (h.P[l] > 0) st->depth = (q.p.reserved & 4) ? l : l + 1;   // tune.py counts k from 0
        }
    }
    // A search that has to be run again in another form is launched now (behind whatever is queued), its record staged
    // where the failed run's was, and collected from the back of the queue.
    auto rerun = [&](az_params p2) {
        const int err = h.err;
        (void)err;
        // (a queue that is full cannot take the rerun: the caller queued ahead, so the oldest other search is collected
        //  only after this one -- make room by running this rerun with the queue drained)
        if ((int)c->pend.size() >= az_ctx::AZ_QUEUE_MAX) return fail(c, AZ_ERR_STATE, "az_propose_fetch: no room to rerun a search in another form");
        // the rerun reads THIS search's map (a later one may have been handed over since)
        if (q.feat_is_copy && q.feat_gen != c->feat_gen)
            return fail(c, AZ_ERR_STATE, "az_propose_fetch: the queued search has to be rerun but its feature map copy was reallocated");
        const float *cur_feat = c->feat;
        const int cur_H = c->d.H, cur_W = c->d.W;
        c->feat = q.feat; c->d.H = q.fH; c->d.W = q.fW;
        int rc2 = launch_impl(c, &p2);
        c->feat = cur_feat; c->d.H = cur_H; c->d.W = cur_W;
        if (rc2) return rc2;
        c->pend.back().feat = q.feat; c->pend.back().fH = q.fH; c->pend.back().fW = q.fW;
        c->pend.back().reruns = q.reruns + 1;
        ++c->n_rerun_total;
        if (q.stage_dst && (rc2 = stage_impl(c, q.stage_dst, q.stage_cap)) != AZ_OK) return rc2;
        return fetch_entry(c, c->pend.size() - 1, boxes_out, scores_out, cap, n_out, st);
    };
    if (q.batch && c->batch_set)
        for (int l = 0; l < nlev; ++l) c->batch_set->rows_acc[l] += (l < 2) ? (l == 0 ? h.specU : 0) : h.PR[l];
    if ((h.err & 2048) && q.batch) {
        // a level of the batch held more rois than the head's buffers take rows: every image of it runs again on its own
        az_params p2 = q.p;
        return rerun(p2);
    }
    if ((h.err & 32) && q.is_static) {
        // a zoom score of the tree is not >= Tz (NaN): the one-pass plan's premise fails for this image -> level loop
        az_params p2 = q.p;
        p2.reserved |= 32;
        return rerun(p2);
    }
    if ((h.err & 64) && !(q.p.reserved & 64)) {
        // the pair-speculation rows of a level outgrew the tables: this image shape runs without them from now on
        if (c->nopair.size() >= 32) c->nopair.erase(c->nopair.begin());
        c->nopair.emplace_back(q.p.im_h, q.p.im_w);
        az_params p2 = q.p;
        p2.reserved = (p2.reserved | 64) & ~128;
        return rerun(p2);
    }
    if ((h.err & 1024) && q.cut) {
        // the search was enqueued up to level q.cut only (the previous search of the shape ended there) and this tree goes
        // on: run it in full (its result enters the history below as a search that did not end early)
        az_params p2 = q.p;
        p2.reserved |= 4096;
        return rerun(p2);
    }
    if ((h.err & 256) && !(q.p.reserved & 256)) {
        // the whole-tree pass did not hold a window this search needed (a _sift_dup survivor other than the full tree's):
        // repeat it level by level; its history then says "pruned tree" and the next search of the shape goes that way at once
        if (getenv("AZ_FULL_DEBUG")) fprintf(stderr, "az: whole-tree pass missed a window (%dx%d, err %d)\n", q.p.im_h, q.p.im_w, h.err);
        az_params p2 = q.p;
        p2.reserved = (p2.reserved | 256) & ~512;
        return rerun(p2);
    }
    if ((h.err & 8) && !(q.p.reserved & 2)) {
        // a fused level outgrew its LDS tables: rerun with the multi-launch kernels and remember
        // the image shape so that later calls skip the fused attempt -- first only for the levels after the
        // speculative ones (az_level.hip), then, if levels 1-3 themselves overflow, for everything
        const bool lv_was_on = !(q.p.reserved & 16) && c->level_fused_env != 0 &&
                               !(q.p.im_h == c->nofuse_lv_h && q.p.im_w == c->nofuse_lv_w);
        az_params p2 = q.p;
        const int ovf = h.scratch[5] - 1;          // the level whose fused geometry kernel overflowed (-1: an earlier stage)
        bool limited = false;
        if (lv_was_on && ovf > 3) {
            // a level behind the first fused one: the levels before it keep their fused kernels
            for (auto &e : c->lv_limits)
                if (e.h == q.p.im_h && e.w == q.p.im_w) { if (ovf < e.limit) { e.limit = ovf; limited = true; } }
            bool known = false;
            for (const auto &e : c->lv_limits) known = known || (e.h == q.p.im_h && e.w == q.p.im_w);
            if (!known) {
                if (c->lv_limits.size() >= 32) c->lv_limits.erase(c->lv_limits.begin());
                c->lv_limits.push_back({q.p.im_h, q.p.im_w, ovf});
                limited = true;
            }
        }
        if (limited) { }
        else if (lv_was_on) { c->nofuse_lv_h = q.p.im_h; c->nofuse_lv_w = q.p.im_w; p2.reserved |= 16; }
        else { c->nofuse_h = q.p.im_h; c->nofuse_w = q.p.im_w; p2.reserved |= 2; }
        return rerun(p2);
    }
    if (h.err)
        return fail(c, AZ_ERR_CAPACITY,
                    std::string("az_propose: ctx capacity exceeded (flags ") + std::to_string(h.err) +
                        "): raise az_set_limits");
    if (!q.is_static && !(q.p.reserved & 4)) {
        hint_load(c, q.p.im_h, q.p.im_w, nlev);           // (the shape's records: a search of another shape may have been launched since)
        if (c->hint_n > 0) {                              // the records move down by one, the oldest drops out
            for (int r = az_ctx::HINT_K - 2; r > 0; --r) c->hint_old[r] = c->hint_old[r - 1];
            std::memcpy(c->hint_old[0].rows, c->hint_rows, sizeof(c->hint_rows)); std::memcpy(c->hint_old[0].P, c->hint_P, sizeof(c->hint_P));
            std::memcpy(c->hint_old[0].PZ, c->hint_PZ, sizeof(c->hint_PZ)); std::memcpy(c->hint_old[0].U, c->hint_U, sizeof(c->hint_U));
            std::memcpy(c->hint_old[0].SPN, c->hint_SPN, sizeof(c->hint_SPN));
        }
        c->hint_n = c->hint_n < az_ctx::HINT_K ? c->hint_n + 1 : az_ctx::HINT_K;
        bool walked_full = true;
        for (int l = 0; walked_full && l + 1 < nlev; ++l) walked_full = h.P[l] > 0 && h.PZ[l] == h.P[l];
        c->hint_full_streak = walked_full ? c->hint_full_streak + 1 : 0;
        for (int l = 0; l < AZ_MAX_LEVELS; ++l) {
            const bool in = l < nlev;
            // rows of the pass at that level (fused level loop: PR; multi-launch forms: the level's unique rois)
            c->hint_rows[l] = in ? (h.PR[l] > 0 ? h.PR[l] : (((q.pair_mask >> (l > 0 ? l - 1 : 0)) & 1) && l > 0 ? 0 : h.U[l])) : 0;
            c->hint_P[l] = in ? h.P[l] : 0;
            c->hint_PZ[l] = in ? h.PZ[l] : 0;
            c->hint_U[l] = in ? h.U[l] : 0;
            c->hint_SPN[l] = (in && ((q.pair_mask >> l) & 1)) ? h.SPN[l] : -1;
        }
        c->hint_h = q.p.im_h; c->hint_w = q.p.im_w; c->hint_nlev = nlev;
        hint_store(c);
        int first_empty = 15;
        for (int l = 1; l < nlev && l < 15; ++l)
            if (h.P[l] == 0) { first_empty = l; break; }
        c->early_hist = (c->early_hist << 4) | (unsigned)first_empty;
        if (c->n_hist < 1000000) ++c->n_hist;
    }
    const int n = h.nsel;
    // (the candidate list stays readable only while no later search has been queued: it would be overwriting it)
    c->cand_n = c->pend.empty() ? h.ytot[q.cut ? q.cut : nlev] : -1;
    c->his_n = h.nhis;
    if (st) st->n_proposals = n;
    *n_out = n;
    if (n > cap) return fail(c, AZ_ERR_CAPACITY, "az_propose: output capacity too small");
    std::memcpy(boxes_out, hY, (size_t)n * 4 * sizeof(double));
    if (scores_out) std::memcpy(scores_out, hS, (size_t)n * sizeof(float));
    return AZ_OK;
}


int stage_impl(az_ctx *c, void *dst_dev, size_t cap_bytes)
{
    if (!c || c->pend.empty()) return fail(c, AZ_ERR_STATE, "az_propose_stage_result_dev without az_propose_launch");
    az_ctx::PendingSearch &q = c->pend.back();
    if (!q.p.fixed_num) return fail(c, AZ_ERR_STATE, "az_propose_stage_result_dev: fixed proposal count only");
    const size_t bytes = RES_HDR + (size_t)q.p.num_proposals * 36;
    if (!dst_dev || cap_bytes < bytes) return fail(c, AZ_ERR_INVALID, "az_propose_stage_result_dev: destination too small");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t ls = q.last_s ? q.last_s : c->stream;
    HIPCHK(c, hipMemcpyAsync(dst_dev, c->cnt, bytes, hipMemcpyDeviceToDevice, ls));
    // az_propose_fetch waits for the slot's event: recorded again HERE, behind the staging copy, so that "the record is
    // staged when az_propose_fetch returns" holds (the launch recorded it behind the host copy only)
    if (q.copied) HIPCHK(c, hipEventRecord(c->ev_res[q.slot], ls));
    if (ls == c->stream2 && c->stream2) HIPCHK(c, hipEventRecord(c->ev_s2, c->stream2));
    if (ls == c->stream3 && c->stream3) HIPCHK(c, hipEventRecord(c->ev_s3, c->stream3));
    q.stage_dst = dst_dev; q.stage_cap = cap_bytes;
    return AZ_OK;
}


// ------------------------------------------------------------------------------------------------------------------------
// A batch of images of one shape searched in lockstep (include/aznet_hip.h: az_batch_launch; az_batch.hip).
// Image b's tree lives in slots[b] (an az_ctx of its own); the head passes run in lane L's buffers on L's stream:
//   pass 0   the root and its children of every image (rows that depend on the image shape only: the first 1 + |B1| rows of
//            the cached speculative pre-pass), outputs straight into L's zoom_s / score_s / delta_s -- image b's at row
//            b * (1 + |B1|), a host-known offset; k_spec_levels (two fused levels) of every image in one launch
//   level l  (l = 2 .. nlev-1) gather of the images' unique rois -> ONE head pass -> scatter -> k_level_geom of every image
//            in one launch (the last level: k_final_select, which also makes the top-k into the image's result block)
// then every image's result block on its way to the host, as for a search launched alone.
namespace {

template <typename T> T *args_at(unsigned char *base, size_t &off, int n)
{
    off = (off + 15) & ~(size_t)15;
    T *p = reinterpret_cast<T *>(base + off);
    off += sizeof(T) * (size_t)n;
    return p;
}

void head_pass_batch(az_ctx *L, az_ctx::Batch &B, const AzHeadDims &d, const int *Mptr, int im_h, int im_w, double eps, float *zoom, float *score,
                     float *delta, double min_side, bool keep_flags, bool keys, bool many_rows)
{
    hipStream_t s = L->stream;
    // (gemm mode 3 -- int6 on the 16-bit matrix cores, every fp32 operand as three bf16 terms: the planes carry no per-map
    //  scale, so the images of a batch share a pass there as well; mode 2's fp16 terms are scaled per map: not taken)
    azk_roi_pool(s, nullptr, d, L->spatial_scale, B.rois_cat, Mptr, L->maxR, L->pool5, L->pool5p,
                 azk_act_plane_elems(L->maxR, d.K6), L->gemm_parts, 0, 0, nullptr, B.feats, B.feat_hw);
    const bool can12 = (d.n6 / 128) * L->S6 >= 256 && d.n6 % 128 == 0 && d.K6 % 32 == 0 &&
                       azk_fc_chunk(d.K6, L->S6) * L->S6 == d.K6 && azk_fc_chunk(d.K6, L->S6) >= 64 &&
                       L->gemm12_min_rows < 0x7fffffff;
    // (only the device knows the row count; both kernels are correct and bit-identical for any: the last batch's rows decide)
    if (L->gemm_parts)
        azk_fc_gemm_terms(s, L->pool5p, d.K6, azk_act_plane_elems(L->maxR, d.K6), L->W6p, d.K6, azk_weight_plane_elems(d.n6, d.K6), Mptr,
                          L->maxR, d.n6, d.K6, L->S6, azk_fc_chunk(d.K6, L->S6), L->part, L->gemm_parts, L->gscale);
    else if (can12 && many_rows)
        azk_fc_gemm12(s, L->pool5, d.K6, L->W6, d.K6, Mptr, L->maxR, d.n6, d.K6, L->S6, azk_fc_chunk(d.K6, L->S6), L->part, 0, nullptr);
    else
        azk_fc_gemm(s, L->pool5, d.K6, L->W6, d.K6, Mptr, L->maxR, d.n6, d.K6, L->S6, L->part, 1 << 30, nullptr);
    azk_fc_reduce(s, L->part, L->b6, Mptr, L->maxR, d.n6, L->S6, L->h6, d.n6, 1);
    float *p7 = L->part7 ? L->part7 : L->part;
    azk_fc_gemm(s, L->h6, d.n6, L->W7, d.n6, Mptr, L->maxR, d.n7, d.n6, L->S7, p7, 1 << 30, nullptr);
    azk_tail(s, p7, L->S7, L->b7, d.n7, L->Wt, L->bt, B.ubox_cat, Mptr, L->maxR, im_h, im_w, eps, zoom, score, delta,
             L->pred_u, keep_flags ? L->keep_u : nullptr, min_side, (keep_flags && keys) ? L->key_u : nullptr, B.row_hw);
}

}  // namespace

int batch_launch_impl(az_ctx *L, az_ctx::Batch &B, int n_all, az_ctx **slots_all, const az_params *pa_all, const float *const *maps_all,
                      const int *Hs_all, const int *Ws_all, int *not_taken)
{
    *not_taken = 0;
    int rc = check_ready(L, false, true);          // (join: the passes work in the lane's per-search head buffers)
    if (rc) return rc;
    if (!pa_all || n_all < 1 || n_all > AZ_BATCH_MAX || !slots_all || !maps_all || !Hs_all || !Ws_all)
        return fail(L, AZ_ERR_INVALID, "az_batch_launch: bad arguments");
    const az_params *p = &pa_all[0];                // (what the images of a batch must share is checked against the first)
    for (int b = 0; b < n_all; ++b) {
        const az_params &q = pa_all[b];
        if (Hs_all[b] <= 0 || Ws_all[b] <= 0 || q.im_h <= 0 || q.im_w <= 0 || !(q.scale > 0) || q.batch_size <= 0 || !(q.min_side > 0))
            return fail(L, AZ_ERR_INVALID, "az_batch_launch: bad arguments");
        if (!q.fixed_num || (q.reserved & 4)) return fail(L, AZ_ERR_INVALID, "az_batch_launch: fixed proposal count, not the tuner's variant");
        if (q.num_proposals != p->num_proposals || q.reserved != p->reserved || q.eps != p->eps || q.min_side != p->min_side)
            return fail(L, AZ_ERR_INVALID, "az_batch_launch: the images of a batch share num_proposals, eps, min_side and the flags");
    }
    const int k = p->num_proposals;
    if (k <= 0) return fail(L, AZ_ERR_INVALID, "az_batch_launch: num_proposals must be positive");
    if (k > AZ_TOPK_MAX) return fail(L, AZ_ERR_CAPACITY, "az_batch_launch: num_proposals > 4096");
    int nlev = 0;                                   // the batch's deepest tree; image b walks nl_all[b] levels
    int nl_all[AZ_BATCH_MAX];
    for (int b = 0; b < n_all; ++b) {
        nl_all[b] = num_levels(pa_all[b].im_h, pa_all[b].im_w, pa_all[b].min_side) - 1;
        nlev = nl_all[b] > nlev ? nl_all[b] : nlev;
    }
    for (int b = 0; b < n_all; ++b) {
        if (!slots_all[b] || !maps_all[b]) return fail(L, AZ_ERR_INVALID, "az_batch_launch: null slot / map");
        if (!slots_all[b]->pend.empty()) return fail(L, AZ_ERR_STATE, "az_batch_launch: an image slot still holds an unfetched search");
    }
    HIPCHK(L, hipSetDevice(L->device));
    hipStream_t s = L->stream;
    // (before anything can decide that the images are searched one by one: those searches use the slices, too)
    const size_t res_slot = RES_HDR + (size_t)AZ_TOPK_MAX * 36;
    if (!B.res_dev) {
        HIPCHK(L, hipMalloc((void **)&B.res_dev, res_slot * AZ_BATCH_MAX));
        HIPCHK(L, hipHostMalloc((void **)&B.res_host, res_slot * AZ_BATCH_MAX));
        HIPCHK(L, hipMemsetAsync(B.res_dev, 0, res_slot * AZ_BATCH_MAX, s));
    }
    // this batch's blocks: k proposals each, side by side
    const size_t res_stride = (RES_HDR + (size_t)k * 36 + 255) & ~(size_t)255;
    for (int b = 0; b < n_all; ++b) {
        az_ctx *t = slots_all[b];
        if (!t->h_res_own0) t->h_res_own0 = t->h_res[0];
        t->cnt = reinterpret_cast<AzCounts *>(B.res_dev + (size_t)b * res_stride);
        t->h_res[0] = B.res_host + (size_t)b * res_stride;
    }
    auto skip_at = [&](int line) {
        if (getenv("AZ_BATCH_DEBUG")) fprintf(stderr, "az: batch not taken in lockstep (az_search.hip:%d)\n", line);
        *not_taken = 1;
        return AZ_ERR_STATE;
    };
#define skip() skip_at(__LINE__)
    if (nlev > AZ_MAX_LEVELS || (p->reserved & (1 | 2 | 8 | 16)) || L->gemm_parts == 2) return skip();
    for (int b = 0; b < n_all; ++b) if (nl_all[b] < 3) return skip();        // (an image too small for two fused levels + one more)
    if (L->level_fused_env < 0) { const char *e = getenv("AZ_LEVEL_FUSED"); L->level_fused_env = (e && !atoi(e)) ? 0 : 1; }
    if (!L->level_fused_env) return skip();
    // the images may differ in shape (each has its own pre-pass, its own map size, its own clipping box) and in the number of
    // levels (an image's last level gets its final selection where the others get their mid-tree geometry kernel; it has no
    // rows in the passes after that); a shape one of the contexts has learnt not to take on the fused kernels keeps the batch
    // off them
    struct Pre { const float *urois; const double *B1; const int *choff, *Udev; int P1, CH; };
    std::vector<Pre> pre_all(n_all);
    long rows0_all = 0;
    for (int b = 0; b < n_all; ++b) {
        const az_params &q = pa_all[b];
        az_ctx *t = slots_all[b];
        for (const az_ctx *x : {(const az_ctx *)L, (const az_ctx *)t}) {
            if ((q.im_h == x->nofuse_h && q.im_w == x->nofuse_w) || (q.im_h == x->nofuse_lv_h && q.im_w == x->nofuse_lv_w)) return skip();
            for (const auto &e : x->lv_limits) if (e.h == q.im_h && e.w == q.im_w) return skip();
        }
        // the shape's pre-pass (B1, the rois of root + B1, counters): cached per shape on the lane
        SearchPlan sp{};
        sp.fused = true; sp.defer_root = false;
        if ((rc = ensure_spec_cache(L, &q, sp)) != AZ_OK) return rc;
        if (L->spc[0].h != q.im_h || L->spc[0].w != q.im_w) return skip();     // (the pre-pass outgrew the context: nofuse_*)
        pre_all[b] = {L->spec_urois[0], L->specB1[0], L->spec_choff[0], L->spec_U[0], L->spc[0].P1, L->spc[0].CH};
        rows0_all += 1 + L->spc[0].P1;
    }
    if ((size_t)rows0_all > (size_t)L->maxR) return skip();
    if (!B.off) {
        HIPCHK(L, hipMalloc((void **)&B.off, (AZ_BATCH_MAX + 2) * sizeof(int)));
        HIPCHK(L, hipMalloc((void **)&B.rois_cat, (size_t)L->maxR * 5 * sizeof(float)));
        HIPCHK(L, hipMalloc((void **)&B.ubox_cat, (size_t)L->maxR * 4 * sizeof(double)));
        HIPCHK(L, hipMalloc((void **)&B.feats, AZ_BATCH_MAX * sizeof(float *)));
        HIPCHK(L, hipMalloc((void **)&B.feat_hw, AZ_BATCH_MAX * 2 * sizeof(int)));
        HIPCHK(L, hipMalloc((void **)&B.row_hw, (size_t)L->maxR * 2 * sizeof(int)));
        HIPCHK(L, hipMemsetAsync(B.ubox_cat, 0, (size_t)L->maxR * 4 * sizeof(double), s));
    }
    if (B.gemm12_rows < 0) { const char *e = getenv("AZ_BATCH_GEMM12_ROWS"); B.gemm12_rows = e ? atoi(e) : L->gemm12_dual_rows; }
    const size_t need = 64 + ((sizeof(AzFusedArgs) + 16) + (sizeof(AzLevelArgs) + sizeof(AzFinalArgs) + 32) * (size_t)nlev) * AZ_BATCH_MAX;
    if (B.args_cap < need) {
        if (B.args_dev) { HIPCHK(L, hipStreamSynchronize(s)); hipFree(B.args_dev); hipHostFree(B.args_host); B.args_dev = nullptr; B.args_host = nullptr; B.args_cap = 0; }
        HIPCHK(L, hipMalloc((void **)&B.args_dev, need));
        HIPCHK(L, hipHostMalloc((void **)&B.args_host, need));
        B.args_cap = need;
    }
    // A batch whose levels would not fit the head's buffers (max_regions rows per pass) -- going by the rows per image of the
    // last batch fetched on this lane -- is enqueued as several lockstep programs, one after the other, of as many images each
    // as fit (a pass that overflows all the same marks its images: they are searched again alone by az_batch_fetch).
    int per = n_all;
    if (B.hint_n > 0) {
        long mx = 0;
        for (int l = 0; l < AZ_MAX_LEVELS; ++l) mx = B.rows_hint[l] > mx ? B.rows_hint[l] : mx;
        const double per_img = (double)mx / B.hint_n;
        if (per_img * n_all > 0.9 * L->maxR) per = (int)(0.9 * L->maxR / per_img);
        if (per < 1) per = 1;
    }
    size_t off = 0;
    for (int i0 = 0; i0 < n_all; i0 += per) {
        const int n = n_all - i0 < per ? n_all - i0 : per;
        az_ctx **slots = slots_all + i0;
        const float *const *maps = maps_all + i0;
        const az_params *pa = pa_all + i0;
        const int *Hs = Hs_all + i0, *Ws = Ws_all + i0;
        const Pre *pre = pre_all.data() + i0;
        const int *nl = nl_all + i0;
        int off0[AZ_BATCH_MAX + 1];                   // first row of every image in pass 0 (root + its children: host-known)
        off0[0] = 0;
        for (int b = 0; b < n; ++b) off0[b + 1] = off0[b] + 1 + pre[b].P1;
        AzHeadDims d = L->d;
        d.H = Hs[0]; d.W = Ws[0];                     // (RoIPool takes every image's own size from the batch's table)
        // ---- the geometry kernels' arguments, all levels, all images of the part: one block, one copy
        const size_t off_begin = off;
        AzFusedArgs *fa = args_at<AzFusedArgs>(B.args_host, off, n);
        const size_t off_fa = (size_t)((unsigned char *)fa - B.args_host);
        // (per level: the images that go on -- k_level_geom -- and the images whose last level it is -- k_final_select)
        std::vector<size_t> off_lv(nlev, 0), off_fin(nlev, 0);
        std::vector<AzLevelArgs *> la(nlev, nullptr);
        std::vector<AzFinalArgs *> fin(nlev, nullptr);
        std::vector<int> n_mid(nlev, 0), n_fin(nlev, 0);
        for (int l = 2; l < nlev; ++l) {
            la[l] = args_at<AzLevelArgs>(B.args_host, off, n); off_lv[l] = (size_t)((unsigned char *)la[l] - B.args_host);
            fin[l] = args_at<AzFinalArgs>(B.args_host, off, n); off_fin[l] = (size_t)((unsigned char *)fin[l] - B.args_host);
        }
        for (int b = 0; b < n; ++b) {
            az_ctx *t = slots[b];
            const az_params *p = &pa[b];
            const int P1 = pre[b].P1, rows0 = 1 + P1;
            auto INV = [&](int l) { return (l & 1) ? t->inv_odd : t->inv; };
            {
                AzFusedArgs a;
                std::memset(&a, 0, sizeof(a));
                a.cnt = t->cnt;
                a.B[0] = t->B[0]; a.B[1] = t->B[1]; a.srcB[0] = t->srcB[0]; a.srcB[1] = t->srcB[1];
                a.index = t->index; a.inv = INV(2); a.zr = t->zr; a.choff = t->choff; a.csrc = t->csrc;
                a.choff_all = pre[b].choff; a.specB1 = pre[b].B1;
                a.reset = 1; a.specP1 = P1; a.specCH = pre[b].CH; a.specU = rows0;
                a.ubox = t->ubox; a.pred_u = t->pred_u; a.Yall = t->Yall; a.Z = t->Z; a.child = t->child;
                a.zoom_u = t->zoom_u; a.score_u = t->score_u; a.delta_u = t->delta_u; a.Sall = t->Sall;
                a.zoom_s = L->zoom_s + (size_t)off0[b]; a.score_s = L->score_s + (size_t)off0[b] * AZ_NSUB;
                a.delta_s = L->delta_s + (size_t)off0[b] * 4 * AZ_NSUB;
                a.scale = p->scale; a.Tz = p->Tz; a.min_side = p->min_side; a.eps = p->eps; a.dedup = (float)p->dedup;
                a.batch = p->batch_size; a.im_h = p->im_h; a.im_w = p->im_w; a.nlev = nl[b]; a.n_fused = 2;
                a.capR = t->maxR; a.capCh = t->maxCh; a.capCand = t->maxCand;
                a.rois = t->rois; a.urois = t->urois; a.next_dedup = 1; a.defer_root = 0; a.cut_next = 0; a.cut_short = 0;
                a.spec_next = 0; a.choff_next = t->choff_pair; a.crow = t->crow; a.spatial_scale = L->spatial_scale;
                a.row_map = nullptr; a.root_row = 0; a.stab = nullptr; a.stabT = 0;
                a.pred_v = t->pred_v; a.score_v = t->score_v; a.zoom_v = t->zoom_v; a.keep_v = t->keep_v; a.key_v = t->key_v;
                fa[b] = a;
            }
            for (int l = 2; l + 1 < nl[b]; ++l) {
                AzLevelArgs a;
                std::memset(&a, 0, sizeof(a));
                const int cur = l & 1;
                a.cnt = t->cnt; a.level = l; a.nlev = nl[b]; a.cut_next = 0;
                a.B = t->B[cur]; a.Bnext = t->B[cur ^ 1];
                a.pred_u = t->pred_u; a.score_u = t->score_u; a.zoom_u = t->zoom_u; a.keep_u = t->keep_u; a.Uptr = &t->cnt->U[l];
                a.urois = t->urois; a.index = t->index; a.inv = INV(l); a.inv_next = INV(l + 1); a.ubox = t->ubox;
                a.Yall = t->Yall; a.Sall = t->Sall;
                a.scale = p->scale; a.Tz = p->Tz; a.min_side = p->min_side; a.dedup = (float)p->dedup;
                a.batch = p->batch_size; a.capR = t->maxR; a.capCh = t->maxCh; a.capCand = t->maxCand;
                a.force_root = 1; a.root_row = 0; a.lookup_next = 0; a.spec_next = 0;
                a.delta_u = t->delta_u; a.choff_all = t->choff_pair; a.choff_next = t->choff_pair; a.crow = t->crow;
                a.stab = nullptr; a.stabT = 0; a.root_row_full = 0; a.score_all = t->score_s; a.zoom_all = t->zoom_s;
                a.pred_v = t->pred_v; a.score_v = t->score_v; a.zoom_v = t->zoom_v; a.keep_v = t->keep_v; a.key_v = t->key_v;
                a.im_h = p->im_h; a.im_w = p->im_w; a.eps = p->eps; a.spatial_scale = L->spatial_scale;
                la[l][n_mid[l]++] = a;
            }
            {
                AzFinalArgs a;
                std::memset(&a, 0, sizeof(a));
                const int l = nl[b] - 1;
                a.cnt = t->cnt; a.level = l; a.inv = INV(l); a.key_u = t->key_u; a.pred_u = t->pred_u;
                a.score_u = t->score_u; a.zoom_u = t->zoom_u; a.Yall = t->Yall; a.Sall = t->Sall; a.Tz = p->Tz;
                a.force_root = 0; a.capCand = t->maxCand; a.k = k;
                a.Yout = (double *)((unsigned char *)t->cnt + RES_HDR);
                a.Sout = (float *)((unsigned char *)t->cnt + RES_HDR + (size_t)k * 32);
                fin[l][n_fin[l]++] = a;
            }
        }
        HIPCHK(L, hipMemcpyAsync(B.args_dev + off_begin, B.args_host + off_begin, off - off_begin, hipMemcpyHostToDevice, s));

        // ---- pass 0: root + B1 of every image
        AzGatherArgs g;
        std::memset(&g, 0, sizeof(g));
        g.n = n; g.capR = L->maxR; g.off_out = B.off; g.rois_cat = B.rois_cat; g.ubox_cat = B.ubox_cat; g.feats_out = B.feats;
        g.feat_hw_out = B.feat_hw; g.row_hw_out = B.row_hw;
        for (int b = 0; b < n; ++b) {
            g.rows[b] = pre[b].Udev + 1;                  // (ensure_spec_cache: the pass without the third level's rows)
            g.err[b] = nullptr;                           // (the image's counters are cleared by k_spec_levels, behind this pass)
            g.rois[b] = pre[b].urois; g.ubox[b] = nullptr; g.feat[b] = maps[b];
            g.fh[b] = Hs[b]; g.fw[b] = Ws[b]; g.im_h[b] = pa[b].im_h; g.im_w[b] = pa[b].im_w;
        }
        azk_batch_gather(s, g);
        const int *Mptr = B.off + AZ_BATCH_MAX + 1;
        head_pass_batch(L, B, d, Mptr, p->im_h, p->im_w, p->eps, L->zoom_s, L->score_s, L->delta_s, 0.0, false, false,
                        off0[n] >= B.gemm12_rows);
        azk_spec_levels_batch(s, reinterpret_cast<const AzFusedArgs *>(B.args_dev + off_fa), n);
        // ---- the levels
        for (int l = 2; l < nlev; ++l) {
            const bool last = n_fin[l] > 0;               // (some image's last level: the heads also emit the selection keys)
            for (int b = 0; b < n; ++b) {
                az_ctx *t = slots[b];
                g.rows[b] = &t->cnt->PR[l]; g.err[b] = &t->cnt->err; g.rois[b] = t->urois; g.ubox[b] = t->ubox; g.feat[b] = maps[b];
            }
            azk_batch_gather(s, g);
            head_pass_batch(L, B, d, Mptr, p->im_h, p->im_w, p->eps, L->zoom_u, L->score_u, L->delta_u, p->min_side, true, last,
                            (B.hint_n > 0 ? (long)B.rows_hint[l] * n / B.hint_n : 0) >= B.gemm12_rows);
            AzScatterArgs sc;
            std::memset(&sc, 0, sizeof(sc));
            sc.n = n; sc.off = B.off; sc.zoom = L->zoom_u; sc.score = L->score_u; sc.pred = L->pred_u; sc.keep = L->keep_u;
            sc.key = last ? L->key_u : nullptr;
            for (int b = 0; b < n; ++b) {
                az_ctx *t = slots[b];
                sc.zoom_d[b] = t->zoom_u; sc.score_d[b] = t->score_u; sc.pred_d[b] = t->pred_u; sc.keep_d[b] = t->keep_u; sc.key_d[b] = t->key_u;
            }
            azk_batch_scatter(s, sc);
            if (n_mid[l]) azk_level_geom_batch(s, reinterpret_cast<const AzLevelArgs *>(B.args_dev + off_lv[l]), n_mid[l]);
            if (n_fin[l]) azk_final_select_batch(s, reinterpret_cast<const AzFinalArgs *>(B.args_dev + off_fin[l]), n_fin[l]);
        }
        HIPCHK(L, hipGetLastError());
        // ---- every image's record on its way to the host; the searches enter the slots' queues
        for (int b = 0; b < n; ++b) {
            az_ctx *t = slots[b];
            az_ctx::PendingSearch q;
            q.p = pa[b]; q.nlev = nl[b]; q.batch = 1;
            q.npass = 0;
            q.pass_lv[q.npass] = -1; q.pass_src[q.npass++] = -(1 + pre[b].P1) - 1;
            for (int l = 2; l < nl[b]; ++l) {
                q.pass_lv[q.npass] = l;
                q.pass_src[q.npass++] = (int)(&t->cnt->PR[l] - reinterpret_cast<int *>(t->cnt));
            }
            t->feat = maps[b]; t->d.H = Hs[b]; t->d.W = Ws[b];
            q.feat = maps[b]; q.fH = Hs[b]; q.fW = Ws[b]; q.feat_gen = t->feat_gen; q.feat_is_copy = false;
            q.slot = 0;                                   // (the slot's queue is empty: its first result slot, a slice of the arena)
            if (b == 0) HIPCHK(L, hipMemcpyAsync(B.res_host + (size_t)i0 * res_stride, B.res_dev + (size_t)i0 * res_stride, res_stride * n, hipMemcpyDeviceToHost, s));
            HIPCHK(L, hipEventRecord(t->ev_res[q.slot], s));
            q.copied = true;
            q.last_s = s;
            t->last_s = s;
            t->cand_n = -1;
            t->slot_busy[q.slot] = true;
            t->pend.push_back(q);
        }

    }
    L->last_s = s;
    return AZ_OK;
#undef skip
}
