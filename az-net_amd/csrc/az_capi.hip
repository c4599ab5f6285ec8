// az_capi.hip -- the C ABI of libaznet_hip.so (include/aznet_hip.h): context, HBM buffers,
// the level loop of im_propose as one stream-ordered launch sequence, and the unit entry
// points.  There is no CPU fallback anywhere in this library: without a gfx950 device
// az_create fails with AZ_ERR_NO_DEVICE.
#include "az_dev.h"

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <string>
#include <vector>

#define AZ_VERSION_STR "aznet_hip 0.1 (gfx950)"

struct AzEventRec { std::string name; int level; hipEvent_t a, b; int slot; /* >= 0: an in-kernel span (a, b unused) */ };

constexpr size_t RES_HDR = 1024;    // AzCounts, padded, at the head of the result block
static_assert(sizeof(AzCounts) <= RES_HDR, "AzCounts outgrew its slot");

struct az_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int maxR = 16384, maxCand = 16384 * AZ_NSUB, maxCh = 65536;
    bool head_loaded = false;
    AzHeadDims d{};
    int S6 = 1, S7 = 1;
    // int6 on the 16-bit matrix cores (az_set_gemm_mode): 0 = off (fp32 MFMA everywhere), 2 = two fp16 terms of
    // x * 2^k / 3 MFMAs per product (~2^-21), 3 = three bf16 terms / 6 MFMAs (every fp32 value exactly)
    int gemm_parts = 0;
    unsigned short *W6p = nullptr, *pool5p = nullptr;
    float *gscale = nullptr;            // two-term (fp16) mode: {pool5 scale of this map, 1 / (sx * sw), scratch, scratch}
    float w6_scale = 0.f;               // power-of-two scale of the fp16 weight terms
    float spatial_scale = 0.0625f;                 // test_fc.prototxt:22
    // weights (HBM)
    float *W6 = nullptr, *b6 = nullptr, *W7 = nullptr, *b7 = nullptr, *Wt = nullptr, *bt = nullptr;
    // feature map
    const float *feat = nullptr;        // channel-last copy of the current map (what RoIPool reads)
    // [H][W][C] copies of NCHW maps, two of them used in turn: the map of a search that is still queued (and might have to
    // be run again in another form) survives the hand-over of the next image's map
    float *feat_owned[2] = {nullptr, nullptr};
    int feat_turn = 0;
    unsigned feat_gen = 0;              // bumped when the copies are reallocated
    float *feat_stage = nullptr;        // NCHW staging for host uploads
    size_t feat_owned_elems = 0;
    // level-loop buffers (HBM)
    AzCounts *cnt = nullptr;
    double *B[2] = {nullptr, nullptr};
    float *rois = nullptr, *urois = nullptr;
    long long *key = nullptr, *ckey = nullptr;
    int *grp = nullptr, *index = nullptr, *inv = nullptr, *choff = nullptr, *bc_c = nullptr, *bc_z = nullptr;
    // inv_index of the ODD levels of a search (even levels and the unit entry points: `inv`): a level's fused geometry
    // kernel writes the next level's inv_index while its second workgroup may still be reading this level's
    int *inv_odd = nullptr;
    unsigned char *first = nullptr, *cflag = nullptr, *zflag = nullptr, *keep_u = nullptr;
    double *ubox = nullptr, *pred_u = nullptr, *Yall = nullptr, *Z = nullptr, *child = nullptr, *Yout = nullptr;
    float *pool5 = nullptr, *part = nullptr, *h6 = nullptr, *h7 = nullptr;
    float *zoom_u = nullptr, *score_u = nullptr, *delta_u = nullptr, *Sall = nullptr, *Sout = nullptr;
    int *sel_idx = nullptr, *rank_part = nullptr;
    // speculative levels 1-3: provenance of zoomed regions / children / regions, head outputs of the pass
    int *zr = nullptr, *csrc = nullptr, *choff_all = nullptr, *srcB[2] = {nullptr, nullptr};
    float *zoom_s = nullptr, *score_s = nullptr, *delta_s = nullptr;
    // the speculative pre-pass depends on the image shape only: its outputs are kept per shape (one entry)
    // (two entries: with the root's row in the pass [0] / deferred to level 4's pass [1] -- a context whose images
    //  alternate between trees that reach level 4 and trees that do not keeps both)
    float *spec_urois[2] = {nullptr, nullptr};
    double *specB1[2] = {nullptr, nullptr};
    int *spec_choff[2] = {nullptr, nullptr}, *spec_U[2] = {nullptr, nullptr};
    struct SpecCache { int h = -1, w = -1; double scale = 0, min_side = 0; int P1 = 0, CH = 0, U = 0; } spc[2];
    // the pre-pass writes into the scratch buffers; its result is kept per image shape in exact-size buffers (a dataset
    // mixes shapes: a shape seen before costs neither the pre-pass nor its host synchronisation) and spec_urois / specB1 /
    // spec_choff / spec_U[defer] POINT at the entry of the shape in use
    float *spec_scr_urois[2] = {nullptr, nullptr};
    double *spec_scr_B1[2] = {nullptr, nullptr};
    int *spec_scr_choff[2] = {nullptr, nullptr};
    struct SpecEntry { int h = -1, w = -1, defer = 0; double scale = 0, min_side = 0; int P1 = 0, CH = 0, U = 0;
                       float *urois = nullptr; double *B1 = nullptr; int *choff = nullptr, *Udev = nullptr;
                       unsigned long long use = 0; };
    std::vector<SpecEntry> spec_store;
    unsigned long long spec_clock = 0;
    // Tz <= 0: the whole tree is a function of the image shape (az_static.hip); its rois / anchors / region -> row
    // map are kept per shape in exact-size HBM buffers (~100 B per roi: 70 KB for a 600x1000 image), least recently
    // used shapes are dropped beyond AZ_PLAN_CACHE entries; the per-level sizes stay on the host
    struct StaticPlan {
        int h = -1, w = -1, nlev = 0, batch = 0, Utot = 0, coop = 1;
        double scale = 0, min_side = 0, dedup = 0;
        int roff[AZ_MAX_LEVELS + 1] = {0}, U[AZ_MAX_LEVELS] = {0}, CH[AZ_MAX_LEVELS] = {0};
        float *urois = nullptr;
        double *ubox = nullptr;
        int *reg_u = nullptr, *cand_src = nullptr, *meta = nullptr;
        unsigned long long last_use = 0;
        // whole-tree speculation (SearchPlan::full): window table over the pass's rows, the speculative rows' map, the
        // pass's rois.  Two row sets per shape:
        //   fs[0] "tree":    the plan's non-root rows (the unique rois of the FULL tree) ++ extra rows (speculative rows
        //                    whose window the plan lacks) ++ the root.  Serves a search whose tree is the full tree; a
        //                    pruned tree may keep another _sift_dup survivor (same 10-px hash, other window) -> err bit 256.
        //   fs[1] "closure": one row per distinct RoIPool window among ALL regions any pruning can produce -- level l+1 =
        //                    every child of every region of level l, no _sift_dup (whichever duplicate survives is among
        //                    them) -- ++ the root.  Serves every Tz; never misses.
        struct FullSet {
            unsigned long long *htab = nullptr; unsigned hT = 0;
            int *spec_map = nullptr, *full_meta = nullptr;
            float *full_urois = nullptr; double *full_ubox = nullptr;
            int Ufull = 0, full_state = 0;    // 0: not built, 1: ready, -1: cannot be used for this shape
        } fs[2];
    };
    std::vector<StaticPlan *> plans;
    StaticPlan *plan = nullptr;               // the plan of the search being launched / in flight
    unsigned long long plan_clock = 0;
    int plan_cache_max = 64;
    unsigned *key_u = nullptr;                // selection keys of the decoded boxes (tail kernel), [row][11]
    std::vector<std::pair<int, int>> nostatic; // image shapes whose trees outgrew the plan buffers (a few; oldest dropped)
    int static_env = -1;                      // AZ_STATIC_TREE=0: always run the level loop (measurements)
    int last_static = 0;
    int final_env = 1;                        // AZ_FINAL_FUSED=0: separate candidate / selection kernels at the last level
    int hint_rows[AZ_MAX_LEVELS] = {0};       // rows of the head pass launched at each level in the last fetched level-loop search (kernel choice)
    // the last fetched level-loop search, per level: regions, zoomed regions, unique rois, pair-speculation rows (-1: none)
    int hint_P[AZ_MAX_LEVELS] = {0}, hint_PZ[AZ_MAX_LEVELS] = {0}, hint_U[AZ_MAX_LEVELS] = {0}, hint_SPN[AZ_MAX_LEVELS] = {0};
    int hint_h = -1, hint_w = -1, hint_nlev = 0;
    // ... kept per image shape (a dataset mixes a few dozen shapes: each keeps the history of ITS last search; the fields
    // above are the entry of the shape being launched / last fetched)
    struct ShapeHint { int h, w, nlev; int rows[AZ_MAX_LEVELS], P[AZ_MAX_LEVELS], PZ[AZ_MAX_LEVELS], U[AZ_MAX_LEVELS], SPN[AZ_MAX_LEVELS];
                       unsigned long long use; };
    std::vector<ShapeHint> hints;
    unsigned long long hint_clock = 0;
    int pair_env = -1;                        // AZ_PAIR_SPEC: 0 never, 1 by history (default), 2 always
    std::vector<std::pair<int, int>> nopair;  // image shapes whose pair-speculation rows outgrew the tables
    int full_env = -1;                        // AZ_FULL_SPEC: 0 never, 1 by history (default), 2 always
    int full_now = 0;                         // the search being launched takes the whole-tree pass: 1 = tree rows, 2 = closure
    int last_full = 0;
    // the closure's rows of the shape last looked at by the cost model (0: not built): what the one pass would cost
    int n_rerun_total = 0;                    // searches this context has had to run twice (any reason) since it was created
    // Two lanes (az_set_lanes): a second stream with its own per-search buffers (`twin`, an az_ctx of its own that shares
    // this context's head weights) takes every other queued search, so that consecutive images overlap on the GPU -- one
    // image's single-workgroup geometry kernels and its small head kernels run beside the other image's GEMM.
    az_ctx *twin = nullptr, *owner = nullptr;
    int lanes = 1, lane_next = 0, last_fetch_lane = 0;
    std::deque<int> lane_order;               // lanes of the searches launched through the public entry points, oldest first
    hipEvent_t ev_hand = nullptr;             // (in a twin) orders the lane behind the owner's stream when it reads the owner's map
    void *comm = nullptr;                     // ncclComm_t of az_rccl_init
    int comm_ranks = 0, comm_rank = 0;
    // the collective runs on a stream of its own, behind events of the lanes: in a lane's stream it would hold that lane's
    // next search back until the collective's kernel finds free CUs, i.e. until the OTHER lane's GEMM is done
    hipStream_t comm_stream = nullptr;
    hipEvent_t comm_ev[2] = {nullptr, nullptr};
    // cost of one head pass (RoIPool + int6 + reduce + int7 + heads) at a few row counts, measured on THIS device with HIP
    // events the first time a search is launched (calibrate_passes): what the choice between the search forms goes by
    struct PassCal { int state = 0; int n = 0; int rows[6] = {0}; double us[6] = {0}; } cal;   // state 0: not yet, 1: measured, -1: off
    double *pred_w = nullptr; float *score_w = nullptr, *zoom_w = nullptr; unsigned char *keep_w = nullptr; unsigned *key_w = nullptr;   // second *_v set
    int last_pair_mask = 0;                   // levels whose head pass carried pair-speculation rows (search in flight / last)
    // pair speculation: all-children offsets / child -> row of the level whose pass carries the rows; looked-up outputs
    int *choff_pair = nullptr, *crow = nullptr;
    double *pred_v = nullptr;
    float *score_v = nullptr, *zoom_v = nullptr;
    unsigned char *keep_v = nullptr;
    unsigned *key_v = nullptr;
    int gemm12_env = -1;
    int gemm12_min_rows = 161;                // rows from which a host-known launch takes az_head12.hip (AZ_GEMM12_MIN;
                                              // measured crossover with k_fc_splitk: 160 rows)
    int gemm12_dual_rows = 161;               // ... and from which a launch whose row count only the device knows takes it, going by the previous search
    // Fast R-CNN head on the shared map (az_load_det_head)
    bool det_loaded = false;
    int det_n6 = 0, det_n7 = 0, det_ncls = 0, det_S6 = 1, det_S7 = 1;
    std::vector<void *> allocs_det;
    float *dW6 = nullptr, *db6 = nullptr, *dW7 = nullptr, *db7 = nullptr, *dWt = nullptr, *dbt = nullptr;
    // 16-bit-term modes: the detection head's fc6 on the same kernel as int6 (its own weight planes, weight scale and
    // -- two fp16 terms -- its own {pool5 scale, 1 / (sx * sw)} pair)
    unsigned short *dW6p = nullptr; float *dgscale = nullptr; float det_w6_scale = 0.f;
    float *dh6 = nullptr, *dh7 = nullptr, *dpart = nullptr, *dprob_u = nullptr, *ddelta_u = nullptr, *dprob = nullptr;
    double *dpred_u = nullptr, *dpred = nullptr;
    // nms scratch (grown on demand)
    int nms_cap = 0;
    float *nms_dets = nullptr, *nms_sdets = nullptr;
    int *nms_order = nullptr;
    unsigned long long *nms_mask = nullptr;
    int *nms_rank = nullptr;            // [nms_cap] rank scratch of k_nms_rank_count: zero between calls
    long long *nms_keep = nullptr;
    unsigned char *h_nms = nullptr;     // host-mapped block of az_nms's small case
    unsigned char *h_nmsg = nullptr; size_t h_nmsg_cap = 0;     // ... of az_nms's general case (keep list + count)
    unsigned char *h_nmsb = nullptr; size_t h_nmsb_cap = 0; int *nms_done = nullptr; int nms_seq = 0;   // ... of az_nms_batched's
    unsigned nms_tag = 0;               // sequence number carried by every word an NMS kernel writes to host-mapped memory
    // tuner (az_eval.hip): anchor history of the last search, score pool over an image set
    double *hisB = nullptr;
    float *hisZ = nullptr;
    int capHis = 0;
    float *pool = nullptr, *pool_tmp = nullptr;
    unsigned long long *pool_n = nullptr, *pool_hist = nullptr;      // [2], [256]
    long long pool_cap = 0;
    // grow-on-demand scratch of the evaluation / front-end entry points
    void *ev_a = nullptr, *ev_b = nullptr, *ev_c = nullptr, *ev_d = nullptr, *ev_e = nullptr, *ev_f = nullptr,
         *ev_g = nullptr, *ev_h = nullptr;
    size_t ev_sz[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // front-end on a caller's stream (az_image_blob_dev_on): two pinned host slots and two device slots for the uint8 image,
    // used in turn; a slot's event says its last upload + kernel are done
    unsigned char *io_host[2] = {nullptr, nullptr}, *io_dev[2] = {nullptr, nullptr};
    size_t io_cap = 0;
    hipEvent_t io_ev[2] = {nullptr, nullptr};
    int io_turn = 0;
    // pinned host staging
    AzCounts *h_cnt = nullptr;
    double *h_Y = nullptr;
    float *h_S = nullptr;
    int h_cap = 0;
    // Searches launched and not yet fetched, oldest first (at most two: the host may enqueue the next image's launch
    // sequence while the GPU still works on the current one -- same stream, so the searches never overlap on the GPU).
    // With a fixed proposal count the result block's device-to-host copy is enqueued right behind the search's kernels,
    // into a pinned slot of its own; az_propose_fetch then only waits for that copy's event.
    struct PendingSearch {
        az_params p{};
        int nlev = 0, is_static = 0, defer = 0, pair_mask = 0, npass = 0, full = 0, reruns = 0;
        int pass_src[AZ_MAX_LEVELS + 2] = {0};
        void *stage_dst = nullptr;          // az_propose_stage_result_dev target
        size_t stage_cap = 0;
        int slot = 0;
        bool copied = false;                // result block already on its way to h_res[slot]
        const float *feat = nullptr;        // the map the search reads (a rerun in another form needs it again)
        int fH = 0, fW = 0;
        unsigned feat_gen = 0;
        bool feat_is_copy = false;          // `feat` is one of the ctx's own channel-last copies (gone if they are reallocated)
    };
    std::deque<PendingSearch> pend;
    unsigned char *h_res[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_res[3] = {nullptr, nullptr, nullptr};
    bool slot_busy[3] = {false, false, false};
    // parameters of the last FETCHED search
    az_params last{};
    int nofuse_h = -1, nofuse_w = -1;   // image shape for which the fused levels 1-3 overflowed
    int nofuse_lv_h = -1, nofuse_lv_w = -1;   // ... for which the first level after the speculative ones outgrew the fused level kernel
    // image shapes whose level `limit` (> the first fused level) outgrew the fused level kernel: the levels before it stay
    // on it, the step from level `limit` on runs on the multi-launch kernels
    struct LvLimit { int h, w, limit; };
    std::vector<LvLimit> lv_limits;
    int defer_root_env = -1;            // AZ_DEFER_ROOT=0: keep the root's row in the speculative pass (measurements)
    int level_fused_env = -1;           // AZ_LEVEL_FUSED=0: keep levels >= 4 as separate launches (measurements)
    struct GraphEntry { hipGraphExec_t exec; int npass; int pass_src[AZ_MAX_LEVELS + 2]; };
    std::map<std::string, GraphEntry> graphs;        // captured launch sequences (az_set_graphs)
    int use_graphs = -1;                             // -1: take the AZ_GRAPH environment variable
    int last_defer = 0;
    // head passes of the search being enqueued / last launched: where each one's row count lives
    // (>= 0: int index into AzCounts; < 0: -(rows + 1), a count the host knows)
    int npass = 0;
    int pass_src[AZ_MAX_LEVELS + 2] = {0};
    int his_n = 0;                      // rows of the anchor history of the last fetched tuner search
    int cand_n = -1;                    // candidates of the last fetched search still in Yall/Sall (-1: overwritten)
    // profiling
    int profiling = 0;
    int event_errors = 0;              // hipEvent* calls that failed while profiling
    std::vector<AzEventRec> events;
    // profiling bit 3: the fc GEMM launches time THEMSELVES (AzSpan: first workgroup in, last workgroup out on the 100 MHz
    // clock) into slots of this ring -- exact also when another lane's kernels delay the launch, and free of the ~7 us of
    // stream time an event pair costs
    unsigned long long *span_ring = nullptr;
    int span_next = 0;
    static constexpr int SPAN_SLOTS = 32768;
    std::vector<hipEvent_t> event_pool;   // recycled events
    std::vector<void *> allocs;        // head-sized buffers (az_load_head)
    std::vector<void *> allocs_geom;   // geometry buffers (first use)
    bool geom_ready = false;
};

namespace {

int fail(az_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(c, call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail((c), AZ_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
int dalloc(az_ctx *c, T **p, size_t n, bool geom = false)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T) + 256);
    if (e != hipSuccess)
        return fail(c, AZ_ERR_HIP, std::string("hipMalloc(") + std::to_string(n * sizeof(T)) + " B): " +
                                       hipGetErrorString(e));
    (geom ? c->allocs_geom : c->allocs).push_back(q);
    *p = (T *)q;
    return AZ_OK;
}

template <typename T>
int dalloc_det(az_ctx *c, T **p, size_t n)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T) + 256);
    if (e != hipSuccess) return fail(c, AZ_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    c->allocs_det.push_back(q);
    *p = (T *)q;
    return AZ_OK;
}

void free_all(az_ctx *c)
{
    for (void *p : c->allocs) hipFree(p);
    c->allocs.clear();
}


// Buffers that depend only on the ctx limits (region / candidate capacity).
int ensure_geom(az_ctx *c)
{
    if (c->geom_ready) return AZ_OK;
    hipError_t e0 = hipSetDevice(c->device);
    if (e0 != hipSuccess) return fail(c, AZ_ERR_HIP, "hipSetDevice failed");
    const size_t R = (size_t)c->maxR, CAND = (size_t)c->maxCand, CH = (size_t)c->maxCh;
    int rc;
#define A(p, n) if ((rc = dalloc(c, &c->p, (n), true)) != AZ_OK) return rc
    {   // result block: the counters, then (fixed proposal count) the selected boxes and scores, so that
        // az_propose_fetch is ONE device-to-host copy
        unsigned char *blk = nullptr;
        if ((rc = dalloc(c, &blk, RES_HDR + (size_t)AZ_TOPK_MAX * 36, true)) != AZ_OK) return rc;
        c->cnt = (AzCounts *)blk;
    }
    A(B[0], R * 4); A(B[1], R * 4); A(rois, R * 5); A(urois, R * 5); A(key, R); A(ckey, CH);
    A(grp, R); A(index, R); A(inv, R); A(inv_odd, R); A(choff, R); A(bc_c, (R * AZ_NSUB + 255) / 256 + 1);
    A(bc_z, (R * AZ_NSUB + 255) / 256 + 1);
    A(first, CH > R ? CH : R); A(cflag, R * AZ_NSUB); A(zflag, R); A(keep_u, R * AZ_NSUB);
    A(ubox, R * 4); A(pred_u, R * AZ_NSUB * 4); A(Yall, CAND * 4); A(Z, R * 4); A(child, CH * 4);
    A(Yout, CAND * 4); A(Sout, CAND); A(sel_idx, CAND); A(rank_part, (size_t)azk_topk_scratch_ints((int)CAND));
    A(zoom_u, R); A(score_u, R * AZ_NSUB); A(delta_u, R * 4 * AZ_NSUB); A(Sall, CAND);
    A(zr, R); A(csrc, CH); A(choff_all, R); A(srcB[0], R); A(srcB[1], R);
    A(zoom_s, R); A(score_s, R * AZ_NSUB); A(delta_s, R * 4 * AZ_NSUB);
    for (int i = 0; i < 2; ++i) { A(spec_scr_urois[i], R * 5); A(spec_scr_B1[i], R * 4); A(spec_scr_choff[i], R); }
    A(key_u, R * AZ_NSUB);
    A(choff_pair, R); A(crow, CH > 8192 ? CH : 8192);
    A(pred_v, R * AZ_NSUB * 4); A(score_v, R * AZ_NSUB); A(zoom_v, R); A(keep_v, R * AZ_NSUB); A(key_v, R * AZ_NSUB);
    A(pred_w, R * AZ_NSUB * 4); A(score_w, R * AZ_NSUB); A(zoom_w, R); A(keep_w, R * AZ_NSUB); A(key_w, R * AZ_NSUB);
#undef A
    if (hipMemset(c->ubox, 0, R * 4 * sizeof(double)) != hipSuccess) return fail(c, AZ_ERR_HIP, "hipMemset failed");
    // (a search whose fused level kernel overflows is rerun by the host, but the kernels already enqueued behind it still
    //  run, on whatever the level's inv_index buffer holds: it must always hold valid rows)
    if (hipMemset(c->inv, 0, R * sizeof(int)) != hipSuccess || hipMemset(c->inv_odd, 0, R * sizeof(int)) != hipSuccess ||
        hipMemset(c->index, 0, R * sizeof(int)) != hipSuccess)
        return fail(c, AZ_ERR_HIP, "hipMemset failed");
    c->geom_ready = true;
    return AZ_OK;
}

// Profiling modes (az_set_profiling): bit 0 = time the GEMM launches only, bit 1 = time every
// launch group, bit 2 = keep events across az_propose calls (read them once at the end).
struct Timed {
    az_ctx *c; bool on; hipEvent_t a{}, b{}; const char *name; int level;
    // (events are recycled through c->event_pool: creating one costs about as much as recording it)
    static bool grab(az_ctx *c, hipEvent_t *e)
    {
        if (!c->event_pool.empty()) { *e = c->event_pool.back(); c->event_pool.pop_back(); return true; }
        return hipEventCreate(e) == hipSuccess;
    }
    // AZ_TRACE=1 (debugging): every launch group is announced on stderr and waited for, so a faulting kernel is the one
    // named last
    static bool trace() { static const bool t = getenv("AZ_TRACE") && atoi(getenv("AZ_TRACE")); return t; }
    Timed(az_ctx *c_, const char *n, int l, int cls = 2) : c(c_), name(n), level(l)
    {
        if (trace()) { fprintf(stderr, "az[%p]: %s L%d ...", (void *)c_, n, l); fflush(stderr); }
        on = (c_->profiling & 2) || ((c_->profiling & 1) && cls == 1);
        if (!on) return;
        // a failed event call drops this measurement (and is reported by az_last_kernel_times), never the search
        if (!grab(c, &a)) { on = false; ++c->event_errors; return; }
        if (!grab(c, &b)) { hipEventDestroy(a); on = false; ++c->event_errors; return; }
        if (hipEventRecord(a, c->stream) != hipSuccess) { hipEventDestroy(a); hipEventDestroy(b); on = false; ++c->event_errors; }
    }
    ~Timed()
    {
        if (trace()) { const hipError_t e = hipStreamSynchronize(c->stream); fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e)); }
        if (!on) return;
        if (hipEventRecord(b, c->stream) != hipSuccess) { hipEventDestroy(a); hipEventDestroy(b); ++c->event_errors; return; }
        c->events.push_back({name, level, a, b, -1});
    }
};

void clear_events(az_ctx *c)
{
    for (auto &e : c->events) {
        if (e.slot >= 0) continue;
        for (hipEvent_t ev : {e.a, e.b}) {
            if (c->event_pool.size() < 4096) c->event_pool.push_back(ev); else hipEventDestroy(ev);
        }
    }
    c->events.clear();
}

// K of lib/detect/test.py:365-368 (Python-2 integer division when MIN_SIDE is integral).
int num_levels(int h, int w, double min_side)
{
    const int side = h < w ? h : w;
    double q;
    if (min_side == std::floor(min_side) && min_side >= 1.0) q = (double)(side / (int)min_side);
    else q = (double)side / min_side;
    if (!(q >= 1.0)) return 0;
    return (int)(std::log2(q) + 1.0);
}

int ensure_host(az_ctx *c, int cap)
{
    if (cap <= c->h_cap) return AZ_OK;
    if (c->h_Y) { hipHostFree(c->h_Y); hipHostFree(c->h_S); }
    HIPCHK(c, hipHostMalloc((void **)&c->h_Y, (size_t)cap * 4 * sizeof(double)));
    HIPCHK(c, hipHostMalloc((void **)&c->h_S, (size_t)cap * sizeof(float)));
    c->h_cap = cap;
    return AZ_OK;
}

int set_count(az_ctx *c, int *dptr, int v)
{
    // (a 32-bit fill carries the value in the command: nothing on this frame to keep alive, no synchronisation)
    HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)dptr, v, 1, c->stream));
    return AZ_OK;
}

// Two-term (fp16) mode: the scale of this map's pool5 terms, once per enqueued search / head forward (one small launch).
void prep_scale(az_ctx *c)
{
    if (c->gemm_parts == 2 && c->feat)
        azk_feat_scale(c->stream, c->feat, (long long)c->d.C * c->d.H * c->d.W, c->gscale, c->w6_scale);
}

// One forward of the head on the `U` rois in ctx->urois (anchors in ctx->ubox); scores and
// deltas go to the given arrays, decoded boxes to ctx->pred_u.
void launch_head(az_ctx *c, const int *Uptr, int level, int im_h, int im_w, double eps, float *zoom, float *score,
                 float *delta, double min_side = 0.0, bool keep_flags = false, int coop_tail = 0,
                 const float *urois = nullptr, const double *ubox = nullptr, int rows_hint = 0, bool keys = false)
{
    const AzHeadDims &d = c->d;
    if (c->npass < AZ_MAX_LEVELS + 2) {
        const int *c0 = reinterpret_cast<const int *>(c->cnt);
        const bool in_cnt = Uptr >= c0 && Uptr < c0 + sizeof(AzCounts) / sizeof(int);
        c->pass_src[c->npass++] = in_cnt ? (int)(Uptr - c0) : -(rows_hint > 0 ? rows_hint : 0) - 1;
    }
    if (c->gemm12_env < 0) {            // AZ_GEMM12_MIN=<rows> (0: never): measurements
        const char *f = getenv("AZ_GEMM12_MIN");
        if (f) c->gemm12_min_rows = atoi(f) > 0 ? atoi(f) : 0x7fffffff;
        c->gemm12_env = 1;
    }
    { Timed t(c, "roi_pool", level);
      azk_roi_pool(c->stream, c->feat, d, c->spatial_scale, urois ? urois : c->urois, Uptr, c->maxR, c->pool5, c->pool5p,
                   azk_act_plane_elems(c->maxR, d.K6), c->gemm_parts, 0, coop_tail, c->gemm_parts == 2 ? c->gscale : nullptr); }
    // (profiling bit 3: the fp32 GEMM launches record their own span instead of an event pair)
    auto span_slot = [&](const char *name) -> unsigned long long * {
        if (!(c->profiling & 8) || !c->span_ring || c->span_next >= az_ctx::SPAN_SLOTS) return nullptr;
        const int sl = c->span_next++;
        c->events.push_back({name, level, nullptr, nullptr, sl});
        return c->span_ring + 2 * (size_t)sl;
    };
    const int prof_keep = c->profiling;
    unsigned long long *ts6 = c->gemm_parts ? nullptr : span_slot("fc6_gemm");
    if (ts6) c->profiling &= ~(1 | 2);                     // (no event pair around a launch that times itself)
    { Timed t(c, "fc6_gemm", level, 1);
      if (c->gemm_parts)
          azk_fc_gemm_terms(c->stream, c->pool5p, d.K6, azk_act_plane_elems(c->maxR, d.K6), c->W6p, d.K6, azk_weight_plane_elems(d.n6, d.K6), Uptr,
                           c->maxR, d.n6, d.K6, c->S6, azk_fc_chunk(d.K6, c->S6), c->part, c->gemm_parts, c->gscale);
      else {
          const bool can12 = (d.n6 / 128) * c->S6 >= 256 && d.n6 % 128 == 0 && d.K6 % 32 == 0 &&
                             azk_fc_chunk(d.K6, c->S6) * c->S6 == d.K6 && azk_fc_chunk(d.K6, c->S6) >= 64 &&
                             c->gemm12_min_rows < 0x7fffffff;
          if (can12 && rows_hint >= c->gemm12_min_rows)
              // the caller knows the row count on the host (a one-pass plan): many rows -> one weight tile per 12 strips
              azk_fc_gemm12(c->stream, c->pool5, d.K6, c->W6, d.K6, Uptr, c->maxR, d.n6, d.K6, c->S6,
                            azk_fc_chunk(d.K6, c->S6), c->part, 0, ts6);
          else if (can12 && rows_hint == -1)
              // only the device knows the row count, and the last search had many rows at this level: the many-row
              // kernel takes the launch.  Both kernels are correct (and bit-identical) for any row count; a wrong guess
              // costs efficiency, never a result.
              azk_fc_gemm12(c->stream, c->pool5, d.K6, c->W6, d.K6, Uptr, c->maxR, d.n6, d.K6, c->S6,
                            azk_fc_chunk(d.K6, c->S6), c->part, 0, ts6);
          else
              azk_fc_gemm(c->stream, c->pool5, d.K6, c->W6, d.K6, Uptr, c->maxR, d.n6, d.K6, c->S6, c->part, 1 << 30, ts6);
      } }
    c->profiling = prof_keep;
    { Timed t(c, "fc6_reduce", level);
      azk_fc_reduce(c->stream, c->part, c->b6, Uptr, c->maxR, d.n6, c->S6, c->h6, d.n6, 1); }
    unsigned long long *ts7 = span_slot("fc7_gemm");
    if (ts7) c->profiling &= ~(1 | 2);
    { Timed t(c, "fc7_gemm", level, 1);
      azk_fc_gemm(c->stream, c->h6, d.n6, c->W7, d.n6, Uptr, c->maxR, d.n7, d.n6, c->S7, c->part, 1 << 30, ts7); }
    c->profiling = prof_keep;
    { Timed t(c, "tail", level);       // (finishes int7 as well: slab sum + bias + ReLU while staging its rows)
      azk_tail(c->stream, c->part, c->S7, c->b7, d.n7, c->Wt, c->bt, ubox ? ubox : c->ubox, Uptr, c->maxR, im_h, im_w,
               eps, zoom, score, delta, c->pred_u, keep_flags ? c->keep_u : nullptr, min_side,
               (keep_flags && keys) ? c->key_u : nullptr); }
}

// Scratch slot `i` of the evaluation entry points, grown to at least `bytes`.
int ev_grow(az_ctx *c, int i, void **slot, size_t bytes)
{
    if (bytes <= c->ev_sz[i] && *slot) return AZ_OK;
    if (*slot) hipFree(*slot);
    *slot = nullptr;
    c->ev_sz[i] = 0;
    const size_t want = bytes + bytes / 2 + 256;
    hipError_t e = hipMalloc(slot, want);
    if (e != hipSuccess) return fail(c, AZ_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    c->ev_sz[i] = want;
    return AZ_OK;
}

void free_plan(az_ctx::StaticPlan *q)
{
    for (void *p : {(void *)q->urois, (void *)q->ubox, (void *)q->reg_u, (void *)q->cand_src, (void *)q->meta})
        if (p) hipFree(p);
    q->urois = nullptr; q->ubox = nullptr; q->reg_u = nullptr; q->cand_src = nullptr; q->meta = nullptr;
    for (auto &f : q->fs) {
        for (void *p : {(void *)f.htab, (void *)f.spec_map, (void *)f.full_meta, (void *)f.full_urois, (void *)f.full_ubox})
            if (p) hipFree(p);
        f = az_ctx::StaticPlan::FullSet();
    }
}

int check_geom(az_ctx *c)
{
    if (!c) return AZ_ERR_INVALID;
    return ensure_geom(c);
}

int check_ready(az_ctx *c, bool need_feat)
{
    if (!c) return AZ_ERR_INVALID;
    if (!c->head_loaded) return fail(c, AZ_ERR_STATE, "az_load_head has not been called");
    if (need_feat && !c->feat) return fail(c, AZ_ERR_STATE, "no feature map set");
    return AZ_OK;
}

// NMS results in host-mapped memory are polled by the host.  Words written by the GPU may become visible out of order
// (posted PCIe writes), so each word carries the call's sequence number and is taken only once it shows it.
unsigned nms_next_tag(az_ctx *c)
{
    do { ++c->nms_tag; } while (c->nms_tag == 0u || (c->nms_tag & 0x3FFFFFu) == 0u);
    return c->nms_tag;
}

// true when every one of the n keep words shows `tag` (spins a bounded number of times on each)
bool nms_keep_tagged(const long long *hk, int n, unsigned tag, long spins)
{
    for (int i = 0; i < n; ++i) {
        const volatile long long *w = hk + i;
        long k = 0;
        while ((unsigned)((unsigned long long)*w >> 32) != tag) if (++k > spins) return false;
    }
    return true;
}

}  // namespace

// ======================================================================================
extern "C" {

const char *az_version(void) { return AZ_VERSION_STR; }

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

int az_destroy(az_ctx *c)
{
    if (!c) return AZ_ERR_INVALID;
    destroy_twin(c);
    if (c->comm) { if (c->comm_stream) hipStreamSynchronize(c->comm_stream); azk_rccl_destroy(c->comm); c->comm = nullptr; }
    if (c->comm_stream) { hipStreamDestroy(c->comm_stream); c->comm_stream = nullptr; }
    for (auto &e : c->comm_ev) if (e) { hipEventDestroy(e); e = nullptr; }
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
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
    if (c->feat_owned[0]) { hipFree(c->feat_owned[0]); hipFree(c->feat_owned[1]); hipFree(c->feat_stage); }
    for (void *p : {c->ev_a, c->ev_b, c->ev_c, c->ev_d, c->ev_e, c->ev_f, c->ev_g, c->ev_h, (void *)c->hisB,
                    (void *)c->hisZ, (void *)c->pool, (void *)c->pool_tmp, (void *)c->pool_n, (void *)c->pool_hist})
        if (p) hipFree(p);
    if (c->nms_dets) { hipFree(c->nms_dets); hipFree(c->nms_sdets); hipFree(c->nms_order); hipFree(c->nms_mask); hipFree(c->nms_keep); hipFree(c->nms_rank); }
    if (c->h_cnt) hipHostFree(c->h_cnt);
    for (int i = 0; i < 3; ++i) {
        if (c->h_res[i]) hipHostFree(c->h_res[i]);
        if (c->ev_res[i]) hipEventDestroy(c->ev_res[i]);
    }
    for (int i = 0; i < 2; ++i) {
        if (c->io_host[i]) hipHostFree(c->io_host[i]);
        if (c->io_dev[i]) hipFree(c->io_dev[i]);
        if (c->io_ev[i]) hipEventDestroy(c->io_ev[i]);
    }
    if (c->ev_hand) hipEventDestroy(c->ev_hand);
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
    destroy_twin(c);                          // (the second lane reads this head's buffers: rebuilt at the next launch)
    free_all(c);
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

static int set_feature_map_common(az_ctx *c, const float *src, bool src_is_host, int C, int H, int W,
                                  bool wait = true)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!src || C != c->d.C || H <= 0 || W <= 0)
        return fail(c, AZ_ERR_INVALID, "feature map: channel count must match the loaded head");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)C * H * W;
    if (n > c->feat_owned_elems) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->feat_owned[0]) { hipFree(c->feat_owned[0]); hipFree(c->feat_owned[1]); hipFree(c->feat_stage); }
        c->feat_owned[0] = c->feat_owned[1] = c->feat_stage = nullptr; c->feat_owned_elems = 0;
        ++c->feat_gen;
        HIPCHK(c, hipMalloc((void **)&c->feat_owned[0], n * 4));
        HIPCHK(c, hipMalloc((void **)&c->feat_owned[1], n * 4));
        HIPCHK(c, hipMalloc((void **)&c->feat_stage, n * 4));
        c->feat_owned_elems = n;
    }
    const float *nchw = src;
    if (src_is_host) {
        HIPCHK(c, hipMemcpyAsync(c->feat_stage, src, n * 4, hipMemcpyHostToDevice, c->stream));
        nchw = c->feat_stage;
    }
    // RoIPool reads the map channel-last: one transpose per image, outside the level loop.
    c->feat_turn ^= 1;
    azk_nchw_to_nhwc(c->stream, nchw, c->feat_owned[c->feat_turn], C, H * W);
    if (wait) HIPCHK(c, hipStreamSynchronize(c->stream));      // the caller may now reuse / free `src`
    c->feat = c->feat_owned[c->feat_turn];
    c->d.H = H; c->d.W = W;
    return AZ_OK;
}

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

// Which form of the search a call takes.
struct SearchPlan { int n_spec; bool fused, fused_lv, defer_root; int pair_mask; int lv_limit; int full; /* 0 / 1 tree rows / 2 closure */ };

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
static int pair_plan(az_ctx *c, const az_params *p, int nlev, int n_spec, bool fused_lv, int lv_limit)
{
    if (c->pair_env < 0) { const char *e = getenv("AZ_PAIR_SPEC"); c->pair_env = e ? atoi(e) : 1; }
    if (!fused_lv || (p->reserved & 64) || c->pair_env == 0) return 0;
    for (const auto &hw : c->nopair)
        if (hw.first == p->im_h && hw.second == p->im_w) return 0;
    const bool force = (p->reserved & 128) || c->pair_env == 2;
    const bool hist = c->hint_h == p->im_h && c->hint_w == p->im_w && c->hint_nlev == nlev;
    int mask = 0;
    for (int l = n_spec; l + 1 < nlev && l < lv_limit; ++l) {      // (the lookup runs in level l's fused geometry kernel)
        bool want = force;
        if (!want && hist && c->hint_P[l] > 0 && c->hint_U[l + 1] > 0) {
            // rows the speculation adds: what it added last time, else level l+1's unique rois scaled by parents / zoomed parents
            const double S = c->hint_SPN[l] >= 0 ? (double)c->hint_SPN[l]
                                                 : (double)c->hint_U[l + 1] * c->hint_P[l] / (c->hint_PZ[l] > 0 ? c->hint_PZ[l] : 1);
            const double with = pass_us(c, c->hint_U[l] + S) + PASS_OVERHEAD_US + LOOKUP_US;
            const double without = pass_us(c, c->hint_U[l]) + pass_us(c, c->hint_U[l + 1]) + 2 * PASS_OVERHEAD_US;
            want = with < without && c->hint_U[l] + S + 2 < c->maxR;
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
    q.lv_limit = AZ_MAX_LEVELS + 1;
    for (const auto &e : c->lv_limits)
        if (e.h == p->im_h && e.w == p->im_w) q.lv_limit = e.limit;
    q.pair_mask = pair_plan(c, p, nlev, q.n_spec, q.fused_lv, q.lv_limit);
    // whole-tree speculation (decided and prepared by az_propose_launch: full_prepare): one head pass over the rows of
    // the image shape's full tree, every level's outputs by window lookup -- no deferred root, no pair rows
    q.full = (c->full_now && q.fused && q.fused_lv && q.n_spec == 3 && q.lv_limit >= q.n_spec && c->plan &&
              c->plan->fs[c->full_now - 1].full_state == 1 && plan_is_for(*c->plan, p, nlev)) ? c->full_now : 0;
    if (q.full) { q.defer_root = false; q.pair_mask = 0; }
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
    hipStream_t s = c->stream;
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
            c->hint_h = h; c->hint_w = w; c->hint_nlev = nlev;
            e.use = ++c->hint_clock;
            return;
        }
    c->hint_h = -1; c->hint_w = -1; c->hint_nlev = 0;          // no search of this shape seen (yet)
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

// What the level-by-level form the context would pick for this shape (pair_plan on the same history) costs, in us.
static double level_forms_cost(az_ctx *c, int nlev, int n_spec, int specU, int pair_mask)
{
    double t = pass_us(c, specU) + PASS_OVERHEAD_US;
    for (int l = n_spec; l < nlev; ++l) {
        if (c->hint_U[l] <= 0) break;
        if ((pair_mask >> l) & 1) {
            const double S = c->hint_SPN[l] >= 0 ? (double)c->hint_SPN[l]
                                                 : (double)c->hint_U[l + 1] * c->hint_P[l] / (c->hint_PZ[l] > 0 ? c->hint_PZ[l] : 1);
            t += pass_us(c, c->hint_U[l] + S) + PASS_OVERHEAD_US + LOOKUP_US;
            ++l;
        } else
            t += pass_us(c, c->hint_U[l]) + PASS_OVERHEAD_US;
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
    const bool have_hist = c->hint_h == p->im_h && c->hint_w == p->im_w && c->hint_nlev == nlev;
    // the previous search of this shape walked the FULL tree (every region zoomed at every level but the last)?
    bool full_hist = have_hist;
    for (int l = 0; full_hist && l + 1 < nlev; ++l) full_hist = c->hint_P[l] > 0 && c->hint_PZ[l] == c->hint_P[l];
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
        now = level_forms_cost(c, nlev, q0.n_spec, c->spc[q0.defer_root ? 1 : 0].h == p->im_h ? c->spc[q0.defer_root ? 1 : 0].U : 48,
                               q0.pair_mask);
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
      if (p->fixed_num && !(p->reserved & 8) && azk_static_select(c->stream, a)) return AZ_OK;
      azk_static_candidates(c->stream, a); }
    enqueue_select(c, p, nlev, k);
    return AZ_OK;
}

// In the level loop only the device knows a level's row count.  If the previous search on this context forwarded many
// rois at level l, the next one probably does too: its int6 is then sent to both GEMM kernels (rows_hint -1, see
// launch_head).  A wrong guess costs an idle launch, never a result.
static int many_rows_expected(const az_ctx *c, int l)
{
    return (l >= 0 && l < AZ_MAX_LEVELS && c->hint_rows[l] >= c->gemm12_dual_rows &&
            c->gemm12_min_rows < 0x7fffffff) ? -1 : 0;        // (hint_rows: rows of the PASS at that level, speculative rows included)
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
        launch_head(c, fp->full_meta, -1, p->im_h, p->im_w, p->eps, c->zoom_s, c->score_s, c->delta_s, 0.0, false, 1,
                    fp->full_urois, fp->full_ubox, fp->Ufull);
    else if (fused)
        launch_head(c, c->spec_U[defer_root ? 1 : 0], -1, p->im_h, p->im_w, p->eps, c->zoom_s, c->score_s, c->delta_s, 0.0, false,
                    0, c->spec_urois[defer_root ? 1 : 0], nullptr, c->spc[defer_root ? 1 : 0].U);
    else if (n_spec)
        launch_head(c, &c->cnt->specU, -1, p->im_h, p->im_w, p->eps, c->zoom_s, c->score_s, c->delta_s);
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
        a.zoom_s = c->zoom_s; a.score_s = c->score_s; a.delta_s = c->delta_s;
        a.scale = p->scale; a.Tz = p->Tz; a.min_side = p->min_side; a.eps = p->eps; a.dedup = (float)p->dedup;
        a.batch = p->batch_size; a.im_h = p->im_h; a.im_w = p->im_w; a.nlev = nlev; a.n_fused = n_spec;
        a.capR = c->maxR; a.capCh = c->maxCh; a.capCand = c->maxCand;
        a.rois = c->rois; a.urois = c->urois; a.next_dedup = fused_lv ? 1 : 0; a.defer_root = defer_root ? 1 : 0;
        a.spec_next = (plan.pair_mask >> n_spec) & 1; a.choff_next = c->choff_pair; a.crow = c->crow;
        a.spatial_scale = c->spatial_scale;
        a.row_map = full ? fp->spec_map : nullptr; a.root_row = full ? fp->Ufull - 1 : 0;
        a.stab = full ? fp->htab : nullptr; a.stabT = full ? fp->hT : 0;
        a.pred_v = Vp(n_spec); a.score_v = Vs(n_spec); a.zoom_v = Vz(n_spec); a.keep_v = Vk(n_spec); a.key_v = Vy(n_spec);
        azk_spec_levels(s, a);
    }
    bool have_v = full;               // this level's head outputs were looked up among the previous pass's rows (*_v arrays)
    for (int l = fused ? n_spec : 0; l < nlev; ++l) {
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
            a.delta_u = full ? c->delta_s : c->delta_u; a.choff_all = c->choff_pair; a.choff_next = c->choff_pair; a.crow = c->crow;
            a.stab = full ? fp->htab : nullptr; a.stabT = full ? fp->hT : 0; a.root_row_full = full ? fp->Ufull - 1 : 0;
            a.score_all = c->score_s; a.zoom_all = c->zoom_s;
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
            azk_full_lookup(s, Uptr, c->urois, c->ubox, fp->htab, fp->hT, fp->Ufull - 1, c->spatial_scale, c->delta_s, c->score_s,
                            c->zoom_s, p->im_h, p->im_w, p->eps, p->min_side, Vp(l), Vs(l), Vz(l), Vk(l), Vy(l), &c->cnt->err);
            have_v = true;
        }
        if (l < n_spec) {
            Timed t(c, "spec_lookup", l);
            azk_spec_lookup(s, l, Uptr, c->index, c->srcB[cur], c->ubox, c->zoom_s, c->score_s, c->delta_s,
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
    enqueue_select(c, p, nlev, k);
    if (tune && c->pool) {
        Timed t(c, "pool_append", nlev);
        azk_pool_append(s, c->hisZ, &c->cnt->nhis, c->capHis, c->pool, c->pool_n, c->pool_cap);
    }
    return AZ_OK;
}

static int stage_impl(az_ctx *c, void *dst_dev, size_t cap_bytes);

// One search enqueued on THIS context's stream (the public az_propose_launch picks the lane first).
static int launch_impl(az_ctx *c, const az_params *p)
{
    int rc = check_ready(c, true);
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
    if (c->pend.size() >= 2) return fail(c, AZ_ERR_STATE, "az_propose_launch: two searches are already queued, fetch one first");
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
    hipStream_t s = c->stream;
    auto enqueue = [&]() { c->npass = 0; prep_scale(c); return stat ? enqueue_static(c, p, nlev, k) : enqueue_search(c, p, K, nlev, k, tune); };
    // az_set_graphs / AZ_GRAPH=1: capture the launch sequence once per (parameters, feature map) and replay it
    // as a hipGraph.  Every size is read on the device, so the sequence never changes for given parameters.
    if (c->use_graphs < 0) { const char *e = getenv("AZ_GRAPH"); c->use_graphs = (e && atoi(e)) ? 1 : 0; }
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
            it = c->graphs.emplace(key, ent).first;
        }
        c->npass = it->second.npass;
        std::memcpy(c->pass_src, it->second.pass_src, sizeof(c->pass_src));
        HIPCHK(c, hipGraphLaunch(it->second.exec, s));
    } else {
        if ((rc = enqueue()) != AZ_OK) return rc;
    }
    HIPCHK(c, hipGetLastError());
    az_ctx::PendingSearch q;
    q.p = *p; q.nlev = nlev; q.is_static = c->last_static; q.defer = c->last_defer; q.pair_mask = c->last_pair_mask;
    q.full = c->last_full;
    q.npass = c->npass;
    q.feat = c->feat; q.fH = c->d.H; q.fW = c->d.W; q.feat_gen = c->feat_gen;
    q.feat_is_copy = c->feat && (c->feat == c->feat_owned[0] || c->feat == c->feat_owned[1]);
    std::memcpy(q.pass_src, c->pass_src, sizeof(q.pass_src));
    for (q.slot = 0; q.slot < 2 && c->slot_busy[q.slot]; ++q.slot) { }
    if (p->fixed_num) {
        // the result block follows the search's kernels in stream order: whatever is enqueued next (the next image's
        // search, a unit call) finds it already on its way to the host
        HIPCHK(c, hipMemcpyAsync(c->h_res[q.slot], c->cnt, RES_HDR + (size_t)k * 36, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipEventRecord(c->ev_res[q.slot], s));
        q.copied = true;
    }
    c->slot_busy[q.slot] = true;
    c->pend.push_back(q);
    return AZ_OK;
}

int az_set_feature_map_dev_nhwc(az_ctx *c, const float *dev_ptr, int C, int H, int W)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!dev_ptr || C != c->d.C || H <= 0 || W <= 0)
        return fail(c, AZ_ERR_INVALID, "feature map: channel count must match the loaded head");
    c->feat = dev_ptr;                 // already in the layout RoIPool reads: borrowed, no copy
    c->d.H = H; c->d.W = W;
    return AZ_OK;
}


// Collect the result of the search at position `idx` of the pending queue (0 = the oldest; a fallback rerun sits at
// the back) and remove it from the queue.
static int fetch_entry(az_ctx *c, size_t idx, double *boxes_out, float *scores_out, int cap, int *n_out, az_stats *st)
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
        st->n_candidates = h.ytot[nlev];
        st->spec_rows = h.specU;
        st->root_deferred = q.defer;
        st->static_plan = q.is_static;
        st->search_form = q.is_static ? 4 : (q.full == 2 ? 3 : (q.full == 1 ? 2 : (q.pair_mask ? 1 : 0)));
        st->n_reruns = q.reruns;
        const int *hc = reinterpret_cast<const int *>(&h);
        for (int i = 0; i < q.npass && i < AZ_MAX_LEVELS; ++i) {
            const int r = q.pass_src[i] >= 0 ? hc[q.pass_src[i]] : -q.pass_src[i] - 1;
            if (r > 0) st->pass_rows[st->n_passes++] = r;
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
        if (c->pend.size() >= 2) return fail(c, AZ_ERR_STATE, "az_propose_fetch: no room to rerun a search in another form");
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
    }
    const int n = h.nsel;
    // (the candidate list stays readable only while no later search has been queued: it would be overwriting it)
    c->cand_n = c->pend.empty() ? h.ytot[nlev] : -1;
    c->his_n = h.nhis;
    if (st) st->n_proposals = n;
    *n_out = n;
    if (n > cap) return fail(c, AZ_ERR_CAPACITY, "az_propose: output capacity too small");
    std::memcpy(boxes_out, hY, (size_t)n * 4 * sizeof(double));
    if (scores_out) std::memcpy(scores_out, hS, (size_t)n * sizeof(float));
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

// The second lane: an az_ctx with its own stream, staging and per-search buffers that READS this context's weights.
static int ensure_twin(az_ctx *c)
{
    if (c->twin) return AZ_OK;
    if (!c->head_loaded) return fail(c, AZ_ERR_STATE, "two lanes need a loaded head");
    az_ctx *t = nullptr;
    int rc = az_create(c->device, &t);
    if (rc) return fail(c, rc, "az_set_lanes: could not create the second lane");
    t->owner = c;
    t->maxR = c->maxR; t->maxCand = c->maxCand; t->maxCh = c->maxCh;
    auto bail = [&](int code, const char *msg) { c->err = t->err.empty() ? msg : t->err; az_destroy(t); return code; };
    if ((rc = ensure_geom(t)) != AZ_OK) return bail(rc, "second lane: geometry buffers");
    t->d = c->d; t->d.H = t->d.W = 0;
    t->S6 = c->S6; t->S7 = c->S7; t->gemm_parts = c->gemm_parts; t->w6_scale = c->w6_scale; t->spatial_scale = c->spatial_scale;
    t->W6 = c->W6; t->b6 = c->b6; t->W7 = c->W7; t->b7 = c->b7; t->Wt = c->Wt; t->bt = c->bt; t->W6p = c->W6p;
    const size_t R = (size_t)t->maxR;
    const AzHeadDims &d = t->d;
#define A(p, n) if ((rc = dalloc(t, &t->p, (n))) != AZ_OK) return bail(rc, "second lane: head buffers")
    A(pool5, R * d.K6);
    {
        const size_t p6 = (size_t)t->S6 * R * d.n6, p7 = (size_t)t->S7 * R * d.n7;
        A(part, p6 > p7 ? p6 : p7);
    }
    A(h6, R * d.n6); A(h7, R * d.n7);
    if (t->gemm_parts) {
        A(pool5p, (size_t)t->gemm_parts * azk_act_plane_elems((int)R, d.K6)); A(gscale, 4);
        if (hipMemsetAsync(t->pool5p, 0, (size_t)t->gemm_parts * azk_act_plane_elems((int)R, d.K6) * 2, t->stream) != hipSuccess ||
            hipMemsetAsync(t->gscale, 0, 4 * sizeof(float), t->stream) != hipSuccess || hipStreamSynchronize(t->stream) != hipSuccess)
            return bail(AZ_ERR_HIP, "second lane: clearing the operand planes");
    }
#undef A
    t->gemm12_env = c->gemm12_env; t->gemm12_min_rows = c->gemm12_min_rows; t->gemm12_dual_rows = c->gemm12_dual_rows;
    t->head_loaded = true;
    t->profiling = c->profiling; t->use_graphs = c->use_graphs; t->cal = c->cal;
    c->twin = t;
    return AZ_OK;
}

int az_set_lanes(az_ctx *c, int lanes)
{
    if (!c || c->owner || (lanes != 1 && lanes != 2)) return fail(c, AZ_ERR_INVALID, "az_set_lanes: 1 or 2");
    if (!c->lane_order.empty() || !c->pend.empty()) return fail(c, AZ_ERR_STATE, "az_set_lanes: searches are still queued");
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
    return (void *)t->stream;
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
        // the map was set on the context itself: the lane reads it where it lies, behind whatever the context's stream
        // still has to do to it (an un-awaited transpose)
        t->feat = c->feat; t->d.H = c->d.H; t->d.W = c->d.W;
        if (!t->ev_hand) HIPCHK(c, hipEventCreateWithFlags(&t->ev_hand, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(t->ev_hand, c->stream));
        HIPCHK(c, hipStreamWaitEvent(t->stream, t->ev_hand, 0));
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

static int stage_impl(az_ctx *c, void *dst_dev, size_t cap_bytes)
{
    if (!c || c->pend.empty()) return fail(c, AZ_ERR_STATE, "az_propose_stage_result_dev without az_propose_launch");
    az_ctx::PendingSearch &q = c->pend.back();
    if (!q.p.fixed_num) return fail(c, AZ_ERR_STATE, "az_propose_stage_result_dev: fixed proposal count only");
    const size_t bytes = RES_HDR + (size_t)q.p.num_proposals * 36;
    if (!dst_dev || cap_bytes < bytes) return fail(c, AZ_ERR_INVALID, "az_propose_stage_result_dev: destination too small");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst_dev, c->cnt, bytes, hipMemcpyDeviceToDevice, c->stream));
    // az_propose_fetch waits for the slot's event: recorded again HERE, behind the staging copy, so that "the record is
    // staged when az_propose_fetch returns" holds (the launch recorded it behind the host copy only)
    if (q.copied) HIPCHK(c, hipEventRecord(c->ev_res[q.slot], c->stream));
    q.stage_dst = dst_dev; q.stage_cap = cap_bytes;
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
    if (c->cand_n < 0)
        return fail(c, AZ_ERR_STATE, "az_last_candidates: no fetched search, or a later call reused the candidate buffers");
    const int n = c->cand_n;
    *n_out = n;
    if (n > cap) return fail(c, AZ_ERR_CAPACITY, "az_last_candidates: cap too small");
    if (boxes_out) HIPCHK(c, hipMemcpy(boxes_out, c->Yall, (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost));
    if (scores_out) HIPCHK(c, hipMemcpy(scores_out, c->Sall, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return AZ_OK;
}

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
        if (W * sizeof(unsigned long long) > 60000) return fail(c, AZ_ERR_CAPACITY, "az_nms: n too large");
        HIPCHK(c, hipMalloc((void **)&c->nms_dets, (size_t)cap * 5 * 4));
        HIPCHK(c, hipMalloc((void **)&c->nms_sdets, (size_t)cap * 5 * 4));
        HIPCHK(c, hipMalloc((void **)&c->nms_order, (size_t)cap * 4 + 16));
        HIPCHK(c, hipMalloc((void **)&c->nms_mask, (size_t)cap * W * 8));
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
        azk_nms(s, c->nms_dets, n, thresh, c->nms_order, c->nms_sdets, c->nms_mask, (unsigned long long *)c->nms_rank, hk, (int *)c->h_nmsg, tag);
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
      azk_nms(s, c->nms_dets, n, thresh, c->nms_order, c->nms_sdets, c->nms_mask, (unsigned long long *)c->nms_rank, c->nms_keep, nk); }
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
        // (grow: nothing may still be reading the old slots)
        for (int i = 0; i < 2; ++i) if (c->io_ev[i]) HIPCHK(c, hipEventSynchronize(c->io_ev[i]));
        for (int i = 0; i < 2; ++i) {
            if (c->io_host[i]) hipHostFree(c->io_host[i]);
            if (c->io_dev[i]) hipFree(c->io_dev[i]);
            c->io_host[i] = nullptr; c->io_dev[i] = nullptr;
        }
        c->io_cap = 0;
        const size_t cap = nin + nin / 4 + 256;
        for (int i = 0; i < 2; ++i) {
            HIPCHK(c, hipHostMalloc((void **)&c->io_host[i], cap));
            HIPCHK(c, hipMalloc((void **)&c->io_dev[i], cap));
            if (!c->io_ev[i]) HIPCHK(c, hipEventCreateWithFlags(&c->io_ev[i], hipEventDisableTiming));
        }
        c->io_cap = cap;
    }
    const int t = c->io_turn;
    c->io_turn ^= 1;
    HIPCHK(c, hipEventSynchronize(c->io_ev[t]));            // (the slot's previous image: two uploads ago, long done)
    std::memcpy(c->io_host[t], im, nin);                    // the caller's array may go away as soon as this returns
    HIPCHK(c, hipMemcpyAsync(c->io_dev[t], c->io_host[t], nin, hipMemcpyHostToDevice, s));
    azk_image_blob(s, c->io_dev[t], h, w, means, 1.0 / scale, 1.0 / scale, oh, ow, blob_dev);
    HIPCHK(c, hipEventRecord(c->io_ev[t], s));
    HIPCHK(c, hipGetLastError());
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
    if (on & 8) {
        // every slot starts as (first-in = ~0, last-out = 0); a slot is used once between two calls of this function
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(c->stream));
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
    if (c->twin) az_set_profiling(c->twin, on);
    return AZ_OK;
}

int az_last_kernel_times(az_ctx *c, char *names_out, float *ms_out, int32_t *level_out, int cap, int *n_out)
{
    if (!c || !n_out) return AZ_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
