// az_ctx.h -- the context behind the C ABI (include/aznet_hip.h) and the host-side helpers its translation units share:
// az_capi.hip (lifecycle, head, maps, lanes, public launch / fetch, measurement, exchange), az_search.hip (the forms of a
// search: plans, caches, cost model, launch sequence, collecting a result), az_units.hip (unit entry points, detection head,
// NMS, tuner, recall, front-end).  Helpers are internal (anonymous namespace: one copy per translation unit).
#pragma once
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
    // [H][W][C] copies of NCHW maps, three of them used in turn (two until round 5): the map of a search that is still queued (and might have to
    // be run again in another form) survives the hand-over of the next image's map
    static constexpr int NFEAT = 3;      // (one per search a lane may have queued: AZ_QUEUE_MAX)
    static constexpr int AZ_QUEUE_MAX = 3;
    float *feat_owned[NFEAT] = {nullptr, nullptr, nullptr};
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
    // (round 5) A dataset is a stream of DIFFERENT images: the tree of the previous image is one sample of what the next one
    // looks like, not a prediction of it.  Each shape keeps its last HINT_K level-loop searches (newest = the hint_* fields
    // above, then hint_old[0..]); the choices that cost little when wrong and gain little when right are made on the
    // EXPECTED cost over those (pair_plan, full_prepare), the ones that cost a second search when wrong need a streak
    // (tree rows of the whole-tree pass: the last two searches of the shape both walked the full tree).
    static constexpr int HINT_K = 4;
    struct HintRec { int rows[AZ_MAX_LEVELS], P[AZ_MAX_LEVELS], PZ[AZ_MAX_LEVELS], U[AZ_MAX_LEVELS], SPN[AZ_MAX_LEVELS]; };
    HintRec hint_old[HINT_K - 1];
    int hint_n = 0;                           // records held for the shape in the working fields (0: none, 1: hint_* only, ...)
    int hint_full_streak = 0;                 // consecutive searches of the shape, up to the last, that walked the FULL tree
    struct ShapeHint { int h, w, nlev; int rows[AZ_MAX_LEVELS], P[AZ_MAX_LEVELS], PZ[AZ_MAX_LEVELS], U[AZ_MAX_LEVELS], SPN[AZ_MAX_LEVELS];
                       HintRec old[HINT_K - 1]; int n, full_streak;
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
    // Two stages (round 5).  A search whose ONE head pass does not depend on its own geometry -- the whole-tree / closure
    // pass, the one-pass plan: its rows are a function of the image shape -- is enqueued on two streams: stage 1 = RoIPool +
    // int6 + slab sum on `stream`, stage 2 = int7 + heads + every geometry kernel + the selection + the result copy on
    // `stream2`, behind an event.  The next search's stage 1 then does not queue behind this one's single-workgroup geometry
    // kernels (which cannot get a CU while a lane's int6 holds them all), and this one's MFMA-bound int7 may run beside the
    // next int6's HBM-bound neighbours (the other image's slab sum and RoIPool) instead of in a slot of its own.
    //   h6 (slab sum -> int7) is the one buffer both stages touch: ev_h6 orders int7 behind the slab sum, ev_i7 the NEXT
    //   slab sum behind this int7; int7's slabs live in `part7`, not in `part`.
    //   ev_s2 = everything enqueued on stream2 so far: whatever is not stage 1 of another two-stage search (a search in
    //   another form, a plan builder, a unit entry point) makes `stream` wait for it first (join_s2).
    //   The whole-tree / closure form goes one step further (three stages): its geometry kernels + selection + result copy
    //   run on a THIRD stream behind the heads (ev_tail).  They are single-workgroup kernels that get a CU only between two
    //   int6 launches; on the second stream they held the NEXT search's int7 back for ~70 us.  The head outputs they read
    //   (zoom_s / score_s / delta_s) exist twice, used in turn (out_par); ev_geo[p] = the geometry that read set p is done.
    hipStream_t stream2 = nullptr, stream3 = nullptr;
    hipEvent_t ev_tail = nullptr, ev_s3 = nullptr, ev_geo[2] = {nullptr, nullptr};
    bool s3_live = false, g_live[2] = {false, false};
    int out_par = 0, three_now = 0;
    float *zoom_s2 = nullptr, *score_s2 = nullptr, *delta_s2 = nullptr;
    hipStream_t gs = nullptr;                 // while a search is being enqueued: where the kernels behind its head pass go (nullptr: `stream`)
    hipStream_t ts = nullptr;                 // ... and where profiling events are recorded (nullptr: `stream`)
    hipStream_t last_s = nullptr;             // the stream the search launched last ends on
    hipEvent_t ev_h6 = nullptr, ev_i7 = nullptr, ev_s2 = nullptr;
    bool i7_live = false, s2_live = false;
    int split_env = -1, split_now = 0, async_err = 0;
    float *part7 = nullptr;                   // int7's split-K slabs [S7][maxR][n7]
    // Staged searches write int7's slabs into THREE buffers in turn (ring[0] = part7): the heads of search i - 3 -- the last
    // reader of the buffer search i writes -- ran before that search was fetched, and a lane never holds more than
    // AZ_QUEUE_MAX = 3 unfetched searches, so int7 needs no event wait for the previous search's heads any more (each wait on
    // the main stream is a ~6 us bubble between two chip-wide kernels).  AZ_P7_RING=0: one buffer + the wait (measurements).
    static_assert(AZ_QUEUE_MAX <= 3, "the int7 slab ring has one buffer per search a lane may hold unfetched");
    float *part7_ring[3] = {nullptr, nullptr, nullptr};
    int part7_turn = 0, part7_ring_env = -1;
    az_ctx *twin = nullptr, *owner = nullptr;
    // A batch of images searched in lockstep (az_batch_launch; az_search.hip: batch_launch_impl, az_batch.hip).  Held by the
    // lane whose head buffers the batch's passes run in (the context, or its twin for every other batch when two lanes are
    // on); every image of the batch has a SLOT: an az_ctx of its own for the tree (regions, counters, candidates, result
    // slots, history), created without head buffers -- it gets them if one of its searches ever has to be run again alone.
    struct Batch {
        std::vector<az_ctx *> slots;
        int *off = nullptr;                   // device: AzGatherArgs::off_out of the pass being enqueued, [AZ_BATCH_MAX + 2]
        float *rois_cat = nullptr;            // the pass's rois / anchors / map table
        double *ubox_cat = nullptr;
        const float **feats = nullptr;
        int *feat_hw = nullptr, *row_hw = nullptr;   // every image's map size [AZ_BATCH_MAX][2]; every row's image size [maxR][2]
        unsigned char *args_dev = nullptr, *args_host = nullptr;   // the geometry kernels' arguments, one block per image and launch
        size_t args_cap = 0;
        // the images' result blocks (counters + selected boxes and scores) lie side by side, on the device and in pinned host
        // memory: ONE device-to-host copy per batch (eight copies of 12 KB cost 75 us of stream time, one of 95 KB ~10)
        unsigned char *res_dev = nullptr, *res_host = nullptr;
        int gemm12_rows = -1;                 // rows of a pass from which int6 takes the many-row kernel (AZ_BATCH_GEMM12_ROWS)
        int n_live = 0, next_fetch = 0;       // images of the batch in flight; the one az_batch_fetch returns next
        bool lockstep = false;                // false: the batch's images were launched one after the other on their slots
        int rows_hint[AZ_MAX_LEVELS] = {0};   // rows of the passes of the last fetched batch (which int6 kernel takes a level)
        int rows_acc[AZ_MAX_LEVELS] = {0};
        int hint_n = 0;                       // images of the batch rows_hint was taken from (0: none yet)
    } bsets[2];                               // two sets per lane, used in turn: the host enqueues a lane's next batch while its
                                              // current one runs (same stream: the two do not interleave on the GPU)
    int bset_turn = 0;
    std::deque<int> batch_order;              // (owner) lane | set << 1 of the batches in flight, oldest first
    int batch_next = 0;
    bool head_bufs = true;                    // false: a batch slot that has not needed pool5 / slabs / h6 / h7 yet
    // A batch slot that searches on its own (its batch's pass overflowed, its shape is not taken in lockstep) does not get a
    // head set of its own -- at max_regions 16384 and the VGG16 dims that is ~8.5 GB, times 32 slots, times two sets per
    // lane -- but takes the OWNER's one spare set in turn: `ev` (recorded behind each such search) orders the next user's
    // stream behind the previous one.  The searches were chip-wide one after the other anyway.
    struct SpareHead {
        float *pool5 = nullptr, *part = nullptr, *h6 = nullptr, *h7 = nullptr, *part7 = nullptr, *gscale = nullptr;
        unsigned short *pool5p = nullptr;
        bool ready = false, ev_live = false;
        hipEvent_t ev = nullptr;
    } spare;                                  // (owner)
    bool head_shared = false;                 // (a batch slot) pool5 / part / h6 / h7 / part7 are the owner's spare set
    Batch *batch_set = nullptr;               // (a batch slot) the batch set it belongs to
    // (a batch slot) its counters / result block and its first host result slot are slices of the lane's arenas
    // (Batch::res_dev / res_host); the slot's own allocations stay in its lists and are freed with it
    unsigned char *h_res_own0 = nullptr;
    int lanes = 1, lane_next = 0, last_fetch_lane = 0;
    std::deque<int> lane_order;               // lanes of the searches launched through the public entry points, oldest first
    hipEvent_t ev_hand = nullptr;             // (in a twin) orders the lane behind the owner's stream when it reads the owner's map
    // (in a twin) recorded behind the lane's private copy of an owner-held map: the owner's stream waits for it before it
    // writes its channel-last copies again (the lane may still be busy with an earlier search when the owner is handed
    // the map after next)
    hipEvent_t ev_copy = nullptr;
    bool ev_copy_live = false;
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
    // upload slots of az_image_blob_dev_on (pinned host + device staging + "slot free" event).  Two to start with; a slot
    // whose previous upload is still queued on the GPU (a lockstep batch's uploads wait behind the previous batch's search)
    // makes the ring GROW, up to IO_SLOTS_MAX, instead of making the host wait.
    static constexpr int IO_SLOTS_MAX = 40;
    struct IoSlot { unsigned char *host = nullptr, *dev = nullptr; hipEvent_t ev = nullptr; };
    std::vector<IoSlot> io;
    size_t io_cap = 0;
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
        int cut = 0;                        // > 0: the search was enqueued up to (not including) this level
        int pass_src[AZ_MAX_LEVELS + 2] = {0};
        int pass_lv[AZ_MAX_LEVELS + 2] = {0};   // the `level` each head pass was launched for (-1: speculative / whole-tree / one-pass)
        void *stage_dst = nullptr;          // az_propose_stage_result_dev target
        size_t stage_cap = 0;
        int slot = 0;
        hipStream_t last_s = nullptr;       // the stream the search's last kernels and its result copy are on
        bool copied = false;                // result block already on its way to h_res[slot]
        const float *feat = nullptr;        // the map the search reads (a rerun in another form needs it again)
        int fH = 0, fW = 0;
        unsigned feat_gen = 0;
        bool feat_is_copy = false;          // `feat` is one of the ctx's own channel-last copies (gone if they are reallocated)
        int batch = 0;                      // 1: an image of a lockstep batch (az_batch_launch)
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
    struct GraphEntry { hipGraphExec_t exec; int npass; int pass_src[AZ_MAX_LEVELS + 2]; int pass_lv[AZ_MAX_LEVELS + 2]; };
    std::map<std::string, GraphEntry> graphs;        // captured launch sequences (az_set_graphs)
    int use_graphs = -1;                             // -1: take the AZ_GRAPH environment variable
    int last_defer = 0;
    // Early end: a tree whose previous search of the shape had no regions from some level on is enqueued only up to that
    // level (the passes and geometry kernels of an empty level cost ~35 us each, a fifth of a sparse search); the last
    // geometry kernel checks -- regions after all set err bit 1024 and az_propose_fetch runs the search again in full
    // (params.reserved bit 12 / AZ_EARLY_END=0: never).  A miss costs more than a hit saves (a second search against two
    // empty levels), so the cut is taken only when the context's last FOUR level-loop searches all ended at or before the
    // level in question: early_hist holds the first empty level of the last eight (4 bits each, 15 = none).
    int last_cut = 0, cut_env = -1;
    unsigned early_hist = 0xFFFFFFFFu;
    int n_hist = 0;                     // level-loop searches this context has recorded in early_hist (saturating)
    // head passes of the search being enqueued / last launched: where each one's row count lives
    // (>= 0: int index into AzCounts; < 0: -(rows + 1), a count the host knows)
    int npass = 0;
    int pass_src[AZ_MAX_LEVELS + 2] = {0};
    int pass_lv[AZ_MAX_LEVELS + 2] = {0};
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
    A(zoom_s2, R); A(score_s2, R * AZ_NSUB); A(delta_s2, R * 4 * AZ_NSUB);
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
        if (hipEventRecord(a, c->ts ? c->ts : c->stream) != hipSuccess) { hipEventDestroy(a); hipEventDestroy(b); on = false; ++c->event_errors; }
    }
    ~Timed()
    {
        if (trace()) { const hipError_t e = hipStreamSynchronize(c->ts ? c->ts : c->stream); fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e)); }
        if (!on) return;
        if (hipEventRecord(b, c->ts ? c->ts : c->stream) != hipSuccess) { hipEventDestroy(a); hipEventDestroy(b); ++c->event_errors; return; }
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
        c->pass_lv[c->npass] = level;
        c->pass_src[c->npass++] = in_cnt ? (int)(Uptr - c0) : -(rows_hint > 0 ? rows_hint : 0) - 1;
    }
    if (c->gemm12_env < 0) {            // AZ_GEMM12_MIN=<rows> (0: never): measurements
        const char *f = getenv("AZ_GEMM12_MIN");
        if (f) c->gemm12_min_rows = atoi(f) > 0 ? atoi(f) : 0x7fffffff;
        c->gemm12_env = 1;
    }
    const bool split = c->split_now && c->stream2 && c->ev_h6 && c->ev_i7;
    hipStream_t s2 = split ? c->stream2 : c->stream;
    // Stage 1 of a staged search waits for the int7 of the previous one (ev_i7) -- RoIPool included: beside int7 the
    // HBM-bound RoIPool of the next image costs that int7 40 us (70 -> 110) to hide 27 of its own, and the int6 behind it
    // starts later for it.  Measured on one lane, three searches queued: 1.181 -> 1.128 ms per image (two queued: 1.135 ->
    // 1.141: there the host's launch latency did the same by accident).  AZ_ROI_AFTER_I7=0: RoIPool does not wait (the
    // round-5 first version; measurements).  The same order across the two lanes of a context (a lane's many-row RoIPool
    // behind the OTHER lane's int7, an event both ways) was measured too: 1.114 -> 1.138 ms, not kept.
    static const int roi_after_i7 = getenv("AZ_ROI_AFTER_I7") ? atoi(getenv("AZ_ROI_AFTER_I7")) : 1;
    if (roi_after_i7 && c->i7_live) { if (hipStreamWaitEvent(c->stream, c->ev_i7, 0) != hipSuccess) c->async_err = 1; c->i7_live = false; }
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
    // int6 waits for the int7 of the previous two-stage search of this context (as the RoIPool above by default): h6, which this
    // pass's slab sum writes, is that int7's operand -- and an int6 that starts while int7's workgroups still hold CUs runs
    // with stragglers to its end (one persistent workgroup per CU, work dealt statically: measured 1.13 -> 1.24 ms per image
    // when the two overlapped)
    if (c->i7_live) { if (hipStreamWaitEvent(c->stream, c->ev_i7, 0) != hipSuccess) c->async_err = 1; c->i7_live = false; }
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
    // int7 stays in stage 1 -- RoIPool, int6, slab sum, int7 back to back on `stream`: the chip-wide kernels of consecutive
    // images follow each other without an event hop (each ~13 us on the critical path: slab sum -> int7 on the second
    // stream, int7 -> the next image's RoIPool back on the first); stage 2 = the heads (+ geometry, or stage 3).  int7 writes
    // its slabs where the previous search's heads read theirs: it waits for that search's stage 2, which ended ~1000 us
    // before.  One lane, three queued: 1.130-1.135 -> 1.116 ms per image.  AZ_I7_STAGE1=0: int7 in stage 2 (measurements).
    static const int i7s1 = getenv("AZ_I7_STAGE1") ? atoi(getenv("AZ_I7_STAGE1")) : 1;
    const bool i7_first = split && i7s1 && c->part7;
    if (split && !i7_first) {
        // stage 2 from here on: int7 behind the slab sum, everything the caller enqueues behind this pass behind int7
        if (hipEventRecord(c->ev_h6, c->stream) != hipSuccess || hipStreamWaitEvent(s2, c->ev_h6, 0) != hipSuccess) c->async_err = 1;
        c->gs = s2; c->ts = s2;
    }
    float *p7 = c->part7 ? c->part7 : c->part;
    bool ring = false;
    if (i7_first) {
        if (c->part7_ring_env < 0) { const char *e = getenv("AZ_P7_RING"); c->part7_ring_env = (e && !atoi(e)) ? 0 : 1; }
        if (c->part7_ring_env && !c->part7_ring[1]) {
            c->part7_ring[0] = c->part7;
            for (int q = 1; q < 3 && c->part7_ring_env; ++q)
                if (dalloc(c, &c->part7_ring[q], (size_t)c->S7 * c->maxR * d.n7) != AZ_OK) { c->err.clear(); (void)hipGetLastError(); c->part7_ring_env = 0; }
            if (!c->part7_ring_env) c->part7_ring[1] = c->part7_ring[2] = nullptr;     // (no room: one buffer + the wait)
        }
        if (c->part7_ring_env && c->part7_ring[2] && c->part7_ring[0] == c->part7) {
            p7 = c->part7_ring[c->part7_turn];
            c->part7_turn = (c->part7_turn + 1) % 3;
            ring = true;
        }
    }
    unsigned long long *ts7 = span_slot("fc7_gemm");
    if (ts7) c->profiling &= ~(1 | 2);
    if (i7_first && !ring && c->s2_live && hipStreamWaitEvent(c->stream, c->ev_s2, 0) != hipSuccess) c->async_err = 1;
    if (i7_first && c->s3_live && !c->three_now && hipStreamWaitEvent(c->stream, c->ev_s3, 0) != hipSuccess) c->async_err = 1;
    { Timed t(c, "fc7_gemm", level, 1);
      azk_fc_gemm(i7_first ? c->stream : s2, c->h6, d.n6, c->W7, d.n6, Uptr, c->maxR, d.n7, d.n6, c->S7, p7, 1 << 30, ts7); }
    c->profiling = prof_keep;
    if (i7_first) {
        // stage 2 from here on: the heads behind int7
        if (hipEventRecord(c->ev_h6, c->stream) != hipSuccess || hipStreamWaitEvent(s2, c->ev_h6, 0) != hipSuccess) c->async_err = 1;
        c->gs = s2; c->ts = s2;
    } else if (split) { if (hipEventRecord(c->ev_i7, s2) != hipSuccess) c->async_err = 1; c->i7_live = true; }
    const bool three = split && c->three_now && c->stream3 && c->ev_tail;
    // (three stages: the heads write the output set the geometry of the search before the previous one read)
    if (three && c->g_live[c->out_par]) {
        if (hipStreamWaitEvent(s2, c->ev_geo[c->out_par], 0) != hipSuccess) c->async_err = 1;
        c->g_live[c->out_par] = false;
    }
    { Timed t(c, "tail", level);       // (finishes int7 as well: slab sum + bias + ReLU while staging its rows)
      azk_tail(s2, p7, c->S7, c->b7, d.n7, c->Wt, c->bt, ubox ? ubox : c->ubox, Uptr, c->maxR, im_h, im_w,
               eps, zoom, score, delta, c->pred_u, keep_flags ? c->keep_u : nullptr, min_side,
               (keep_flags && keys) ? c->key_u : nullptr); }
    if (three) {
        // stage 3 from here on: whatever the caller enqueues behind this pass goes to the third stream, behind the heads
        if (hipEventRecord(c->ev_tail, s2) != hipSuccess || hipStreamWaitEvent(c->stream3, c->ev_tail, 0) != hipSuccess) c->async_err = 1;
        c->gs = c->stream3; c->ts = c->stream3;
    }
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

// `stream` waits for everything the context has enqueued on its second stream (stage 2 of two-stage searches): called by
// whatever touches the per-search buffers and is not stage 1 of another two-stage search.
void join_s2(az_ctx *c)
{
    if (c && c->s2_live && c->stream2 && c->ev_s2) {
        if (hipStreamWaitEvent(c->stream, c->ev_s2, 0) != hipSuccess) c->async_err = 1;
        c->i7_live = false;                       // (ev_i7 lies before ev_s2 on that stream)
    }
    if (c && c->s3_live && c->stream3 && c->ev_s3 && hipStreamWaitEvent(c->stream, c->ev_s3, 0) != hipSuccess) c->async_err = 1;
}

int check_geom(az_ctx *c)
{
    if (!c) return AZ_ERR_INVALID;
    join_s2(c);
    return ensure_geom(c);
}

int check_ready(az_ctx *c, bool need_feat, bool join = true)
{
    if (!c) return AZ_ERR_INVALID;
    if (!c->head_loaded) return fail(c, AZ_ERR_STATE, "az_load_head has not been called");
    if (need_feat && !c->feat) return fail(c, AZ_ERR_STATE, "no feature map set");
    if (join) join_s2(c);
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

// ---- az_search.hip ------------------------------------------------------------------------------------------------------
// one search enqueued on THIS context's stream / its result collected (idx: position in c->pend) / its record staged
int launch_impl(az_ctx *c, const az_params *p);
int fetch_entry(az_ctx *c, size_t idx, double *boxes_out, float *scores_out, int cap, int *n_out, az_stats *st);
int stage_impl(az_ctx *c, void *dst_dev, size_t cap_bytes);
// n images (one shape or several, the same number of levels) in lockstep on lane L (whose head buffers the passes use), image b
// on slots[b] with parameters params[b] and map maps[b] of Hs[b] x Ws[b] cells;
// AZ_ERR_STATE + *not_taken = 1: this shape / these settings do not take the lockstep form (nothing enqueued)
int batch_launch_impl(az_ctx *L, az_ctx::Batch &B, int n, az_ctx **slots, const az_params *params /* [n] */, const float *const *maps,
                      const int *Hs, const int *Ws, int *not_taken);
// ---- az_capi.hip --------------------------------------------------------------------------------------------------------
int set_feature_map_common(az_ctx *c, const float *src, bool src_is_host, int C, int H, int W, bool wait = true);
int ensure_lane_head(az_ctx *t);          // the head buffers of a lane / batch slot created without them
