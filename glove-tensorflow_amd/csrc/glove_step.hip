// GloVe training step on gfx950: fused forward + gradient passes and the optimizer applies.
//
// What the kernels replace (reference = /root/reference, TF graph built by model_fn):
//   rowpass   ResourceGather x4 + dot + Add (src/models/model_utils.py:41-54), weighted MSE head
//             (src/models/estimator.py:48-56), activity-L2 losses (model_utils.py:18-21,52) and
//             the row-side half of tf.gradients.
//   colpass   the col-side half of tf.gradients.
//   apply     OptimizerV2 dedup (Unique + UnsortedSegmentSum) + ResourceSparseApplyAdagradV2.
//   dense_*   the same sums written to dense [V,d] buffers (DP all-reduce), ResourceApplyAdagradV2
//             and Keras-legacy Adam (whole-table decay).
//
// Memory-bound gather/scatter: no MFMA.  A row of d floats is spread over LPR lanes as float4;
// a wave64 therefore works on 64/LPR chunks at once.  All sums have a fixed order, so a step is
// bitwise repeatable for a given plan.
#include "glove_common.h"

namespace glove {

#ifdef GLOVE_STAMPS
__device__ unsigned long long *g_stamps = nullptr;
#endif

template <int LPR, int NV>
__device__ inline void load_row(f4 (&dst)[NV], const float *table, int32_t id, int d4, int lg)
{
    // branch-free: lanes past the row end re-read its last float4 and are zeroed by a select
    // (a predicated load makes hipcc branch around every row load and serialise its waits)
    const f4 *p = reinterpret_cast<const f4 *>(table) + (size_t)id * d4;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i4 = lg + k * LPR;
        const f4 v = p[i4 < d4 ? i4 : d4 - 1];
        dst[k] = (i4 < d4) ? v : f4{0.f, 0.f, 0.f, 0.f};
    }
}

// Gather-pass row load: 32-bit byte offsets off a uniform base (tables < 4 GiB, checked on the
// host) so the compiler can use the SGPR-base + VGPR-offset load form instead of 64-bit pointer
// arithmetic per row; FULL (d4 == LPR*NV) drops the row-end predicate altogether, otherwise only the
// last float4 slot of a lane can lie past the row end.
template <int LPR, int NV, bool FULL>
__device__ inline void load_row_fast(f4 (&dst)[NV], const float *table, int32_t id, int d4, int lg)
{
    const char *base = reinterpret_cast<const char *>(table);
    const uint32_t row_bytes = FULL ? (uint32_t)(LPR * NV * 16) : (uint32_t)d4 * 16u;
    const uint32_t off = (uint32_t)id * row_bytes + (uint32_t)lg * 16u;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (FULL || k < NV - 1) {
            dst[k] = *reinterpret_cast<const f4 *>(base + (off + (uint32_t)(k * LPR * 16)));
        } else {
            const int i4 = lg + k * LPR;
            const uint32_t o = (uint32_t)id * row_bytes + (uint32_t)(i4 < d4 ? i4 : d4 - 1) * 16u;
            const f4 v = *reinterpret_cast<const f4 *>(base + o);
            dst[k] = (i4 < d4) ? v : f4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

template <int LPR, int NV>
__device__ inline void store_row(float *table, size_t row_index, int d4, int lg, const f4 (&src)[NV])
{
    f4 *p = reinterpret_cast<f4 *>(table) + row_index * d4;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i4 = lg + k * LPR;
        if (i4 < d4) p[i4] = src[k];
    }
}

// ---- cache policy of the run-merged (fused) passes ----------------------------------------------------------------------
// A fused pass is bound by the bytes its partner gathers miss the XCD's 4 MB L2 with: the L2 is worth what it
// holds of the Zipf head of the partner table, and everything else that streams through it — the own rows and accumulator rows,
// read once, and the finished rows and accumulators, written once and read by nobody in this launch — pushes those rows out.
// So the streams are marked: own rows and accumulators are loaded non-temporal (`nt`: served by the L2, first in line to be
// replaced), finished rows are stored write-through (`sc1`: the line does not stay in the L2; MI355X_MICROARCH.md, stores of
// each flavour).  Same bits; one process, same resident plans, twin form, us per step (tools/ab_kernels.py,
// profiles/r05_exp_cache_policy_fused_passes.txt):
//                                   plain    sc1 stores   nt loads   both
//   V = 400 k, d = 300, B = 1 M     585.1      582.1       575.8     571.2
//   V = 2 M,   d = 128, B = 1 M     377.0      372.6       372.2     368.0
// (nt on the pair fields and chunk descriptors as well: 571.6 / 369.9 — nothing, not taken; nt on the partner gathers of cold
// ids lost in round 3: DESIGN_APPENDIX.md)
template <int LPR, int NV>
__device__ inline void load_row_nt(f4 (&dst)[NV], const float *table, int32_t id, int d4, int lg)
{
    const f4 *p = reinterpret_cast<const f4 *>(table) + (size_t)id * d4;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i4 = lg + k * LPR;
        const f4 v = __builtin_nontemporal_load(p + (i4 < d4 ? i4 : d4 - 1));
        dst[k] = (i4 < d4) ? v : f4{0.f, 0.f, 0.f, 0.f};
    }
}

template <int LPR, int NV>
__device__ inline void store_row_wt(float *table, size_t row_index, int d4, int lg, const f4 (&src)[NV])
{
    f4 *p = reinterpret_cast<f4 *>(table) + row_index * d4;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i4 = lg + k * LPR;
        // (a store the compiler does not count in vmcnt: its own waits can only become longer by it, never shorter)
        if (i4 < d4) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p + i4), "v"(src[k]) : "memory");
    }
}

// ---- optimizer arithmetic (Keras-legacy forms) -------------------------
// x / (sqrt(a) + eps) uses the hardware v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the IEEE
// expansions: ~3 ulp on the update term, far inside the 1e-5 parity tolerance, and a third of the
// instructions and registers of the apply kernels.
__device__ inline float inv_sqrt_eps(float a, float eps) { return __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(a) + eps); }

__device__ inline void adagrad_elem(float &Wv, float &A, float g, float lr, float eps)
{
    A += g * g;
    Wv -= lr * g * inv_sqrt_eps(A, eps);
}

__device__ inline void adagrad_vec(f4 &Wv, f4 &A, const f4 g, float lr, float eps)
{
    A += g * g;
    const f4 inv = f4{inv_sqrt_eps(A.x, eps), inv_sqrt_eps(A.y, eps), inv_sqrt_eps(A.z, eps), inv_sqrt_eps(A.w, eps)};
    Wv -= (lr * g) * inv;
}

// Adam's update has two products feeding one sum, which the compiler may contract either way: the rounding
// sequence is spelled out so that every kernel that applies it (dense sweep, fused step) gives the same bits.
__device__ inline void adam_elem(float &Wv, float &M, float &Vv, float g, float lr_t, float b1, float b2, float eps)
{
#pragma clang fp contract(off)
    const float g1 = (1.0f - b1) * g;
    const float g2 = ((1.0f - b2) * g) * g;
    M = __builtin_fmaf(b1, M, g1);
    Vv = __builtin_fmaf(b2, Vv, g2);
    const float step = lr_t * M;
    Wv = __builtin_fmaf(-step, inv_sqrt_eps(Vv, eps), Wv);
}

__device__ inline void adam_vec(f4 &Wv, f4 &M, f4 &Vv, const f4 g, float lr_t, float b1, float b2, float eps)
{
    float w[4] = {Wv.x, Wv.y, Wv.z, Wv.w}, m[4] = {M.x, M.y, M.z, M.w}, v[4] = {Vv.x, Vv.y, Vv.z, Vv.w};
    const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) adam_elem(w[i], m[i], v[i], gg[i], lr_t, b1, b2, eps);
    Wv = f4{w[0], w[1], w[2], w[3]};
    M = f4{m[0], m[1], m[2], m[3]};
    Vv = f4{v[0], v[1], v[2], v[3]};
}

__device__ inline float adam_lr_t(float lr, double ln_beta1, double ln_beta2, int64_t step)
{
    // 1 - beta^t = -expm1(t ln beta), logs from the host in fp64 (two fp64 pow per thread were a third of the dense sweep)
    const double t = (double)step;
    return lr * sqrtf(-expm1f((float)(t * ln_beta2))) / -expm1f((float)(t * ln_beta1));
}

// ------------------------------------------------------------------------------------------
// Gather passes: one GROUP of LPR lanes per chunk, 64/LPR chunks per wave, BOTH sides in one launch.
//
// The two sides are symmetric and independent.  A row-side chunk (row u, pairs i) gathers the
// partner rows C[col_i]; a col-side chunk (col v, pairs i) gathers R[row_i].  Either side forms
// p_i = own . partner + own_bias + partner_bias + g and e_i = 2 w_i (p_i - y_i) / B by itself (the plan
// carries w, y in both orders), so the col side needs nothing the row side produces: the first
// blocks of the grid take the row chunks, the rest the col chunks, and a step is TWO dependent kernels
// (passes -> apply) instead of three.  The second dot product per pair is cheap next to a kernel
// boundary in the latency-bound regime (B = 131072, d = 64: 9.7 + 7.5 us as two kernels, ~10 us fused).
//
// The step is latency-bound at realistic batch sizes (a few pairs per lane over the whole chip), so
// a chunk is a short dependent chain  descriptor -> pair fields -> partner rows:
//   * the pair fields of the WHOLE chunk are fetched up front (lane t takes pairs t, t+LPR, ...,
//     coalesced) and staged in LDS; every trip reads the fields of its U pairs back with
//     ds_read_b128 (a broadcast inside the group);
//   * after that every trip costs one memory round trip with U partner rows in flight per group;
//   * groups are 8 lanes wide at d <= 64 (8 lanes x 2 float4 = one 128-B line per load), so a wave
//     carries 8 chunks and the per-pair bookkeeping (addresses, DPP butterfly, diff, e) is issued
//     once for 8 pairs;
//   * chunks are dealt to groups round-robin over the side's blocks so the full-length chunks of the
//     Zipf head do not pile up in a few workgroups.
// Measured in-process (tools/ab_kernels.py) against earlier forms: a wave per chunk (-40 %), per-pair
// ds_bpermute broadcasts on 16-lane groups (-7 % at B = 131072, -20 % at B = 1048576, d = 64).
// kChunkMax bounds chunk_cap.
// ------------------------------------------------------------------------------------------
constexpr int kChunkMax = 32;
constexpr int kFieldStride = kChunkMax + 4;      // dwords; +16 B staggers the groups over the LDS banks

// partner rows in flight per group, and the occupancy the register allocator is held to.  With both sides
// in one launch the grid at B = 131072, d = 64 is ~3700 waves: U = 4 + 4 waves/SIMD keeps all of them
// resident at once (A/B in-process: 12.1 us vs 13.8 us for U = 8 at 2 waves/SIMD; 42.9 vs 45.5 us at B = 1 M).
__device__ inline int grp_of(int tid, int lpr) { return tid / lpr; }

template <int NV> struct PassUnroll { static constexpr int value = NV == 1 ? 8 : 4; };
template <int LPR> struct PassWaves { static constexpr int value = LPR == 8 ? 4 : 1; };

struct PassSide {
    const int32_t *partner;        // [B] id of the other side's row, this side's sorted order
    const float *w, *y;            // [B] glove_weight / glove_value in the same order
    const int32_t *chunk_id, *chunk_start;
    const float *own, *other;      // own table (rows of this side's ids) / partner table
    const float *own_bias, *other_bias;
    float *gp, *gb;                // per-chunk partial gradient rows / bias gradients of this side
    float *e_out;                  // optional [B]: e_i in this side's order (row side: diagnostics, eval)
    int n_host;                    // chunks of this side if known on the host, else -1
    int count_index;               // counts[0] (row) or counts[2] (col)
    float *mark;                   // if not null: mark[id] = 1 for every id of this side (fused Adam step)
    const int32_t *crec;           // per-chunk records {id, n, first pair, only chunk of its id | partner | w | y} or nullptr
    const uint32_t *chunk_hw;      // without records: per chunk, (first chunk of its id) << 31 | chunks of the id behind it, or nullptr
    int capP;                      // chunk_cap rounded up to a multiple of 8 (record field length)
    // fused Adagrad apply (FUSE kernels): a chunk that is its id's ONLY chunk holds the id's whole gradient in
    // registers, next to the id's own row, so the update is formed here instead of travelling through a partial
    // row: the accumulator is updated in place (nobody else reads it), the new row goes
    //   fuse == kFuseInPlace  into the table (only legal when no concurrent work reads this side's table as partner
    //                         rows: the col side in a launch of its own, after the row side)
    //   fuse == kFuseSlot     into the chunk's partial-row slot (gp / gb), from where the apply kernel copies it
    //                         into the table once every pass that gathers the old rows has finished
    // fuse == kFuseNone: every chunk stores its partial sums (the apply / dense-gradient kernels do the rest)
    //   fuse == kFuseTwin     into the OTHER copy of a twinned table (glove_tables.R_ver: rows V_row .. 2 V_row - 1 are a
    //                         second copy of the row table, own_ver[u] says which copy of row u is current): later passes
    //                         of the step keep finding the old row in the current copy, the apply kernel only flips
    //                         own_ver[u] — no slot round trip, no copy
    int fuse;
    float *own_out, *own_bias_out; // this side's table / bias vector, written by kFuseInPlace / kFuseTwin
    float *S1, *S1b;               // this side's Adagrad accumulators
    const uint8_t *own_ver, *other_ver;   // not null: that table is twinned, ver[id] != 0 = the current row is id + twin rows
    int own_twin, other_twin;             // rows between the two copies (V_row)
    //   fuse == kFusePack     nothing is applied: the id's summed gradient (incl. the activity-L2 term) goes straight into
    //                         its entry of a packed list (glove_pack_grad_f32's layout) — the multi-GPU forms, whose
    //                         gradients travel before anything is applied; the pack launch then only adds the other ids
    //                         (no fields of their own — the struct is held in scalar registers, which are short: the list
    //                         is passed in own_out, the number of entries in front of this side's in own_twin)
};

constexpr int kFuseNone = 0, kFuseSlot = 1, kFuseInPlace = 2, kFuseTwin = 3, kFusePack = 4;

struct StepConsts {
    float kappa, kappa_b;       // 2 m l2 inv_batch / d , 2 m l2 inv_batch
    float lr, eps;
    float l2, m, inv_batch, inv_d;
};

// FUSE == 1: the accumulator row a whole run is going to need is requested when the run starts (it arrives under the
// partner-row trips) and waits in LDS, written there by the load itself (global_load_lds_dwordx4: no VGPRs held across
// the trips; the image is [k][lane of the wave], which is the order that instruction writes).  Kept in registers instead
// (NV float4 more per lane) the d = 300 shape spills at 3 waves per SIMD (measured variants).
// FUSE kernels keep the accumulator row of a run in registers as well: held to 3 waves per SIMD (168 VGPRs), which the
// d = 300 shape misses by one register otherwise
// FUSE: 0 classic schedule, 1 run-merged and applying (kFuseSlot / kFuseInPlace / kFuseTwin), 2 run-merged and packing
// (kFusePack) — a build of its own: the apply code and its accumulator rows would cost the other their registers
// Run-merged passes over rows of ONE float4 per lane (d = 128: 512-byte rows, bound by requests in flight, not by bytes): four
// partner rows per trip at four waves per SIMD (<= 128 VGPRs; 114 - 121 as built, nothing spilled) instead of eight at three —
// most chunks of a big batch hold one to three pairs, so half of an eight-row trip's loads were repeats of the chunk's first
// partner, and the fourth wave hides more of every trip than the second half of the trip did.  One process, same resident plans
// (tools/ab_kernels.py, V = 2 M, d = 128, B = 1 M, twin form): 505 -> 407 us per step, bit-identical tables; five waves spill
// (815 us), eight rows at four waves spill (1,088).  Three float4 per lane (d = 300) stay at four rows and three waves: two rows
// at four waves (128 VGPRs, nothing spilled) measured 610 against 582 us at V = 400 k and 107 against 97 at V = 50 k; four rows
// at four waves (compiled for one head: 165 VGPRs at three waves, 16 spilled at four) 697 against 574 and 122 against 95.
// (the run-merged passes of 8-lane groups without records — d <= 64 on vocabularies far beyond text8's: off the benchmarked
// path — do not fit four waves any more: three asked for, nothing spilled)
// (... and with the loss head read at run time — HEAD below: the logistic heads' exp / log expansions — the run-merged passes of
// 8-lane groups do not fit four waves with records either: 29 - 34 VGPRs spilled at 128; three asked for)
// (... and neither do their builds with solid workgroups — SOLID below: 2 VGPRs spilled at four waves)
template <int LPR, int NV, int FUSE, bool REC = true, int HEAD = 0, bool SOLID = false> struct FusePass {
    static constexpr bool narrow = FUSE != 0 && LPR != 8 && NV == 1;
    static constexpr int unroll = narrow ? 4 : PassUnroll<NV>::value;
    static constexpr int waves = narrow ? 4 : (FUSE != 0 && NV <= 3 && (LPR != 8 || !REC || HEAD < 0 || SOLID)) ? 3 : PassWaves<LPR>::value;
};
// HEAD: -1 = the loss head is read from the `head` argument at run time (the logistic heads); 0 = compiled for
// GLOVE_HEAD_REGRESSION (every pass of the GloVe estimator: the logistic epilogue's exp / log expansions cost the
// regression build 18 VGPRs at one float4 per lane — 119 against 101 — and 26 spilled scalar registers, whether they run or
// not.  V = 2 M, d = 128, B = 1 M on the same plans, one process: 397 -> 355 us per step; V = 400 k, d = 300: 562 -> 554)
// SIDE: -1 / -2 = the launch holds both sides (row side in the first row_blocks workgroups; -2: the twin form's launches on row
// tables of 128 MB and more, with the streaming cache policy below); 1 / 0 = it holds the row / the col side
// alone (the three-launch fused form's launches, and the twin form's on tables the caches hold; the col-side launch of the twin
// form — SIDE 0 or -2 — may carry the list tail): the col side's build then drops what only the loss needs — |c|^2 of every
// partner row (a fifth of a trip's arithmetic), the bias squares, e . diff — and with them 17 - 27 VGPRs (d = 300: 126 against
// 153: a fourth wave per SIMD).
// The twin form's col-side launch carries, behind its pass workgroups, the workgroups that list the light ids the apply launch
// still has to visit (a thread per distinct id: an id one run did not hold completely goes onto work[4 ...], work[0] counts
// them — zeroed by the row-side launch before).  The list depends on the plan alone, so it is drawn in the shadow of the pass's
// last workgroups instead of in a launch of its own between pass and apply (triage_kernel: 5 us + a launch boundary per step;
// the version flips of the finished row ids, which do have to wait for the col side's gathers, moved into the apply launch).
struct ListTail {
    const int32_t *counts;          // device counts[8]
    const int32_t *rec_r, *rec_c;   // {id, first chunk, chunks, pairs} per distinct id
    int nu_r_host, nu_c_host;       // -1 = read the device counts
    int heavy_chunks, per;
    int first_block;                // workgroups from here on list; < 0: no tail in this launch
};

__device__ inline void list_unfinished_ids(const ListTail &tl, int32_t *__restrict__ work, int q)
{
    const int nu_r = tl.nu_r_host >= 0 ? tl.nu_r_host : tl.counts[1];
    const int nu_c = tl.nu_c_host >= 0 ? tl.nu_c_host : tl.counts[3];
    if (q >= nu_r + nu_c) return;
    const int4 rec = q < nu_r ? reinterpret_cast<const int4 *>(tl.rec_r)[q] : reinterpret_cast<const int4 *>(tl.rec_c)[q - nu_r];
    if (rec.z > tl.heavy_chunks) return;                                            // heavy ids keep their own workgroups
    const int runs = 1 + (rec.y + rec.z - 1) / tl.per - rec.y / tl.per;             // (Slots::count)
    if (runs == 1) return;                                                          // one run held it: the pass applied it
    work[4 + atomicAdd(work, 1)] = q;
}

// SOLID: the build that adds up, per workgroup, the sums of lane groups that all hold one id (batches of 131,072 chunks a side and
// more: solid_groups) — a build of its own, so that the others keep their registers (compiled into all of them the epilogue cost
// five shapes a wave per SIMD and the 8-lane shapes 45 spilled VGPRs)
template <int LPR, int NV, bool FULL, bool REC, int FUSE, int HEAD = -1, int SIDE = -1, bool SOLID = false>
__global__ __launch_bounds__(kBlock, (FusePass<LPR, NV, FUSE, REC, HEAD, SOLID>::waves)) void sidepass_kernel(
    const int32_t *__restrict__ counts, PassSide rowside, PassSide colside, int row_blocks,
    const float *__restrict__ scalars, int64_t *__restrict__ step, int d4, float inv_batch,
    float *__restrict__ blockpart, int head, float neg_factor, StepConsts kc, int per, int32_t *__restrict__ work, ListTail tail)
{
    if (FUSE == 1 && (SIDE == -2 || SIDE == 0) && tail.first_block >= 0 && (int)blockIdx.x >= tail.first_block) {      // (block-uniform)
        list_unfinished_ids(tail, work, ((int)blockIdx.x - tail.first_block) * kBlock + (int)threadIdx.x);
        return;
    }
    constexpr int GPB = kBlock / LPR;
    constexpr int U = FusePass<LPR, NV, FUSE>::unroll;
    constexpr int SL = kChunkMax / LPR > 0 ? kChunkMax / LPR : 1;     // pairs a lane stages
    static_assert(U % 4 == 0 && U <= kRecPad && kRecPad % U == 0, "fields are read back four pairs at a time; record fields are padded to kRecPad slots");
    // [group][field][pair]: 0 partner, 1 w (REC) or w2 = 2 w inv_batch, 2 y.  With records the group's LDS image
    // is the record itself: header float4, then the blocks of 8 pairs (glove_common.h)
    constexpr int kRecStride = 4 + 3 * kChunkMax + 4;        // dwords; +16 B staggers the groups over the banks
    static_assert(kChunkMax % kRecPad == 0 && 4 + 3 * kChunkMax <= kRecStride, "the largest record fits a group's LDS image");
    __shared__ __attribute__((aligned(16))) uint32_t fld_raw[GPB * (REC ? kRecStride : 3 * kFieldStride)];
    uint32_t(*fld)[3][kFieldStride] = reinterpret_cast<uint32_t(*)[3][kFieldStride]>(fld_raw);
    uint32_t *rec = fld_raw + grp_of(threadIdx.x, LPR) * kRecStride;
    constexpr bool kPark = FUSE == 1;
    __shared__ __attribute__((aligned(16))) f4 park_raw[kPark ? (kBlock / 64) * NV * 64 : 1];
    f4 *park = park_raw + (kPark ? (threadIdx.x / 64) * NV * 64 : 0);       // this wave's image
    const int lg = threadIdx.x % LPR;
    const int grp = threadIdx.x / LPR;
    const bool is_row = SIDE < 0 ? (int)blockIdx.x < row_blocks : SIDE == 1;
    const PassSide &sd = is_row ? rowside : colside;
    const int bid = is_row ? blockIdx.x : blockIdx.x - row_blocks;
    const int nblk = is_row ? row_blocks : (FUSE == 1 && (SIDE == -2 || SIDE == 0) && tail.first_block >= 0 ? tail.first_block : (int)gridDim.x) - row_blocks;
    GLOVE_STAMP(0);
    const int n_chunks = sd.n_host >= 0 ? sd.n_host : counts[sd.count_index];
    const float g = scalars[0];
    // SIDE == -2: the twin form's launches — tables beyond the caches: the streams of a fused pass leave the L2 to the partner
    // rows (see "cache policy of the run-merged passes" above).  A build of its own: on cache-resident tables (V = 50 k, d = 300:
    // the three-launch form) the policy costs 8 % (94.0 -> 101.8 us per step: the next launch finds the finished rows in the caches
    // there), and as a run-time switch it cost every shape 1.2 %.
    constexpr bool streams = FUSE == 1 && SIDE == -2;
    if (is_row && blockIdx.x == 0 && threadIdx.x == 0) {
        *step += 1;                         // global_step (see glove_hip.h)
        if (FUSE && work) work[0] = 0;      // the apply side's work list starts empty (triage_kernel)
    }

    // loss partials (row side only): [0] sum e diff (= 2 inv_batch sum w diff^2), [1] sum |r|^2+|c|^2,
    // [2] sum b^2, [3] sum e
    float part[kPartials] = {0.f, 0.f, 0.f, 0.f};

    // Chunk schedule.
    //   classic: group (bid, grp) takes chunks bid*GPB+grp, += nblk*GPB; every chunk stores its own partial row.
    //   FUSE:    group number gq = bid*GPB+grp owns the `per` CONSECUTIVE chunks [gq*per, (gq+1)*per).  Chunks are sorted
    //            by id, so the group meets its ids as runs of consecutive chunks: the own row is loaded once per run, the
    //            gradient keeps accumulating in registers across the run's chunks (pairs in plan order), and a run that
    //            holds ALL chunks of its id is applied right here (sd.fuse); any other run stores ONE partial row, in the
    //            slot of its first chunk.  The apply kernel derives the same runs from (first chunk, chunks, per).
    int j = FUSE ? (bid * GPB + grp) * per : bid * GPB + grp;
    const int j_end = FUSE ? (j + per < n_chunks ? j + per : n_chunks) : n_chunks;
    const int j_step = FUSE ? 1 : nblk * GPB;
    // A plan without records (FUSE, run words): the descriptors of ALL the group's chunks — start, id, run word: 12 B each from
    // three arrays — arrive in one round trip up front and wait in LDS; a chunk then costs what a record costs: one trip for its
    // pair fields, then its partner rows.  (Fetched chunk by chunk they were a third dependent trip: 599 against 584 us per C4 step.)
    constexpr bool kDesc = FUSE == 1 && !REC;
    constexpr int kDescSlots = 16;
    __shared__ int32_t desc_raw[kDesc ? GPB * 4 * kDescSlots : 1];
    int32_t *desc = desc_raw + (kDesc ? grp * 4 * kDescSlots : 0);
    const bool use_desc = kDesc && sd.chunk_hw != nullptr && per < kDescSlots;
    const int j_first = j;
    // ... and the pair fields of all its chunks are ONE contiguous range of the side's arrays (a few dozen pairs for the short
    // chunks of a big batch): when they fit the group's stage they come in a second trip, coalesced, and the chunks take them
    // from LDS — no memory trip per chunk but its partner rows.  A longer range (the full chunks of the Zipf head) is read
    // chunk by chunk.
    constexpr int kStagePairs = 4 * LPR;        // 12 KB of LDS per workgroup at every shape
    __shared__ uint32_t stage_raw[kDesc ? GPB * 3 * kStagePairs : 1];
    uint32_t *stage = stage_raw + (kDesc ? grp * 3 * kStagePairs : 0);
    int stage_first = 0;
    bool staged = false;
    if (use_desc && j < j_end) {
        for (int x = lg; x <= j_end - j; x += LPR) {                 // (a group of 8 lanes owns up to 12 chunks: two rounds)
            desc[x] = sd.chunk_start[j + x];
            if (x < j_end - j) {
                const int32_t cid = sd.chunk_id[j + x];
                desc[kDescSlots + x] = cid;
                desc[2 * kDescSlots + x] = (int32_t)sd.chunk_hw[j + x];
                // twinned own table: which copy of the chunk's row is current, looked up here — behind the id, beside the pair
                // fields' trip — instead of as a dependent trip in front of every run's own-row load
                if (sd.own_ver) desc[3 * kDescSlots + x] = sd.own_ver[cid];
            }
        }
        stage_first = desc[0];
        const int span = desc[j_end - j] - stage_first;
        staged = span <= kStagePairs;
        if (staged) {
            for (int i = lg; i < span; i += LPR) {
                uint32_t pid = (uint32_t)sd.partner[stage_first + i];
                // twinned partner table: the staged ids become the rows that are current, here, once for all the group's chunks
                // (one byte gather per pair behind the id's load) instead of a dependent trip in front of every chunk's rows
                if (sd.other_ver) pid += sd.other_ver[pid] ? (uint32_t)sd.other_twin : 0u;
                stage[i] = pid;
                stage[kStagePairs + i] = __float_as_uint(sd.w[stage_first + i]);
                stage[2 * kStagePairs + i] = __float_as_uint(sd.y[stage_first + i]);
            }
        }
    }
    // FUSE == 1: a workgroup all of whose chunks belong to ONE id (the head of a Zipf batch: its full chunks fill dozens of
    // workgroups) adds its groups' sums up before it leaves — one partial row per workgroup instead of one per group for the one
    // workgroup that has to sum them in the apply launch (its longest chain: V = 400 k, B = 1 M, the id of rank 1: 400 rows)
    bool solid = false;
    if (FUSE == 1 && SOLID) {
        const int f0 = bid * GPB * per, l0 = f0 + GPB * per - 1;
        if (l0 < n_chunks) {
            if (REC) {
                const uint4 *rq = reinterpret_cast<const uint4 *>(sd.crec);
                const size_t stride = rec_stride_q(sd.capP);
                solid = rq[(size_t)f0 * stride].x == rq[(size_t)l0 * stride].x;
            } else {
                solid = sd.chunk_id[f0] == sd.chunk_id[l0];
            }
        }
    }
    int32_t cur_u = -1, own_at = 0;
    int run_first = 0, run_pairs = 0, run_q = 0;
    bool run_whole = false;
    f4 r[NV], acc[NV], A[NV];
    float own_b = 0.f, bg = 0.f, Ab = 0.f, se = 0.f;
    (void)A; (void)Ab;

    bool pending = false;                   // FUSE: a run whose sums are still in registers
    for (;; j += j_step) {
        const bool have = j < j_end;
        int32_t u = -1;
        int s = 0, n = 0, uq = 0;           // uq (records only): position of the id among the side's distinct ids
        uint32_t hw = 0;                    // record header word 3: (first chunk of its id) << 31 | chunks of the id after this one
        const int capP = sd.capP;
        if (!have) {
        } else if (REC) {
            // the record (descriptor + pair fields) in contiguous 16-B loads -> LDS as is
            const int rq = 1 + 3 * capP / 4;                // float4 of the packed record (the LDS image)
            const uint4 *rp = reinterpret_cast<const uint4 *>(sd.crec) + (size_t)j * rec_stride_q(capP);   // line 0: header | block 0 | pad
            uint4 *lrec = reinterpret_cast<uint4 *>(rec);
            if (FUSE != 0 && NV >= 3) {
                // run-merged passes over wide rows (bound by bytes): header + first block of 8 pairs (112 B, one round
                // trip); a chunk of more pairs fetches its other blocks once the header has told how many (a second
                // round trip for the long chunks only) — the padding up to the cap, 3/4 of a record on average, is
                // never read.  Measured in one process against whole-record reads: V = 400 k, d = 300, B = 1 M
                // 574 vs 579 us per step; V = 2 M, d = 128 495.6 vs 492.6 (512-B rows: bound by requests in flight,
                // not by bytes — hence NV >= 3; again with records on lines of their own: 532.6 vs 527.2); V = 400 k at B = 131,072 139.0 vs 139.6
                constexpr int kHead = 1 + 6;
                static_assert(LPR >= kHead, "one load per lane covers the header and the first block");
                if (lg < kHead) lrec[lg] = rp[lg];
                const int n1 = (int)lrec[0].y;
                const int nq = 1 + 6 * ((n1 + kRecPad - 1) / kRecPad);
                for (int f = kHead + lg; f < nq; f += LPR) lrec[f] = rp[f + 1];     // blocks 1 ..: behind line 0's padding
            } else {
                // latency-bound forms: ONE round trip for the whole record
                for (int f = lg; f < rq; f += LPR) lrec[f] = rp[rec_gq(f)];
            }
            const uint4 hdr = lrec[0];          // same wave wrote it: LDS ops of one wave complete in order
            u = (int32_t)hdr.x;
            n = (int)hdr.y;
            uq = (int)hdr.z;
            hw = hdr.w;
            GLOVE_DRAIN(); GLOVE_STAMP(1);
        } else {
            if (use_desc) {                     // (same wave wrote them: LDS ops of one wave complete in order)
                const int x = j - j_first;
                s = desc[x];
                n = desc[x + 1] - s;
                u = desc[kDescSlots + x];
                hw = (uint32_t)desc[2 * kDescSlots + x];
            } else {
                u = sd.chunk_id[j];
                s = sd.chunk_start[j];
                n = sd.chunk_start[j + 1] - s;
                if (FUSE && sd.chunk_hw) hw = sd.chunk_hw[j];       // (a plan without records that carries its run words: glove_plan.r_chunk_hw)
            }
            GLOVE_DRAIN(); GLOVE_STAMP(1);      // descriptor arrived
#pragma unroll
            for (int sl = 0; sl < SL; ++sl) {
                const int t = lg + sl * LPR;
                int32_t pv;
                float wv, yv;
                if (kDesc && staged) {              // (the group's stage: same wave wrote it)
                    const int k = s - stage_first + (t < n ? t : 0);
                    pv = (int32_t)stage[k];
                    wv = __uint_as_float(stage[kStagePairs + k]);
                    yv = __uint_as_float(stage[2 * kStagePairs + k]);
                } else {
                    const int k = s + (t < n ? t : 0);
                    pv = sd.partner[k];
                    wv = sd.w[k];
                    yv = sd.y[k];
                }
                if (t < kChunkMax) {
                    fld[grp][0][t] = (uint32_t)pv;
                    fld[grp][1][t] = __float_as_uint(t < n ? wv : 0.f);     // tail slots weigh 0
                    fld[grp][2][t] = __float_as_uint(yv);
                }
            }
        }
        if (FUSE && have && sd.other_ver && !(kDesc && staged)) {      // (staged pairs were resolved when they were staged)
            // twinned partner table: the staged partner ids become the rows that are current.  One byte gather per pair,
            // issued with the own-row load below (whose latency covers it), instead of a dependent hop in every trip
            for (int t = lg; t < n; t += LPR) {
                uint32_t *slot = REC ? rec + rec_pair(t) : &fld[grp][0][t];
                const uint32_t id = *slot;
                *slot = id + (sd.other_ver[id] ? (uint32_t)sd.other_twin : 0u);
            }
        }
        const bool new_run = !FUSE || !pending || u != cur_u;
        if (FUSE && pending && (!have || new_run)) {        // the run in registers is complete
            pending = false;
            if (FUSE == 2 && run_whole) {
                // the id's whole gradient is here and it is wanted as a packed-list entry (PackGrad's arithmetic)
                const float cnt = (float)run_pairs;
                const float kcn = kc.kappa * cnt;
                float Gb = se;
#pragma unroll
                for (int k = 0; k < NV; ++k) acc[k] += kcn * r[k];
                Gb += kc.kappa_b * cnt * own_b;
                f4 *e = reinterpret_cast<f4 *>(sd.own_out) + (size_t)(sd.own_twin + run_q) * ((size_t)d4 + 1);
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int i4 = lg + k * LPR;
                    if (i4 < d4) e[i4] = acc[k];
                }
                if (lg == 0) e[d4] = f4{Gb, __int_as_float(cur_u), __int_as_float(is_row ? 0 : 1), 0.f};
            } else if (FUSE == 1 && run_whole) {
                // the id's whole gradient is here: G = sum + activity-L2 term, then Adagrad (the arithmetic of
                // for_each_id + AdagradApply; on an id with a single chunk expression for expression, bit-identical)
                const float cnt = (float)run_pairs;
                const float kcn = kc.kappa * cnt;
                float bval = own_b, Gb = se;
#pragma unroll
                for (int k = 0; k < NV; ++k) acc[k] += kcn * r[k];
                Gb += kc.kappa_b * cnt * bval;
                // every load of the run has been consumed by now, the parked row (requested before them) has landed
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < NV; ++k) A[k] = park[k * 64 + (threadIdx.x & 63)];
#pragma unroll
                for (int k = 0; k < NV; ++k) adagrad_vec(r[k], A[k], acc[k], kc.lr, kc.eps);
                if (streams) store_row_wt<LPR, NV>(sd.S1, (size_t)cur_u, d4, lg, A); else store_row<LPR, NV>(sd.S1, (size_t)cur_u, d4, lg, A);
                const bool to_slot = sd.fuse == kFuseSlot;
                // in place: the row itself; twin: the copy that is NOT current (own_at is the current one)
                const size_t out_at = sd.fuse == kFuseTwin ? (size_t)(own_at == cur_u ? cur_u + sd.own_twin : cur_u) : (size_t)cur_u;
                if (streams) store_row_wt<LPR, NV>(to_slot ? sd.gp : sd.own_out, to_slot ? (size_t)run_first : out_at, d4, lg, r);
                else store_row<LPR, NV>(to_slot ? sd.gp : sd.own_out, to_slot ? (size_t)run_first : out_at, d4, lg, r);
                if (lg == 0) {
                    adagrad_elem(bval, Ab, Gb, kc.lr, kc.eps);
                    sd.S1b[cur_u] = Ab;
                    if (to_slot) sd.gb[run_first] = bval; else sd.own_bias_out[out_at] = bval;
                }
            } else if (!(FUSE == 1 && SOLID && solid)) {     // (a solid workgroup's sums leave together, below)
                store_row<LPR, NV>(sd.gp, (size_t)run_first, d4, lg, acc);
                if (lg == 0) {
                    sd.gb[run_first] = se;
                    if (FUSE == 0 && sd.mark) sd.mark[cur_u] = 1.0f;
                }
            }
        }
        if (!have) break;
        if (new_run) {
            cur_u = u;
            run_first = j;
            run_q = uq;
            run_pairs = 0;
            own_at = u;
            if (FUSE && sd.own_ver)                                                     // the current copy of a twinned table
                own_at = u + ((use_desc ? desc[3 * kDescSlots + (j - j_first)] : (int32_t)sd.own_ver[u]) ? sd.own_twin : 0);
            if (FUSE && streams) load_row_nt<LPR, NV>(r, sd.own, own_at, d4, lg);
            else if (FUSE) load_row<LPR, NV>(r, sd.own, own_at, d4, lg);
            else load_row<LPR, NV>(r, sd.own, u, d4, lg);
            own_b = sd.own_bias[own_at];
            bg = own_b + g;
            // whole: the run starts at the id's first chunk and the id's last chunk is still inside this group's range
            run_whole = FUSE && sd.fuse != kFuseNone && (hw >> 31) != 0 && j + (int)(hw & 0x7fffffffu) < j_end;
            if (FUSE == 1 && run_whole) {   // requested with the own row: arrives under the partner-row trips
                const f4 *src = reinterpret_cast<const f4 *>(sd.S1) + (size_t)u * d4;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int i4 = lg + k * LPR;
                    // destination: the wave-uniform base + lane * 16 B (the hardware adds the lane part)
                    if (!(FULL || i4 < d4)) continue;
                    if (streams) __builtin_amdgcn_global_load_lds(src + i4, (__attribute__((address_space(3))) void *)(park + k * 64), 16, 0, 2);      // aux 2 = nt
                    else __builtin_amdgcn_global_load_lds(src + i4, (__attribute__((address_space(3))) void *)(park + k * 64), 16, 0, 0);
                }
                Ab = sd.S1b[u];
            }
#pragma unroll
            for (int k = 0; k < NV; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
            se = 0.f;
        }
        run_pairs += n;
        float se_c = 0.f, cc_sum = 0.f, bsq = 0.f, ed = 0.f;
        GLOVE_DRAIN(); GLOVE_STAMP(2);      // pair fields staged, own row arrived
        // fields written by this wave's own lanes: LDS ops of one wave complete in order

        for (int q0 = 0; q0 < n; q0 += U) {
            int32_t col[U];
            float w2[U], yq[U];
#pragma unroll
            for (int a4 = 0; a4 < U; a4 += 4) {
                const int rp0 = rec_pair(q0 + a4);          // records: blocks of 8 pairs {partner | w | y}
                const uint4 pc = *reinterpret_cast<const uint4 *>(REC ? &rec[rp0] : &fld[grp][0][q0 + a4]);
                const uint4 pw = *reinterpret_cast<const uint4 *>(REC ? &rec[rp0 + kRecPad] : &fld[grp][1][q0 + a4]);
                const uint4 py = *reinterpret_cast<const uint4 *>(REC ? &rec[rp0 + 2 * kRecPad] : &fld[grp][2][q0 + a4]);
                col[a4] = (int32_t)pc.x; col[a4 + 1] = (int32_t)pc.y; col[a4 + 2] = (int32_t)pc.z; col[a4 + 3] = (int32_t)pc.w;
                const float sc2 = 2.0f * inv_batch;
                w2[a4] = sc2 * __uint_as_float(pw.x); w2[a4 + 1] = sc2 * __uint_as_float(pw.y);
                w2[a4 + 2] = sc2 * __uint_as_float(pw.z); w2[a4 + 3] = sc2 * __uint_as_float(pw.w);
                yq[a4] = __uint_as_float(py.x); yq[a4 + 1] = __uint_as_float(py.y);
                yq[a4 + 2] = __uint_as_float(py.z); yq[a4 + 3] = __uint_as_float(py.w);
            }
            f4 c[U][NV];
            float bcv[U];
#pragma unroll
            for (int a = 0; a < U; ++a) {
                load_row_fast<LPR, NV, FULL>(c[a], sd.other, col[a], d4, lg);
                // the partner bias enters the dot once, through lane 0 of the group (a masked 1-lane-per-group
                // load instead of a 64-lane gather of the same 4 bytes), and the butterfly spreads it
                bcv[a] = 0.f;
                if (lg == 0) bcv[a] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(sd.other_bias) + (uint32_t)col[a] * 4u);
            }
            float dp[U], cc[U];
#pragma unroll
            for (int a = 0; a < U; ++a) {
                dp[a] = 0.f; cc[a] = 0.f;
#pragma unroll
                for (int k = 0; k < NV; ++k) { dp[a] += dot4(r[k], c[a][k]); cc[a] += dot4(c[a][k], c[a][k]); }
                dp[a] += bcv[a];                                     // non-zero in lane 0 only
            }
            // the U butterflies are independent: stage by stage so the DPP hazards overlap
#pragma unroll
            for (int a = 0; a < U; ++a) dp[a] = dpp_add<0xB1>(dp[a]);
#pragma unroll
            for (int a = 0; a < U; ++a) dp[a] = dpp_add<0x4E>(dp[a]);
#pragma unroll
            for (int a = 0; a < U; ++a) dp[a] = dpp_add<0x141>(dp[a]);
            if (LPR >= 16) {
#pragma unroll
                for (int a = 0; a < U; ++a) dp[a] = dpp_add<0x140>(dp[a]);
            }
            if (LPR >= 32) {
#pragma unroll
                for (int a = 0; a < U; ++a) dp[a] += __shfl_xor(dp[a], 16, 64);
            }
            if (LPR >= 64) {
#pragma unroll
                for (int a = 0; a < U; ++a) dp[a] += __shfl_xor(dp[a], 32, 64);
            }
            float ev[U];
#pragma unroll
            for (int a = 0; a < U; ++a) {
                const float valid = (q0 + a < n) ? 1.0f : 0.f;
                float e;
                if (HEAD == GLOVE_HEAD_REGRESSION || (HEAD < 0 && head == GLOVE_HEAD_REGRESSION)) {                 // uniform branch
                    float logit = dp[a] + bg;
                    asm("" : "+v"(logit));                          // (the logit, the difference and e: each a rounded value of its own in
                    float diff = logit - yq[a];                     // every build — see below)
                    asm("" : "+v"(diff));
                    e = w2[a] * diff;                               // 0 on tail slots
                    // (e is a rounded product for every consumer, as it is where the head is a run-time branch and e leaves it
                    // through a select: compiled for one head the sum of e's below would otherwise contract into fma(w2, diff, .)
                    // and the builds would differ in the last bit of the bias gradients)
                    asm("" : "+v"(e));
                    ed += e * diff;
                } else {
                    // pos / neg logistic heads on one logit: dL/dp = (pos (s - 1) + nf neg s) / B, s = sigmoid(p);
                    // `ed` collects 2/B times the pair's loss so that the common rescaling below applies
                    const float p = dp[a] + bg;
                    const float en = expf(-fabsf(p));               // in (0, 1]: no overflow either way
                    const float s = (p >= 0.f ? 1.0f : en) / (1.0f + en);
                    const float lse = log1pf(en);                   // softplus(x) = max(x, 0) + log1p(exp(-|x|))
                    const float wn = 2.0f * inv_batch * neg_factor * valid * yq[a];
                    e = 0.5f * (w2[a] * (s - 1.0f) + wn * s);
                    ed += w2[a] * (fmaxf(-p, 0.f) + lse) + wn * (fmaxf(p, 0.f) + lse);
                }
#pragma unroll
                for (int k = 0; k < NV; ++k) acc[k] += e * c[a][k];
                se_c += e;
                cc_sum += valid * cc[a];
                bsq += valid * bcv[a] * bcv[a];
                ev[a] = e;
            }
            if (FUSE == 0 && !REC && sd.e_out) {                  // glove_rowpass_f32 only (classic schedule): e_i for diagnostics
                constexpr int ES = (U + LPR - 1) / LPR;      // pairs of this trip whose e a lane stores
#pragma unroll
                for (int x = 0; x < ES; ++x) {
                    float mine = 0.f;
#pragma unroll
                    for (int a = x * LPR; a < U && a < (x + 1) * LPR; ++a) mine = (lg == a - x * LPR) ? ev[a] : mine;
                    const int q = q0 + lg + x * LPR;
                    if (lg + x * LPR < U && q < n) sd.e_out[s + q] = mine;
                }
            }
        }
        GLOVE_STAMP(3);                     // all partner-row trips issued and consumed
        se += se_c;
        if (is_row) {
            float rr = 0.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) rr += dot4(r[k], r[k]);
            part[1] += cc_sum + (float)n * rr;
            if (lg == 0) {
                part[0] += ed;
                part[2] += bsq + (float)n * own_b * own_b;
                part[3] += se_c;
            }
        }
        if (FUSE) {
            pending = true;
        } else {
            store_row<LPR, NV>(sd.gp, (size_t)j, d4, lg, acc);
            if (lg == 0) {
                sd.gb[j] = se;
                if (sd.mark) sd.mark[u] = 1.0f;
            }
        }
    }
    if (FUSE == 1 && SOLID && solid) {      // block-uniform
        // every group holds the sums of its one run (same id everywhere): groups 0 .. GPB - 1 added in that order by group 0, one
        // partial row in the slot of the workgroup's first chunk.  (A wave writes into its own part of the parking image, which
        // no run of a solid workgroup has used: no run of it was whole.)
        __shared__ float solid_bias[SOLID ? GPB : 1];
        f4 *sums = park_raw;
#pragma unroll
        for (int k = 0; k < NV; ++k) sums[(grp * NV + k) * LPR + lg] = acc[k];
        if (lg == 0) solid_bias[grp] = se;
        __syncthreads();
        if (grp == 0) {
#pragma unroll 1
            for (int g2 = 1; g2 < GPB; ++g2) {
#pragma unroll
                for (int k = 0; k < NV; ++k) acc[k] += sums[(g2 * NV + k) * LPR + lg];
                se += solid_bias[g2];
            }
            const size_t slot = (size_t)bid * GPB * per;
            store_row<LPR, NV>(sd.gp, slot, d4, lg, acc);
            if (lg == 0) sd.gb[slot] = se;
        }
    }
    GLOVE_DRAIN(); GLOVE_STAMP(4);          // stores retired
    if (is_row) {                           // block-uniform
        part[0] *= 0.5f / inv_batch;        // sum e diff = 2 inv_batch sum w diff^2
        block_partials_store(part, blockpart);
    }
    GLOVE_DRAIN(); GLOVE_STAMP(5);
}

// ------------------------------------------------------------------------------------------
// Summed gradient of one distinct id: chunk partials in a fixed order + activity-L2 term.
// Light ids (at most plan.heavy_chunks chunks: almost all of them) are summed by one group of LPR
// lanes.  The ids of the Zipf head ("the", "<UNK>" own thousands of pairs of a large batch) are
// listed in plan.heavy; each gets a whole workgroup (the first `heavy_blocks` blocks of the grid, so
// they start ahead of the light work and run beside it): GPB groups stride over the partial rows,
// then group 0 adds the GPB sums in order.
// ------------------------------------------------------------------------------------------
struct SideBufs {
    const int32_t *uniq_rec;
    const float *gp, *gb;
    float *W, *S1, *bias, *S1b;
    uint8_t *ver;               // not null: W / bias are twinned (glove_tables.R_ver), ver[id] != 0 = current row is id + twin
    int twin;
};

// row of W / entry of bias that currently holds id (the slots S1 / S1b are never twinned)
__device__ inline int32_t cur_row(const SideBufs &sb, int32_t id) { return sb.ver && sb.ver[id] ? id + sb.twin : id; }

struct IdWork {
    const int32_t *counts;      // device counts[8]
    int nu_r_host, nu_c_host, n_heavy_host;   // host copies, -1 = read the device counts
    const int32_t *heavy;       // (side << 30) | q
    int heavy_blocks;           // leading blocks of the grid reserved for heavy ids
    int heavy_chunks;           // threshold
    int sides;                  // 1 = rows only, 2 = cols only, 3 = both
    int pre_r, pre_c;           // what a FUSE pass already did for the ids one run held completely (kFuse*), per side
    int per;                    // consecutive chunks per group of that FUSE pass
    int gpb;                    // ... and its lane groups per workgroup (solid workgroups leave one partial row: Slots); 0 = none
    const int32_t *work;        // not null: triage_kernel listed the light ids that still need this launch: work[0] = their
                                // number, work[4 ...] = their positions q (row ids first, col ids behind nu_r)
    int flips;                  // 1: the list came from the col-side launch (ListTail): the version flips of the row ids a run of the
                                // twin form's row pass finished are still to do — here, a thread per row id
};

// The partial rows of one id.  Classic passes: one per chunk, slots f .. f+n-1 (per == 1).  FUSE passes: one per run, a
// run being the id's chunks inside one group's range of `per` consecutive chunks: the k-th run starts at the id's
// first chunk (k == 0) or at the k-th multiple of `per` behind it.
// Behind a run-merged applying pass (gpb > 0: its lane groups per workgroup) a workgroup whose gpb * per chunks ALL belong to the
// id — a "solid" workgroup: the head of a Zipf batch owns dozens of them — has added its groups' sums up in LDS and left ONE
// partial row, in the slot of its first chunk (sidepass_kernel): the runs of the id are then the group ranges in front of its first
// solid workgroup, one per solid workgroup, and the group ranges behind the last (the id of rank 1 of the headline batch: 55 partial
// rows instead of 400 for the one workgroup that sums them in the apply launch).
struct Slots {
    int f, per, count;
    int gpb, g0, head, sb0, nsb;
    bool one_group;
    __device__ Slots(int first_chunk, int chunks, int per_, int gpb_ = 0) : f(first_chunk), per(per_), gpb(gpb_)
    {
        g0 = first_chunk / per_;
        const int g1 = (first_chunk + chunks - 1) / per_;
        one_group = g1 == g0;                                       // one group's range held all its chunks: the pass applied it
        count = head = 1 + g1 - g0;
        sb0 = nsb = 0;
        if (gpb_ > 0) {
            const int bc = gpb_ * per_;                             // chunks of a workgroup
            const int bs = (first_chunk + bc - 1) / bc;             // workgroups [bs, be) lie inside the id's chunks
            const int be = (first_chunk + chunks) / bc;
            if (be > bs) {
                sb0 = bs;
                nsb = be - bs;
                head = bs * gpb_ - g0;
                count = head + nsb + (g1 + 1 - be * gpb_);
            }
        }
    }
    __device__ int at(int k) const
    {
        if (k == 0) return f;
        if (k < head) return (g0 + k) * per;
        if (k < head + nsb) return (sb0 + (k - head)) * gpb * per;
        return ((sb0 + nsb) * gpb + (k - head - nsb)) * per;
    }
};

// G += partial rows k0, k0+stride, ... (< sl.count), PB loads in flight at a time, added in order
template <int LPR, int NV, int PB = 4>
__device__ inline void sum_partials(const SideBufs &sb, const Slots &sl, int k0, int stride, int d4, int lg,
                                    f4 (&G)[NV], float &Gb)
{
    for (int k = k0; k < sl.count; k += stride * PB) {
        f4 p[PB][NV];
        float pbias[PB];
#pragma unroll
        for (int a = 0; a < PB; ++a) {
            const int x = k + a * stride;
            const int xs = sl.at(x < sl.count ? x : 0);
            load_row<LPR, NV>(p[a], sb.gp, xs, d4, lg);
            pbias[a] = sb.gb[xs];
        }
#pragma unroll
        for (int a = 0; a < PB; ++a) {
            const float on = (k + a * stride < sl.count) ? 1.0f : 0.f;
#pragma unroll
            for (int kk = 0; kk < NV; ++kk) G[kk] += on * p[a][kk];
            Gb += on * pbias[a];
        }
    }
}

// Visits every distinct id of both sides once.  `fn` supplies
//   fn.prefetch(is_row, id, st)                     its own rows + bias slots (F::State: the Adagrad
//                                                   accumulator, Adam's m and v, or the dense gradient
//                                                   row), requested together with the table row so the
//                                                   latencies overlap
//   fn.finish(is_row, id, wid, q, G, Wv, Gb, bval, st)   G = summed gradient incl. the activity-L2 term; wid = row of
//                                                   the table that holds id (id itself unless the table is twinned); q = position of
//                                                   the id among its side's distinct ids (plan order)
// Returns true in the workgroup that should also do the once-per-step scalar work.
template <int LPR, int NV, class F>
__device__ inline bool for_each_id(const IdWork &wk, const SideBufs &rs, const SideBufs &cs,
                                   int d4, const StepConsts &k, F fn, int grid_blocks = 0)
{
    const int nblocks = grid_blocks > 0 ? grid_blocks : (int)gridDim.x;    // workgroups doing this traversal
    constexpr int GPB = kBlock / LPR;
    __shared__ f4 red[GPB][LPR * NV];
    __shared__ float redb[GPB];
    const int lg = threadIdx.x % LPR;
    const int grp = threadIdx.x / LPR;
    GLOVE_STAMP(0);

    if ((int)blockIdx.x < wk.heavy_blocks) {
        // ---- one heavy id for the whole workgroup
        const int n_heavy = wk.n_heavy_host >= 0 ? wk.n_heavy_host : wk.counts[4];
        if ((int)blockIdx.x >= n_heavy) return false;
        const int code = wk.heavy[blockIdx.x];
        const bool is_row = (code >> 30) == 0;
        if (!(wk.sides & (is_row ? 1 : 2))) return false;
        const SideBufs &sb = is_row ? rs : cs;
        const int4 rec = reinterpret_cast<const int4 *>(sb.uniq_rec)[code & 0x3fffffff];
        const int32_t id = rec.x;
        const int pre = is_row ? wk.pre_r : wk.pre_c;
        const Slots sl(rec.y, rec.z, pre != kFuseNone ? wk.per : 1, pre != kFuseNone ? wk.gpb : 0);
        if (pre != kFuseNone && sl.one_group) {       // one run held the whole id: the pass kernel applied it
            if (pre == kFuseSlot && grp == 0) {
                f4 Wn[NV];
                load_row<LPR, NV>(Wn, sb.gp, sl.f, d4, lg);
                const float bn = sb.gb[sl.f];
                store_row<LPR, NV>(sb.W, (size_t)id, d4, lg, Wn);
                if (lg == 0) sb.bias[id] = bn;
            }
            if (pre == kFuseTwin && threadIdx.x == 0) sb.ver[id] ^= 1;       // the other copy holds the new row
            return false;
        }
        const int32_t wid = cur_row(sb, id);
        f4 G[NV], Wv[NV];
        typename F::State st;
        float Gb = 0.f, bval = 0.f;
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) G[kk] = f4{0.f, 0.f, 0.f, 0.f};
        if (grp == 0) {
            load_row<LPR, NV>(Wv, sb.W, wid, d4, lg);
            bval = sb.bias[wid];
            fn.prefetch(is_row, id, st);
        }
        // a heavy id is the longest dependent chain of the launch (the head of a Zipf batch: ~50 rows per group):
        // four rows in flight per trip at every row width (16 cost the d = 64 shape its occupancy: 8.4 -> 9.3 us; round 5, the
        // twin form's short-list apply, whole step, same plans: V = 400 k, d = 300 548.8 us with 4, 556.5 with 8, 555.4 with 16;
        // V = 2 M, d = 128 359.1 / 358.0 / 360.1 — profiles/r05_exp_heavy_rows_in_flight.txt)
        sum_partials<LPR, NV, 4>(sb, sl, grp, GPB, d4, lg, G, Gb);
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) red[grp][lg + kk * LPR] = G[kk];
        if (lg == 0) redb[grp] = Gb;
        __syncthreads();
        if (grp == 0) {
            for (int g2 = 1; g2 < GPB; ++g2) {
#pragma unroll
                for (int kk = 0; kk < NV; ++kk) G[kk] += red[g2][lg + kk * LPR];
                Gb += redb[g2];
            }
            const float cnt = (float)rec.w;
            const float kc = k.kappa * cnt;
#pragma unroll
            for (int kk = 0; kk < NV; ++kk) G[kk] += kc * Wv[kk];
            Gb += k.kappa_b * cnt * bval;
            fn.finish(is_row, id, wid, code & 0x3fffffff, G, Wv, Gb, bval, st);
        }
        return false;
    }

    // ---- light ids: one group each, dealt round-robin over the light blocks
    const int nu_r = wk.nu_r_host >= 0 ? wk.nu_r_host : wk.counts[1];
    const int nu_c = wk.nu_c_host >= 0 ? wk.nu_c_host : wk.counts[3];
    const int q_begin = (wk.sides & 1) ? 0 : nu_r;
    const int total = (wk.sides & 2) ? nu_r + nu_c : nu_r;
    // the last workgroup of the grid only does the once-per-step scalar work, beside everyone else
    if ((int)blockIdx.x == nblocks - 1) return (wk.sides & 2) != 0;
    const int lb = blockIdx.x - wk.heavy_blocks, nlb = nblocks - wk.heavy_blocks - 1;
    if (wk.flips) {
        // twin form, list drawn by the col-side launch: the row ids one run of the row pass held completely have their new
        // row in the other copy — flip their version now that no gather reads the old one (ids are distinct: a thread per byte;
        // the ids on the list are not among them)
        for (int q = lb * kBlock + (int)threadIdx.x; q < nu_r; q += nlb * kBlock) {
            const int4 rec = reinterpret_cast<const int4 *>(rs.uniq_rec)[q];
            if (rec.z <= wk.heavy_chunks && Slots(rec.y, rec.z, wk.per).count == 1) rs.ver[rec.x] ^= 1;
        }
    }
    // behind a FUSE pass triage_kernel has listed the few ids that still need work: walk that list instead of all ids
    const int n_items = wk.work ? wk.work[0] : total - q_begin;
    for (int it = lb * GPB + grp; it < n_items; it += nlb * GPB) {
        const int q = wk.work ? wk.work[4 + it] : q_begin + it;
        const bool is_row = q < nu_r;
        const SideBufs &sb = is_row ? rs : cs;
        const int qq = is_row ? q : q - nu_r;
        const int4 rec = reinterpret_cast<const int4 *>(sb.uniq_rec)[qq];   // {id, first chunk, chunks, pairs}
        if (rec.z > wk.heavy_chunks) continue;                               // a heavy block has it
        GLOVE_DRAIN(); GLOVE_STAMP(1);      // record arrived
        const int32_t id = rec.x;
        const int pre = is_row ? wk.pre_r : wk.pre_c;
        const Slots sl(rec.y, rec.z, pre != kFuseNone ? wk.per : 1, pre != kFuseNone ? wk.gpb : 0);
        const int sl0 = sl.f;
        if (pre != kFuseNone && sl.one_group) {
            // one run held the whole id and the pass kernel already applied it (sidepass_kernel FUSE): in place ->
            // nothing left to do; into its slot -> move the finished row and bias into the table now that no pass
            // reads the old ones; twin -> the other copy holds the new row: flip the version
            if (pre == kFuseSlot) {
                f4 Wn[NV];
                load_row<LPR, NV>(Wn, sb.gp, sl0, d4, lg);
                const float bn = sb.gb[sl0];
                store_row<LPR, NV>(sb.W, (size_t)id, d4, lg, Wn);
                if (lg == 0) sb.bias[id] = bn;
            }
            if (pre == kFuseTwin && lg == 0) sb.ver[id] ^= 1;
            continue;
        }
        const int32_t wid = cur_row(sb, id);
        const float cnt = (float)rec.w;
        f4 G[NV], Wv[NV];
        typename F::State st;
        load_row<LPR, NV>(G, sb.gp, sl0, d4, lg);
        float Gb = sb.gb[sl0];
        load_row<LPR, NV>(Wv, sb.W, wid, d4, lg);
        const float bval = sb.bias[wid];
        fn.prefetch(is_row, id, st);
        sum_partials<LPR, NV>(sb, sl, 1, 1, d4, lg, G, Gb);
        const float kc = k.kappa * cnt;
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) G[kk] += kc * Wv[kk];
        Gb += k.kappa_b * cnt * bval;
        GLOVE_DRAIN(); GLOVE_STAMP(3);      // rows arrived
        fn.finish(is_row, id, wid, qq, G, Wv, Gb, bval, st);
        GLOVE_DRAIN(); GLOVE_STAMP(4);      // stores retired
    }
    GLOVE_STAMP(5);
    return false;
}

// Deterministic sum of the rowpass block partials by the whole workgroup (thread t takes blocks
// t, t+256, ...; then lanes, then waves, in a fixed order).  Result valid in thread 0.
__device__ inline void sum_blockpart(const float *blockpart, int nblocks, float (&tot)[kPartials])
{
    __shared__ float red2[kBlock / 64][kPartials];
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
    const f4 *bp = reinterpret_cast<const f4 *>(blockpart);
#pragma unroll
    for (int it = 0; it < kMaxBlocks / kBlock; ++it) {
        const int b = threadIdx.x + it * kBlock;
        const f4 v = bp[b < nblocks ? b : 0];
        acc += (b < nblocks) ? v : f4{0.f, 0.f, 0.f, 0.f};
    }
    // fused passes of big batches only (up to kMaxPassBlocks workgroups): same order, eight loads in flight at a time —
    // one dependent load per round was 12 us of the C4 apply launch
    for (int b0 = threadIdx.x + kMaxBlocks; b0 < nblocks; b0 += 8 * kBlock) {
        f4 v[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int b = b0 + x * kBlock;
            v[x] = bp[b < nblocks ? b : 0];
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) acc += (b0 + x * kBlock < nblocks) ? v[x] : f4{0.f, 0.f, 0.f, 0.f};
    }
    tot[0] = wave_sum(acc.x); tot[1] = wave_sum(acc.y); tot[2] = wave_sum(acc.z); tot[3] = wave_sum(acc.w);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < kPartials; ++i) red2[threadIdx.x >> 6][i] = tot[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < kPartials; ++i) {
            float sacc = 0.f;
#pragma unroll
            for (int wv = 0; wv < kBlock / 64; ++wv) sacc += red2[wv][i];
            tot[i] = sacc;
        }
    }
}

__device__ inline void loss_from_partials(const float (&tot)[kPartials], const StepConsts &k, float g,
                                          float &loss, float &L, float &reg)
{
    // tot = {sum w diff^2, sum |r|^2+|c|^2, sum br^2+bc^2, sum e}
    L = tot[0] * k.inv_batch;
    reg = k.l2 * k.inv_d * k.inv_batch * tot[1] + k.l2 * k.inv_batch * tot[2] + k.l2 * g * g;
    loss = L + k.m * reg;
}

// Behind a FUSE pass most ids are finished: the launch that applies the rest spent its time on one dependent record load
// per id, 20 ids per lane group in a row (V = 400 k: 57 us for 24 MB of traffic).  One THREAD per distinct id sorts
// them out first: finished ids of a twinned row table get their version flipped right here, finished ids of the
// in-place side need nothing, and the positions of the light ids that still need work — a copy out of a slot, or the
// sum of several runs — go onto a list (work[0] counts them; the row pass zeroed it).  The list's order varies from
// run to run, the work items are independent: results do not depend on it.  Heavy ids keep their own workgroups.
__global__ __launch_bounds__(kBlock) void triage_kernel(IdWork wk, SideBufs rs, SideBufs cs, int32_t *__restrict__ work)
{
    const int nu_r = wk.nu_r_host >= 0 ? wk.nu_r_host : wk.counts[1];
    const int nu_c = wk.nu_c_host >= 0 ? wk.nu_c_host : wk.counts[3];
    const int q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= nu_r + nu_c) return;
    const bool is_row = q < nu_r;
    if (!(wk.sides & (is_row ? 1 : 2))) return;          // not a side of this step
    const int4 rec = is_row ? reinterpret_cast<const int4 *>(rs.uniq_rec)[q] : reinterpret_cast<const int4 *>(cs.uniq_rec)[q - nu_r];
    if (rec.z > wk.heavy_chunks) return;
    const int pre = is_row ? wk.pre_r : wk.pre_c;
    const bool whole = pre != kFuseNone && Slots(rec.y, rec.z, wk.per).count == 1;
    if (whole && pre == kFuseInPlace) return;
    if (whole && pre == kFuseTwin) {
        rs.ver[rec.x] ^= 1;                 // ids are distinct: one thread per byte
        return;
    }
    work[4 + atomicAdd(work, 1)] = q;
}

// The loss partials of a row pass with `nblocks` workgroups, folded into entry 0 (the rest zeroed): any later reader that
// sums entries 0 .. n-1 for its own idea of n >= 1 (the classic row pass's grid) finds the same totals.
__global__ __launch_bounds__(kBlock) void fold_blockpart_kernel(float *__restrict__ blockpart, int nblocks)
{
    float tot[kPartials];
    sum_blockpart(blockpart, nblocks, tot);
    __shared__ float keep[kPartials];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < kPartials; ++i) keep[i] = tot[i];
    }
    __syncthreads();                        // everybody has read its share before anything is overwritten
    for (int i = threadIdx.x; i < kMaxPassBlocks * kPartials; i += kBlock) blockpart[i] = i < kPartials ? keep[i] : 0.f;
}

// The loss partials of the last row pass as they travel in a packed list's header: {sum e, sum w diff^2, sum |r|^2+|c|^2, sum b^2}
__global__ __launch_bounds__(kBlock) void loss_partials_kernel(const float *__restrict__ blockpart, int nblocks, float *__restrict__ out)
{
    float tot[kPartials];
    sum_blockpart(blockpart, nblocks, tot);
    if (threadIdx.x == 0) { out[0] = tot[3]; out[1] = tot[0]; out[2] = tot[1]; out[3] = tot[2]; }
}

template <int LPR, int NV>
struct AdagradApply {
    SideBufs rs, cs;
    int d4, lg;
    float lr, eps;
    struct State { f4 A[NV]; float Ab; };
    __device__ void prefetch(bool is_row, int32_t id, State &st) const
    {
        const SideBufs &sb = is_row ? rs : cs;
        load_row<LPR, NV>(st.A, sb.S1, id, d4, lg);
        st.Ab = sb.S1b[id];
    }
    __device__ void finish(bool is_row, int32_t id, int32_t wid, int q, f4 (&G)[NV], f4 (&Wv)[NV], float Gb, float bval, State &st) const
    {
        const SideBufs &sb = is_row ? rs : cs;
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) adagrad_vec(Wv[kk], st.A[kk], G[kk], lr, eps);
        store_row<LPR, NV>(sb.S1, (size_t)id, d4, lg, st.A);
        store_row<LPR, NV>(sb.W, (size_t)wid, d4, lg, Wv);
        if (lg == 0) {
            adagrad_elem(bval, st.Ab, Gb, lr, eps);
            sb.S1b[id] = st.Ab;
            sb.bias[wid] = bval;
        }
    }
};

template <int LPR, int NV>
__global__ __launch_bounds__(kBlock) void apply_adagrad_kernel(
    IdWork wk, SideBufs rs, SideBufs cs, int d4, StepConsts k,
    float *__restrict__ scalars, const float *__restrict__ blockpart, int nblocks_rowpass,
    float *__restrict__ loss_out)
{
    const bool scalar_duty = for_each_id<LPR, NV>(wk, rs, cs, d4, k,
                                                  AdagradApply<LPR, NV>{rs, cs, d4, (int)(threadIdx.x % LPR), k.lr, k.eps});
    // global bias (dense Adagrad) + loss scalars: one (light) workgroup
    if (scalar_duty) {
        float tot[kPartials];
        sum_blockpart(blockpart, nblocks_rowpass, tot);
        if (threadIdx.x == 0) {
            const float g = scalars[0];
            float loss, L, reg;
            loss_from_partials(tot, k, g, loss, L, reg);
            const float dg = tot[3] + 2.0f * k.m * k.l2 * g;
            float gn = g, Ag = scalars[1];
            adagrad_elem(gn, Ag, dg, k.lr, k.eps);
            scalars[1] = Ag;
            scalars[0] = gn;
            if (loss_out) { loss_out[0] = loss; loss_out[1] = L; loss_out[2] = reg; loss_out[3] = tot[3]; }
        }
    }
    GLOVE_DRAIN(); GLOVE_STAMP(6);
}

// Adds this plan's summed gradients into the dense buffers (no two work items share an id
// within one side, so plain read-modify-write is race free).
template <int LPR, int NV>
struct DenseGradAdd {
    float *G_R, *G_C, *G_br, *G_bc;
    int d4, lg;
    struct State { f4 old[NV]; float oldb; };
    __device__ void prefetch(bool is_row, int32_t id, State &st) const
    {
        load_row<LPR, NV>(st.old, is_row ? G_R : G_C, id, d4, lg);
        st.oldb = (is_row ? G_br : G_bc)[id];
    }
    __device__ void finish(bool is_row, int32_t id, int32_t wid, int q, f4 (&G)[NV], f4 (&Wv)[NV], float Gb, float bval, State &st) const
    {
        (void)Wv; (void)bval;
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) st.old[kk] += G[kk];
        store_row<LPR, NV>(is_row ? G_R : G_C, (size_t)id, d4, lg, st.old);
        if (lg == 0) (is_row ? G_br : G_bc)[id] = st.oldb + Gb;
    }
};

template <int LPR, int NV>
__global__ __launch_bounds__(kBlock) void dense_grad_kernel(
    IdWork wk, SideBufs rs, SideBufs cs, int d4, StepConsts k,
    float *__restrict__ G_R, float *__restrict__ G_C, float *__restrict__ G_br, float *__restrict__ G_bc,
    float *__restrict__ tail, const float *__restrict__ blockpart, int nblocks_rowpass)
{
    const bool scalar_duty = for_each_id<LPR, NV>(wk, rs, cs, d4, k,
                                                  DenseGradAdd<LPR, NV>{G_R, G_C, G_br, G_bc, d4, (int)(threadIdx.x % LPR)});
    if (scalar_duty) {
        float tot[kPartials];
        sum_blockpart(blockpart, nblocks_rowpass, tot);
        if (threadIdx.x == 0) {
            tail[0] += tot[3];   // sum e
            tail[1] += tot[0];   // sum w diff^2
            tail[2] += tot[1];   // sum |r|^2 + |c|^2
            tail[3] += tot[2];   // sum br^2 + bc^2
        }
    }
}

// ------------------------------------------------------------------------------------------
// Touched-rows exchange: instead of a dense [V,d] gradient buffer a rank hands over ONE packed list
// of (id, summed gradient row) per step.  An entry is d + 4 floats, 16-B aligned:
//     [ gradient row, d floats | bias gradient | id (int bits) | side (0 row, 1 col; int bits) | 0 ]
// Entry 0 of a list is its header: [row entries, col entries (int bits) | sum e, sum w diff^2, sum |r|^2+|c|^2,
// sum br^2+bc^2 | 0 ...]; the row-side entries follow in plan order (ascending id), then the col-side entries.
// The receiving side adds the lists of all ranks into the dense buffer G_flat IN LIST ORDER (one launch per list: ids are
// distinct within a list, launches are ordered, so the sum over ranks has a fixed order), the first list that touches an id
// storing instead of adding (G_flat needs no zeroing) and leaving its tag in `mark`; then one launch applies Adagrad to
// every touched id, each from the list that touched it first, and clears the marks.
// ------------------------------------------------------------------------------------------
constexpr int kPackExtra = 4;

template <int LPR, int NV>
struct PackGrad {
    float *packed;
    int d4, lg, row_entries;
    struct State {};
    __device__ void prefetch(bool, int32_t, State &) const {}
    __device__ void finish(bool is_row, int32_t id, int32_t wid, int q, f4 (&G)[NV], f4 (&Wv)[NV], float Gb, float bval, State &) const
    {
        (void)Wv; (void)bval;
        const size_t stride4 = (size_t)d4 + 1;
        f4 *e = reinterpret_cast<f4 *>(packed) + (size_t)(1 + (is_row ? q : row_entries + q)) * stride4;
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) {
            const int i4 = lg + kk * LPR;
            if (i4 < d4) e[i4] = G[kk];
        }
        if (lg == 0) e[d4] = f4{Gb, __int_as_float(id), __int_as_float(is_row ? 0 : 1), 0.f};
    }
};

template <int LPR, int NV>
__global__ __launch_bounds__(kBlock) void pack_grad_kernel(
    IdWork wk, SideBufs rs, SideBufs cs, int d4, StepConsts k, float *__restrict__ packed,
    const float *__restrict__ blockpart, int nblocks_rowpass)
{
    const int nu_r = wk.nu_r_host >= 0 ? wk.nu_r_host : wk.counts[1];
    const int nu_c = wk.nu_c_host >= 0 ? wk.nu_c_host : wk.counts[3];
    const int row_entries = (wk.sides & 1) ? nu_r : 0;
    const bool scalar_duty = for_each_id<LPR, NV>(wk, rs, cs, d4, k,
                                                  PackGrad<LPR, NV>{packed, d4, (int)(threadIdx.x % LPR), row_entries});
    if (scalar_duty) {                      // the col side carries the loss partials (glove_hyper.sides)
        float tot[kPartials];
        sum_blockpart(blockpart, nblocks_rowpass, tot);
        if (threadIdx.x == 0) {
            f4 *h = reinterpret_cast<f4 *>(packed);
            h[0] = f4{__int_as_float(row_entries), __int_as_float(nu_c), tot[3], tot[0]};
            h[1] = f4{tot[1], tot[2], 0.f, 0.f};
        }
    } else if (!(wk.sides & 2) && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        f4 *h = reinterpret_cast<f4 *>(packed);            // row side only: no loss partials travel with it
        h[0] = f4{__int_as_float(row_entries), __int_as_float(0), 0.f, 0.f};
        h[1] = f4{0.f, 0.f, 0.f, 0.f};
    }
}

struct PackedList {
    const float *entries;       // first entry behind the header (or a bare array of entries)
    const int32_t *ids;         // if not null: ids[i] replaces the id stored in entry i (owner-local indices)
    const float *header;        // if not null: entry count = row entries + col entries of this header
    int n_host;                 // else the entry count
    int side;                   // -1: every entry names its side; 0 / 1: all entries are of that side
};

struct DenseViews { float *G_R, *G_br, *G_C, *G_bc; int32_t *mark; int V_row; };

__device__ inline int packed_count(const PackedList &pl)
{
    if (!pl.header) return pl.n_host;
    return __float_as_int(pl.header[0]) + __float_as_int(pl.header[1]);
}

// mark[id]: bits 0..7 = how many lists touch the id (count_packed_kernel; 0 = not counted), bits 8.. = 1 + tag of the
// first list that stored the id's row in the dense buffer (0 = none yet)
constexpr int kMarkCountBits = 8, kMarkCountMask = (1 << kMarkCountBits) - 1;

// (count_packed_kernel below is the optional first pass over ALL lists: afterwards an id only one list touches needs no
// trip through the dense buffer — the apply kernel takes its gradient straight from that list's entry, combine skips it.)

template <int LPR, int NV>
__global__ __launch_bounds__(kBlock) void combine_packed_kernel(PackedList pl, int tag, DenseViews dv, int d4)
{
    constexpr int GPB = kBlock / LPR;
    const int lg = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const int n = packed_count(pl);
    const size_t stride4 = (size_t)d4 + 1;
    for (int i = blockIdx.x * GPB + grp; i < n; i += gridDim.x * GPB) {
        const f4 *e = reinterpret_cast<const f4 *>(pl.entries) + (size_t)i * stride4;
        const f4 x = e[d4];
        const int32_t id = pl.ids ? pl.ids[i] : __float_as_int(x.y);
        const bool is_row = (pl.side >= 0 ? pl.side : __float_as_int(x.z)) == 0;
        int32_t *mk = dv.mark + (is_row ? 0 : dv.V_row) + id;
        const int32_t m = *mk;              // every lane of the group reads it before lane 0 rewrites it below
        if ((m & kMarkCountMask) == 1) continue;        // counted, and this list alone touches the id: applied from the entry
        const int32_t seen = m >> kMarkCountBits;
        f4 g[NV];
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) {
            const int i4 = lg + kk * LPR;
            g[kk] = e[i4 < d4 ? i4 : d4 - 1];
        }
        float *Gt = is_row ? dv.G_R : dv.G_C;
        float *Gb = is_row ? dv.G_br : dv.G_bc;
        if (seen != 0) {                    // a list before this one touched the id: add behind it
            f4 old[NV];
            load_row<LPR, NV>(old, Gt, id, d4, lg);
#pragma unroll
            for (int kk = 0; kk < NV; ++kk) g[kk] = old[kk] + g[kk];
        }
        store_row<LPR, NV>(Gt, (size_t)id, d4, lg, g);
        if (lg == 0) {
            Gb[id] = seen != 0 ? Gb[id] + x.x : x.x;
            if (seen == 0) *mk = m | ((tag + 1) << kMarkCountBits);
        }
    }
}

struct PackedLists { PackedList l[8]; int n; };

__global__ __launch_bounds__(kBlock) void count_packed_kernel(PackedLists pls, DenseViews dv, int d4)
{
    const PackedList &pl = pls.l[blockIdx.y];
    const int n = packed_count(pl);
    const size_t stride = ((size_t)d4 + 1) * 4;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float *x = pl.entries + (size_t)i * stride + (size_t)d4 * 4;
        const int32_t id = pl.ids ? pl.ids[i] : __float_as_int(x[1]);
        const bool is_row = (pl.side >= 0 ? pl.side : __float_as_int(x[2])) == 0;
        atomicAdd(dv.mark + (is_row ? 0 : dv.V_row) + id, 1);
    }
}

// More than eight lists and no summed tail from the caller: the loss partials of ALL headers are added up first, eight
// lists per launch in list order, into four scratch floats (the reserved half of G_flat's tail).
__global__ void sum_headers_kernel(PackedLists pls, float *__restrict__ acc4, int first)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    if (!first) { t[0] = acc4[0]; t[1] = acc4[1]; t[2] = acc4[2]; t[3] = acc4[3]; }
    for (int r = 0; r < pls.n; ++r) {
        const float *h = pls.l[r].header;
        if (h) { t[0] += h[2]; t[1] += h[3]; t[2] += h[4]; t[3] += h[5]; }
    }
    acc4[0] = t[0]; acc4[1] = t[1]; acc4[2] = t[2]; acc4[3] = t[3];
}

// Keras-legacy Nadam on the same two-part kernel (the legacy optimizer's sparse path decays m and v over the whole variable like
// its Adam, but only the touched rows move: optimizer_v2/nadam.py `_resource_apply_sparse`; restated in oracle/glove_ref.py _nadam).
// The momentum cache — the running product of the schedule u_i = beta1 (1 - 0.5 0.96^(0.004 i)) — lives in scalars[4 + parity]:
// step t reads slot (t - 1) & 1 and its scalar epilogue writes slot t & 1, so no workgroup of step t sees the new value.
struct NadamConsts { float lr, eps, b1, b2, one_minus_u_t, u_t1, om_new, om_next, v_den, sched_new; };
__device__ inline NadamConsts nadam_consts(float lr, float eps, float b1, float b2, double ln_beta2, int64_t t, const float *scalars)
{
#pragma clang fp contract(off)
    const double ln096 = -0.040821994520255166;                 // ln 0.96
    const float u_t = b1 * (1.0f - 0.5f * expf((float)(0.004 * (double)t * ln096)));
    const float u_t1 = b1 * (1.0f - 0.5f * expf((float)(0.004 * (double)(t + 1) * ln096)));
    const float cache = scalars[4 + (int)((t - 1) & 1)];
    NadamConsts k;
    k.lr = lr; k.eps = eps; k.b1 = b1; k.b2 = b2;
    k.sched_new = cache * u_t;
    k.one_minus_u_t = 1.0f - u_t;
    k.u_t1 = u_t1;
    k.om_new = 1.0f - k.sched_new;
    k.om_next = 1.0f - k.sched_new * u_t1;
    k.v_den = -expm1f((float)((double)t * ln_beta2));
    return k;
}
__device__ inline void nadam_elem(float &w, float &m, float &v, float g, const NadamConsts &k)
{
#pragma clang fp contract(off)
    m = m * k.b1 + (1.0f - k.b1) * g;
    v = v * k.b2 + (1.0f - k.b2) * g * g;
    const float m_bar = k.one_minus_u_t * (g / k.om_new) + k.u_t1 * (m / k.om_next);
    w -= k.lr * m_bar / (sqrtf(v / k.v_den) + k.eps);
}

// ---- the element updates of the per-row Keras optimizers (SGD, Adamax, Adadelta, Ftrl): include/glove_hip.h glove_hyper.optimizer;
// used by the apply epilogues of the single-GPU step (SparseOptApply) and of the touched-rows exchange (apply_packed_kernel)
struct OptConsts { float lr, eps, momentum, lr_t, b1, b2; int nesterov; float rho; };
template <int OPT> struct OptSlots {
    static constexpr bool two = OPT == GLOVE_OPT_ADAMAX || OPT == GLOVE_OPT_ADADELTA || OPT == GLOVE_OPT_FTRL || OPT == GLOVE_OPT_NADAM;
};
template <int OPT>
struct OptElem {
    // Adadelta (kernel SparseApplyAdadelta): a = accum_grad, u = accum_var
    __device__ static void adadelta(float &w, float &a, float &u, float g, const OptConsts &o)
    {
#pragma clang fp contract(off)
        a = a * o.rho + g * g * (1.0f - o.rho);
        const float upd = sqrtf(u + o.eps) / sqrtf(a + o.eps) * g;
        w -= upd * o.lr;
        u = u * o.rho + upd * upd * (1.0f - o.rho);
    }
    // Ftrl at its Keras defaults (kernel FtrlCompute; lr_power -0.5, l1 = l2 = 0): a = accumulator, z = linear
    __device__ static void ftrl(float &w, float &a, float &z, float g, const OptConsts &o)
    {
#pragma clang fp contract(off)
        const float na = a + g * g;
        z += g - (sqrtf(na) - sqrtf(a)) / o.lr * w;
        w = fabsf(z) > 0.f ? -z / (sqrtf(na) / o.lr) : 0.f;
        a = na;
    }
    __device__ static void sgd(float &w, float &a, float g, const OptConsts &o)
    {
#pragma clang fp contract(off)
        if (o.momentum == 0.f) { w -= o.lr * g; return; }
        a = a * o.momentum - o.lr * g;
        w += o.nesterov ? a * o.momentum - o.lr * g : a;
    }
    __device__ static void adamax(float &w, float &m, float &v, float g, const OptConsts &o)
    {
#pragma clang fp contract(off)
        m = o.b1 * m + (1.0f - o.b1) * g;
        v = fmaxf(o.b2 * v, fabsf(g));
        w -= o.lr_t * m / (v + o.eps);
    }
    // (nk: Nadam's constants of this step — the touched-rows exchange only: apply_packed_kernel; a = m, b = v)
    __device__ static void one(float &w, float &a, float &b, float g, const OptConsts &o, const NadamConsts &nk)
    {
        if (OPT == GLOVE_OPT_SGD) sgd(w, a, g, o);
        else if (OPT == GLOVE_OPT_ADAMAX) adamax(w, a, b, g, o);
        else if (OPT == GLOVE_OPT_ADADELTA) adadelta(w, a, b, g, o);
        else if (OPT == GLOVE_OPT_NADAM) nadam_elem(w, a, b, g, nk);
        else ftrl(w, a, b, g, o);
    }
    __device__ static void one(float &w, float &a, float &b, float g, const OptConsts &o) { one(w, a, b, g, o, NadamConsts{}); }
};

// A lane group walks entries gg, gg + TG, gg + 2 TG, ... (TG lane groups in the launch).  Looked up one entry at a time
// that is three dependent round trips per entry (id -> mark -> rows) and ~16 entries per group at C5, a latency chain.
// Instead the group's LPR lanes look LPR entries up at once — lane t the id, side, bias gradient and mark of the t-th
// of them — and the rows are then moved two entries at a time, what each needs handed round by lane shuffles: one
// round trip per pair of entries.
// OPT: GLOVE_OPT_ADAGRAD, or one of the per-row Keras optimizers (OptElem: SGD, Adamax, Adadelta, Ftrl) — only touched rows move
// under all of them, so the exchange is the same and the epilogue differs.  s2: the second slot of every variable where the
// optimizer has one (scalars[2] for the global bias).
struct SlotTwo { float *R, *C, *br, *bc; };

// Nadam on the touched-rows exchange: m and v of every row NO rank touched decay (Keras' legacy sparse Nadam decays them over
// the whole variable); the rows some rank touched — marked by count_packed_kernel — are apply_packed_kernel's, the next launch.
template <int LPR, int NV>
__global__ __launch_bounds__(kBlock) void nadam_decay_unmarked_kernel(DenseViews dv, SideBufs rs, SideBufs cs, SlotTwo s2, int d4,
                                                                      float b1, float b2, int V)
{
    constexpr int GPB = kBlock / LPR;
    const int lg = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const int total = dv.V_row + V;
    for (int v = blockIdx.x * GPB + grp; v < total; v += gridDim.x * GPB) {
        if (dv.mark[v] & kMarkCountMask) continue;
        const bool is_row = v < dv.V_row;
        const int id = is_row ? v : v - dv.V_row;
        const SideBufs &sb = is_row ? rs : cs;
        float *S2 = is_row ? s2.R : s2.C, *S2b = is_row ? s2.br : s2.bc;
        f4 M[NV], Vv[NV];
        load_row<LPR, NV>(M, sb.S1, id, d4, lg);
        load_row<LPR, NV>(Vv, S2, id, d4, lg);
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) { M[kk] = b1 * M[kk]; Vv[kk] = b2 * Vv[kk]; }
        store_row<LPR, NV>(sb.S1, (size_t)id, d4, lg, M);
        store_row<LPR, NV>(S2, (size_t)id, d4, lg, Vv);
        if (lg == 0) { sb.S1b[id] = b1 * sb.S1b[id]; S2b[id] = b2 * S2b[id]; }
    }
}

template <int LPR, int NV, int OPT>
__global__ __launch_bounds__(kBlock) void apply_packed_kernel(
    PackedLists pls, DenseViews dv, SideBufs rs, SideBufs cs, SlotTwo s2, int d4, StepConsts k, OptConsts o, double ln_beta1,
    const int64_t *__restrict__ step, int first_tag, const float *__restrict__ tail_in, float *__restrict__ scalars,
    float *__restrict__ loss_out, int do_scalars)
{
    constexpr int GPB = kBlock / LPR;
    constexpr bool kAdagrad = OPT == GLOVE_OPT_ADAGRAD;
    constexpr bool kTwo = !kAdagrad && OptSlots<OPT>::two;
    const bool slots = kAdagrad || !(OPT == GLOVE_OPT_SGD && o.momentum == 0.f);     // (plain SGD keeps no slot)
    // t = global_step as the passes of this step left it (Adamax: lr_t = lr / (1 - beta1^t), as apply_sparse_opt_kernel)
    if (OPT == GLOVE_OPT_ADAMAX) o.lr_t = o.lr / -expm1f((float)((double)(*step) * ln_beta1));
    // Nadam (the ranks' touched rows move; every other row's m and v have decayed in nadam_decay_unmarked_kernel, the launch before):
    // the constants of step t = global_step as the passes left it; ln_beta1 carries ln beta2 here
    NadamConsts nk = {};
    if (OPT == GLOVE_OPT_NADAM) nk = nadam_consts(o.lr, o.eps, o.b1, o.b2, ln_beta1, *step, scalars);
    const int lg = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const size_t stride4 = (size_t)d4 + 1;
    const PackedList &pl = pls.l[blockIdx.y];
    const int tag = first_tag + blockIdx.y;
    const long n = packed_count(pl);
    const long TG = (long)gridDim.x * GPB, gg = (long)blockIdx.x * GPB + grp;
    const f4 *entries = reinterpret_cast<const f4 *>(pl.entries);
    for (long t0 = 0; gg + t0 * TG < n; t0 += LPR) {
        const long il = gg + (t0 + lg) * TG;
        int32_t id_l = 0;
        int row_l = 1, todo_l = 0;      // todo: 0 nothing (past the end, or another list stored the id first: its entry
        float gb_l = 0.f;               // applies it), 1 this entry is the id's only one: its row IS the sum, 2 the dense buffer has the sum
        if (il < n) {
            const f4 x = entries[il * stride4 + d4];
            id_l = pl.ids ? pl.ids[il] : __float_as_int(x.y);
            row_l = (pl.side >= 0 ? pl.side : __float_as_int(x.z)) == 0;
            gb_l = x.x;
            const int32_t m = dv.mark[(row_l ? 0 : dv.V_row) + id_l];
            todo_l = (m & kMarkCountMask) == 1 ? 1 : ((m >> kMarkCountBits) == tag + 1 ? 2 : 0);
        }
        for (int tt = 0; tt < LPR; tt += 2) {
            if (gg + (t0 + tt) * TG >= n) break;
            int todo[2];
            int32_t id[2];
            bool is_row[2];
            float Gb[2], bval[2], Ab[2], Bb[2];
            f4 G[2][NV], Wv[2][NV], A[2][NV], Bv[2][kTwo ? NV : 1];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                todo[e] = __shfl(todo_l, tt + e, LPR);
                id[e] = __shfl(id_l, tt + e, LPR);
                is_row[e] = __shfl(row_l, tt + e, LPR) != 0;
                Gb[e] = __shfl(gb_l, tt + e, LPR);
                Ab[e] = Bb[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (!todo[e]) continue;
                const float *W = is_row[e] ? rs.W : cs.W, *S1 = is_row[e] ? rs.S1 : cs.S1;
                if (todo[e] == 1) {
                    const f4 *row = entries + (size_t)(gg + (t0 + tt + e) * TG) * stride4;
#pragma unroll
                    for (int kk = 0; kk < NV; ++kk) {
                        const int i4 = lg + kk * LPR;
                        const f4 v = row[i4 < d4 ? i4 : d4 - 1];
                        G[e][kk] = (i4 < d4) ? v : f4{0.f, 0.f, 0.f, 0.f};
                    }
                } else {
                    load_row<LPR, NV>(G[e], is_row[e] ? dv.G_R : dv.G_C, id[e], d4, lg);
                    Gb[e] = (is_row[e] ? dv.G_br : dv.G_bc)[id[e]];
                }
                load_row<LPR, NV>(Wv[e], W, id[e], d4, lg);
                bval[e] = (is_row[e] ? rs.bias : cs.bias)[id[e]];
                if (slots) {
                    load_row<LPR, NV>(A[e], S1, id[e], d4, lg);
                    Ab[e] = (is_row[e] ? rs.S1b : cs.S1b)[id[e]];
                }
                if constexpr (kTwo) {
                    load_row<LPR, NV>(Bv[e], is_row[e] ? s2.R : s2.C, id[e], d4, lg);
                    Bb[e] = (is_row[e] ? s2.br : s2.bc)[id[e]];
                }
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (!todo[e]) continue;
                float *W = is_row[e] ? rs.W : cs.W, *S1 = is_row[e] ? rs.S1 : cs.S1;
                if constexpr (kAdagrad) {
#pragma unroll
                    for (int kk = 0; kk < NV; ++kk) adagrad_vec(Wv[e][kk], A[e][kk], G[e][kk], k.lr, k.eps);
                } else {
#pragma unroll
                    for (int kk = 0; kk < NV; ++kk) {
                        float w[4] = {Wv[e][kk].x, Wv[e][kk].y, Wv[e][kk].z, Wv[e][kk].w};
                        float a[4] = {A[e][kk].x, A[e][kk].y, A[e][kk].z, A[e][kk].w};
                        float bb[4] = {0.f, 0.f, 0.f, 0.f};
                        if constexpr (kTwo) { bb[0] = Bv[e][kk].x; bb[1] = Bv[e][kk].y; bb[2] = Bv[e][kk].z; bb[3] = Bv[e][kk].w; }
                        const float g[4] = {G[e][kk].x, G[e][kk].y, G[e][kk].z, G[e][kk].w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) OptElem<OPT>::one(w[i], a[i], bb[i], g[i], o, nk);
                        Wv[e][kk] = f4{w[0], w[1], w[2], w[3]};
                        A[e][kk] = f4{a[0], a[1], a[2], a[3]};
                        if constexpr (kTwo) Bv[e][kk] = f4{bb[0], bb[1], bb[2], bb[3]};
                    }
                }
                if (slots) store_row<LPR, NV>(S1, (size_t)id[e], d4, lg, A[e]);
                if constexpr (kTwo) store_row<LPR, NV>(is_row[e] ? s2.R : s2.C, (size_t)id[e], d4, lg, Bv[e]);
                store_row<LPR, NV>(W, (size_t)id[e], d4, lg, Wv[e]);
                if (lg == 0) {
                    if constexpr (kAdagrad) adagrad_elem(bval[e], Ab[e], Gb[e], k.lr, k.eps);
                    else OptElem<OPT>::one(bval[e], Ab[e], Bb[e], Gb[e], o, nk);
                    if (slots) (is_row[e] ? rs.S1b : cs.S1b)[id[e]] = Ab[e];
                    if constexpr (kTwo) (is_row[e] ? s2.br : s2.bc)[id[e]] = Bb[e];
                    (is_row[e] ? rs.bias : cs.bias)[id[e]] = bval[e];
                    dv.mark[(is_row[e] ? 0 : dv.V_row) + id[e]] = 0;
                }
            }
        }
    }
    if (do_scalars && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        // loss partials: given already summed over the ranks (tail_in), or summed here over the lists' headers in list order
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        if (tail_in) {
            t[0] = tail_in[0]; t[1] = tail_in[1]; t[2] = tail_in[2]; t[3] = tail_in[3];
        } else {
            for (int r = 0; r < pls.n; ++r) {
                const float *h = pls.l[r].header;
                if (h) { t[0] += h[2]; t[1] += h[3]; t[2] += h[4]; t[3] += h[5]; }
            }
        }
        const float g = scalars[0];
        const float tot[kPartials] = {t[1], t[2], t[3], t[0]};
        float loss, L, reg;
        loss_from_partials(tot, k, g, loss, L, reg);
        const float dg = t[0] + 2.0f * k.m * k.l2 * g;
        if constexpr (kAdagrad) {
            adagrad_elem(scalars[0], scalars[1], dg, k.lr, k.eps);
        } else {
            float gn = g, a = scalars[1], b = scalars[2];
            OptElem<OPT>::one(gn, a, b, dg, o, nk);
            scalars[0] = gn; scalars[1] = a; scalars[2] = b;
            if (OPT == GLOVE_OPT_NADAM) scalars[4 + (int)(*step & 1)] = nk.sched_new;      // the momentum cache (see nadam_consts)
        }
        if (loss_out) { loss_out[0] = loss; loss_out[1] = L; loss_out[2] = reg; loss_out[3] = t[0]; }
    }
}

// ------------------------------------------------------------------------------------------
// The sparse Adagrad step for the latency-bound regime (GLOVE_STEP_TAGGED): ONE launch does all the work on the rows — forward,
// gradients AND the Adagrad applies — and a one-workgroup epilogue launch does the once-per-step scalars.  The reference's
// default batch of 1,024 pairs (reference configs/app.ini:39-53) runs the two-launch form as two ramps, a kernel boundary and
// four dependent memory round trips (record -> rows -> partial rows | id record -> partial + table rows -> rows), nothing in
// them is bandwidth; here a chunk is record -> rows -> new row.
// What lets one launch both read every row as it was when the step began and update rows: both tables are twinned and
// step-tagged (glove_tables.R_tag / C_tag).  tag[u] = 0, or (1 + step that wrote row u) << 1 | the copy it wrote.  During step t
// a reader takes the copy the tag names unless the tag says "written in step t": then it takes the OTHER copy — the pre-step
// row, which nobody touches during step t.  The writer (the one lane group that owns row u in this step) puts the new row into
// the copy it did not read and then stores the tag.  A racing reader sees the old tag or the new one; either way it is led to
// the pre-step row.  No barrier, no fence, no atomic (a device-scope fence costs 3 - 6 us on this chip: measured).
//   * the tag is not waited for: the own row and the first trip's partner rows are requested in BOTH copies together with
//     their tags (one round trip; the tables of this regime live in the caches) and the copy is chosen in registers;
//   * an id's chunks are all done by the lane group that holds its FIRST chunk, one after the other (the groups that were
//     dealt its other chunks skip them): per-chunk sums added in chunk order — for ids up to plan.heavy_chunks chunks the very
//     order of the two-launch form's apply, bit for bit — then G = sum + activity-L2 term and Adagrad, expression for
//     expression as AdagradApply.  No partial rows in memory.  (An id of many chunks is a long serial chain: the form is for
//     batches whose ids have few chunks — GLOVE_STEP_AUTO takes it up to 2,048 pairs.)
//   * every row-side workgroup leaves its loss partials; scalars_kernel (one workgroup, behind it on the stream) sums them in
//     the fixed order of the two-launch form, updates the global bias, writes the loss and advances global_step — the one
//     dependency between consecutive steps that needs all pairs, i.e. a grid-wide reduction: a kernel boundary is the
//     cheapest one there is.
// ------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct OneSide {
    const int32_t *crec;
    float *own, *own_bias;              // this side's table and bias vector, twinned
    const float *other, *other_bias;    // the partner table, twinned
    float *S1, *S1b;                    // Adagrad accumulators (never twinned: only the id's owner touches them)
    uint64_t *own_tag;
    const uint64_t *other_tag;
    int own_twin, other_twin;           // rows between the two copies
    int n_host, count_index, capP, cap_chunks;
};

// which copy (0 / 1) holds the row as it was when step t began
__device__ inline uint32_t pre_step_copy(uint64_t tag, uint64_t t1 /* 1 + t */) { return (uint32_t)(tag & 1ull) ^ ((tag >> 1) == t1 ? 1u : 0u); }

// A chain record = what one step of a chain leaves for the next: [0] global bias and [1] its accumulator as they were when the
// step BEGAN, [8 + 4 b ..] the loss partials of its row-side workgroup b.
constexpr int kChainHead = 8;

template <int LPR, int NV, bool FULL>
__global__ __launch_bounds__(kBlock, NV <= 2 ? 2 : 1) void tagged_step_kernel(
    const int32_t *__restrict__ counts, OneSide rowside, OneSide colside, int row_blocks, const float *__restrict__ scalars,
    const int64_t *__restrict__ step, int d4, float inv_batch, int head, float neg_factor, StepConsts kc,
    const float *__restrict__ prev, int prev_blocks, float *__restrict__ mine, int chain_i)
{
    constexpr int GPB = kBlock / LPR;
    constexpr int U = PassUnroll<NV>::value;
    constexpr int kRecStride = 4 + 3 * kChunkMax + 4;
    __shared__ __attribute__((aligned(16))) uint32_t fld_raw[GPB * kRecStride];
    uint32_t *rec = fld_raw + (threadIdx.x / LPR) * kRecStride;
    const int lg = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const bool is_row = (int)blockIdx.x < row_blocks;
    const OneSide &sd = is_row ? rowside : colside;
    const int bid = is_row ? blockIdx.x : blockIdx.x - row_blocks;
    const int nblk = is_row ? row_blocks : gridDim.x - row_blocks;
    // the number of chunks is not waited for either (a plan refilled on the device keeps it in memory), nor the bias below: the
    // group's first record is requested before anything else (records exist up to the plan's capacity; one that lies behind
    // the side's last chunk is dropped unread)
    constexpr int kPre = (1 + 3 * kChunkMax / 4 + LPR - 1) / LPR;
    u32x4 pre[kPre];       // (a native vector type: an array of the struct uint4 went through scratch at kPre = 4)
    {
        const int jf = bid * GPB + grp;
        const uint4 *rp = reinterpret_cast<const uint4 *>(sd.crec) + (size_t)(jf < sd.cap_chunks ? jf : sd.cap_chunks - 1) * rec_stride_q(sd.capP);
        const int rq0 = 1 + 3 * sd.capP / 4;
#pragma unroll
        for (int x = 0; x < kPre; ++x) {
            const int f = lg + x * LPR;
            pre[x] = reinterpret_cast<const u32x4 *>(rp)[rec_gq(f < rq0 ? f : 0)];
        }
    }
    const int n_chunks = sd.n_host >= 0 ? sd.n_host : counts[sd.count_index];
    // global_step stands still during a chain (its last launch's epilogue advances it): this is step *step + chain_i
    const uint64_t t1 = (uint64_t)(*step) + (uint64_t)chain_i + 1ull;
    float part[kPartials] = {0.f, 0.f, 0.f, 0.f};
    const int capP = sd.capP;
    const int rq = 1 + 3 * capP / 4;
    const int sq = rec_stride_q(capP);
    // ---- the global bias of THIS step.  The one thing a step needs of ALL pairs of the step before is sum e (the global bias'
    // gradient): a grid-wide reduction.  Instead of a launch (or a device-scope fence: 3 - 6 us) between the steps, every
    // workgroup of step i derives the bias itself — from the bias step i - 1 began with and the loss partials its row-side
    // workgroups left (prev), summed in the fixed order of the two-launch form: same bits in every workgroup — and leaves what
    // it began with, and its own partials, in a record of its own (mine), which nobody of this launch reads.
    __shared__ float s_g[2];
    if (prev) {
        float tot[kPartials];
        sum_blockpart(prev + kChainHead, prev_blocks, tot);
        if (threadIdx.x == 0) {
            const float g0 = prev[0];
            const float dg = tot[3] + 2.0f * kc.m * kc.l2 * g0;
            float gn = g0, Ag = prev[1];
            adagrad_elem(gn, Ag, dg, kc.lr, kc.eps);
            s_g[0] = gn; s_g[1] = Ag;
        }
    } else if (threadIdx.x == 0) {
        s_g[0] = scalars[0]; s_g[1] = scalars[1];
    }
    __syncthreads();
    const float g = s_g[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) { mine[0] = g; mine[1] = s_g[1]; }
    float *blockpart = mine + kChainHead;

    for (int j0 = bid * GPB + grp; j0 < n_chunks; j0 += nblk * GPB) {
        // ---- round trip 1: the record (descriptor + pair fields) in contiguous 16-B loads -> LDS as is
        {
            uint4 *lrec = reinterpret_cast<uint4 *>(rec);
            if (j0 == bid * GPB + grp) {                        // (requested at the kernel's start)
#pragma unroll
                for (int x = 0; x < kPre; ++x) {
                    const int f = lg + x * LPR;
                    if (f < rq) reinterpret_cast<u32x4 *>(lrec)[f] = pre[x];
                }
            } else {
                const uint4 *rp = reinterpret_cast<const uint4 *>(sd.crec) + (size_t)j0 * sq;
                for (int f = lg; f < rq; f += LPR) lrec[f] = rp[rec_gq(f)];
            }
        }
        uint4 hdr = *reinterpret_cast<const uint4 *>(rec);      // same wave wrote it: LDS ops of one wave complete in order
        if (!(hdr.w >> 31)) continue;                           // not the first chunk of its id: the group that holds the first one does it
        const int32_t u = (int32_t)hdr.x;
        const int chunks = 1 + (int)(hdr.w & 0x7fffffffu);
        // ---- round trip 2: the own row in both copies, its tag, the accumulator (and, below, the first trip's partner rows)
        f4 r[NV], r1[NV], A[NV], G[NV];
        load_row<LPR, NV>(r, sd.own, u, d4, lg);
        load_row<LPR, NV>(r1, sd.own, u + sd.own_twin, d4, lg);
        const uint64_t own_tg = sd.own_tag[u];
        const float ob0 = sd.own_bias[u], ob1 = sd.own_bias[u + sd.own_twin];
        load_row<LPR, NV>(A, sd.S1, u, d4, lg);
        float Ab = sd.S1b[u];
#pragma unroll
        for (int k = 0; k < NV; ++k) G[k] = f4{0.f, 0.f, 0.f, 0.f};
        float Gb = 0.f, own_b = 0.f, bg = 0.f;
        uint32_t own_copy = 0;
        int pairs = 0;
        for (int ch = 0; ch < chunks; ++ch) {
            const int j = j0 + ch;
            if (ch > 0) {                                       // the id's next chunk: its record (a round trip of its own)
                const uint4 *rp = reinterpret_cast<const uint4 *>(sd.crec) + (size_t)j * sq;
                uint4 *lrec = reinterpret_cast<uint4 *>(rec);
                for (int f = lg; f < rq; f += LPR) lrec[f] = rp[rec_gq(f)];
                hdr = *reinterpret_cast<const uint4 *>(rec);
            }
            const int n = (int)hdr.y;
            pairs += n;
            // the tags of all partners of the chunk: lane t looks pair t's up and leaves the row that is current in the record
            // (consumed from the second trip on; the first trip does not wait for it)
            uint64_t ptag[(kChunkMax + LPR - 1) / LPR];
#pragma unroll
            for (int x = 0; x < (kChunkMax + LPR - 1) / LPR; ++x) {
                const int t = lg + x * LPR;
                ptag[x] = t < n ? sd.other_tag[rec[rec_pair(t)]] : 0ull;
            }
            f4 acc[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
            float se_c = 0.f, cc_sum = 0.f, bsq = 0.f, ed = 0.f;
            for (int q0 = 0; q0 < n; q0 += U) {
                int32_t col[U];
                float w2[U], yq[U];
#pragma unroll
                for (int a4 = 0; a4 < U; a4 += 4) {
                    const int rp0 = rec_pair(q0 + a4);
                    const uint4 pc = *reinterpret_cast<const uint4 *>(&rec[rp0]);
                    const uint4 pw = *reinterpret_cast<const uint4 *>(&rec[rp0 + kRecPad]);
                    const uint4 py = *reinterpret_cast<const uint4 *>(&rec[rp0 + 2 * kRecPad]);
                    col[a4] = (int32_t)pc.x; col[a4 + 1] = (int32_t)pc.y; col[a4 + 2] = (int32_t)pc.z; col[a4 + 3] = (int32_t)pc.w;
                    const float sc2 = 2.0f * inv_batch;
                    w2[a4] = sc2 * __uint_as_float(pw.x); w2[a4 + 1] = sc2 * __uint_as_float(pw.y);
                    w2[a4 + 2] = sc2 * __uint_as_float(pw.z); w2[a4 + 3] = sc2 * __uint_as_float(pw.w);
                    yq[a4] = __uint_as_float(py.x); yq[a4 + 1] = __uint_as_float(py.y);
                    yq[a4 + 2] = __uint_as_float(py.z); yq[a4 + 3] = __uint_as_float(py.w);
                }
                f4 c[U][NV];
                float bcv[U];
                if (q0 == 0) {
                    // first trip: both copies of every partner row travel with the tags; the choice happens in registers
                    f4 c1[U][NV];
                    float b0[U], b1[U];
#pragma unroll
                    for (int a = 0; a < U; ++a) {
                        load_row_fast<LPR, NV, FULL>(c[a], sd.other, col[a], d4, lg);
                        load_row_fast<LPR, NV, FULL>(c1[a], sd.other, col[a] + sd.other_twin, d4, lg);
                        b0[a] = b1[a] = 0.f;
                        if (lg == 0) {
                            b0[a] = sd.other_bias[col[a]];
                            b1[a] = sd.other_bias[col[a] + sd.other_twin];
                        }
                    }
                    if (ch == 0) {                              // (the own row's copy: known with this trip)
                        own_copy = pre_step_copy(own_tg, t1);
#pragma unroll
                        for (int k = 0; k < NV; ++k) r[k] = own_copy ? r1[k] : r[k];
                        own_b = own_copy ? ob1 : ob0;
                        bg = own_b + g;
                    }
                    // the partners' tags, handed round: pair a of this trip was looked up by lane a % LPR (slot a / LPR)
#pragma unroll
                    for (int a = 0; a < U; ++a) {
                        const uint64_t tg = __shfl(ptag[a / LPR], (int)((threadIdx.x & 63) / LPR * LPR + a % LPR), 64);
                        const bool second = pre_step_copy(tg, t1) != 0;
#pragma unroll
                        for (int k = 0; k < NV; ++k) c[a][k] = second ? c1[a][k] : c[a][k];
                        bcv[a] = second ? b1[a] : b0[a];
                    }
                    // later trips read the current row straight away: the record's ids become row numbers
#pragma unroll
                    for (int x = 0; x < (kChunkMax + LPR - 1) / LPR; ++x) {
                        const int t = lg + x * LPR;
                        if (t < n && t >= U) rec[rec_pair(t)] += pre_step_copy(ptag[x], t1) ? (uint32_t)sd.other_twin : 0u;
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < U; ++a) {
                        load_row_fast<LPR, NV, FULL>(c[a], sd.other, col[a], d4, lg);
                        bcv[a] = 0.f;
                        if (lg == 0) bcv[a] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(sd.other_bias) + (uint32_t)col[a] * 4u);
                    }
                }
                float dp[U], cc[U];
#pragma unroll
                for (int a = 0; a < U; ++a) {
                    dp[a] = 0.f; cc[a] = 0.f;
#pragma unroll
                    for (int k = 0; k < NV; ++k) { dp[a] += dot4(r[k], c[a][k]); cc[a] += dot4(c[a][k], c[a][k]); }
                    dp[a] += bcv[a];
                }
#pragma unroll
                for (int a = 0; a < U; ++a) dp[a] = dpp_add<0xB1>(dp[a]);
#pragma unroll
                for (int a = 0; a < U; ++a) dp[a] = dpp_add<0x4E>(dp[a]);
#pragma unroll
                for (int a = 0; a < U; ++a) dp[a] = dpp_add<0x141>(dp[a]);
                if (LPR >= 16) {
#pragma unroll
                    for (int a = 0; a < U; ++a) dp[a] = dpp_add<0x140>(dp[a]);
                }
                if (LPR >= 32) {
#pragma unroll
                    for (int a = 0; a < U; ++a) dp[a] += __shfl_xor(dp[a], 16, 64);
                }
                if (LPR >= 64) {
#pragma unroll
                    for (int a = 0; a < U; ++a) dp[a] += __shfl_xor(dp[a], 32, 64);
                }
#pragma unroll
                for (int a = 0; a < U; ++a) {
                    const float valid = (q0 + a < n) ? 1.0f : 0.f;
                    float e;
                    if (head == GLOVE_HEAD_REGRESSION) {
                        const float diff = (dp[a] + bg) - yq[a];
                        e = w2[a] * diff;
                        ed += e * diff;
                    } else {
                        const float p = dp[a] + bg;
                        const float en = expf(-fabsf(p));
                        const float s = (p >= 0.f ? 1.0f : en) / (1.0f + en);
                        const float lse = log1pf(en);
                        const float wn = 2.0f * inv_batch * neg_factor * valid * yq[a];
                        e = 0.5f * (w2[a] * (s - 1.0f) + wn * s);
                        ed += w2[a] * (fmaxf(-p, 0.f) + lse) + wn * (fmaxf(p, 0.f) + lse);
                    }
#pragma unroll
                    for (int k = 0; k < NV; ++k) acc[k] += e * c[a][k];
                    se_c += e;
                    cc_sum += valid * cc[a];
                    bsq += valid * bcv[a] * bcv[a];
                }
            }
            if (is_row) {
                float rr = 0.f;
#pragma unroll
                for (int k = 0; k < NV; ++k) rr += dot4(r[k], r[k]);
                part[1] += cc_sum + (float)n * rr;
                if (lg == 0) {
                    part[0] += ed;
                    part[2] += bsq + (float)n * own_b * own_b;
                    part[3] += se_c;
                }
            }
            // the id's sums: chunk sums in chunk order (the first one taken as it is, like the apply launch's traversal)
            if (ch == 0) {
#pragma unroll
                for (int k = 0; k < NV; ++k) G[k] = acc[k];
                Gb = se_c;
            } else {
#pragma unroll
                for (int k = 0; k < NV; ++k) G[k] += acc[k];
                Gb += se_c;
            }
        }
        // ---- G = summed gradient + activity-L2 term, then Adagrad (for_each_id + AdagradApply, expression for expression)
        {
            float bval = own_b;
            const float cnt = (float)pairs;
            const float kcn = kc.kappa * cnt;
#pragma unroll
            for (int k = 0; k < NV; ++k) G[k] += kcn * r[k];
            Gb += kc.kappa_b * cnt * bval;
#pragma unroll
            for (int k = 0; k < NV; ++k) adagrad_vec(r[k], A[k], G[k], kc.lr, kc.eps);
            store_row<LPR, NV>(sd.S1, (size_t)u, d4, lg, A);
            const int32_t new_at = u + (own_copy ? 0 : sd.own_twin);      // the copy nobody reads during this step
            store_row<LPR, NV>(sd.own, (size_t)new_at, d4, lg, r);
            if (lg == 0) {
                adagrad_elem(bval, Ab, Gb, kc.lr, kc.eps);
                sd.S1b[u] = Ab;
                sd.own_bias[new_at] = bval;
                sd.own_tag[u] = t1 << 1 | (uint64_t)(own_copy ^ 1u);
            }
        }
    }
    if (is_row) {
        part[0] *= 0.5f / inv_batch;        // sum e diff = 2 inv_batch sum w diff^2
        block_partials_store(part, blockpart);
    }
}

// The end of a chain of n tagged steps: the last step's loss partials -> the loss scalars and the global bias the NEXT step
// begins with, into the tables; global_step advances by n.  One workgroup.
__global__ __launch_bounds__(kBlock) void tagged_flush_kernel(float *__restrict__ scalars, int64_t *__restrict__ step,
                                                              const float *__restrict__ last, int nblocks, int n, StepConsts k,
                                                              float *__restrict__ loss_out)
{
    float tot[kPartials];
    sum_blockpart(last + kChainHead, nblocks, tot);
    if (threadIdx.x == 0) {
        const float g = last[0];
        float loss, L, reg;
        loss_from_partials(tot, k, g, loss, L, reg);
        const float dg = tot[3] + 2.0f * k.m * k.l2 * g;
        float gn = g, Ag = last[1];
        adagrad_elem(gn, Ag, dg, k.lr, k.eps);
        scalars[1] = Ag;
        scalars[0] = gn;
        if (loss_out) { loss_out[0] = loss; loss_out[1] = L; loss_out[2] = reg; loss_out[3] = tot[3]; }
        *step += n;
    }
}

// ------------------------------------------------------------------------------------------
// Keras-legacy Adam in ONE launch per step (the reference's default optimizer at its default batch: configs/app.ini:39-53),
// on twinned tables.  The legacy optimizer's sparse path rewrites EVERY row every step (m *= b1, v *= b2, var -= lr_t
// m / (sqrt(v) + eps) over the whole table, a11), so the twin needs no tags: a step reads copy c of both tables and writes
// copy 1 - c of both, all rows — the batch's rows by the lane group that holds the first chunk of the id (gradient, then
// Adam, as in the tagged Adagrad step), every other row by a sweep that finds the batch's ids in the plan's bitmaps
// (glove_plan.r_mark / c_mark) and leaves them alone.  scalars[3] says which copy is current between chains.  The global bias
// travels through chain records like the tagged Adagrad step's ([0] bias, [1] m, [2] v as the step began).
// ------------------------------------------------------------------------------------------
constexpr int kSweepRows = 2;       // table rows a sweep lane group keeps in flight (adam_fused_kernel, nadam_fused_kernel)
// ... in tagged_adam_kernel: four where the registers allow (d = 64: 10.8 -> 10.2 us per step; six spill: 14.2)
constexpr int tagged_sweep_rows(int nv) { return nv <= 2 ? 4 : 2; }

struct AdamSide {
    const int32_t *crec;
    float *own, *own_bias;              // this side's table and bias vector, twinned
    const float *other, *other_bias;    // the partner table, twinned
    float *S1, *S1b, *S2, *S2b;         // m and v (never twinned: only the row's owner or sweeper touches them)
    const uint32_t *mark;               // bitmap of the batch's ids of this side
    int own_twin, other_twin;           // rows between the two copies = rows of the table
    int n_host, count_index, capP, cap_chunks;
};

template <int LPR, int NV, bool FULL>
__global__ __launch_bounds__(kBlock, NV <= 2 ? 2 : 1) void tagged_adam_kernel(
    const int32_t *__restrict__ counts, AdamSide rowside, AdamSide colside, int row_blocks, int chunk_blocks,
    const float *__restrict__ scalars, const int64_t *__restrict__ step, int d4, float inv_batch, int head, float neg_factor,
    StepConsts kc, float b1, float b2, double ln_beta1, double ln_beta2, const float *__restrict__ prev, int prev_blocks,
    float *__restrict__ mine, int chain_i)
{
    constexpr int GPB = kBlock / LPR;
    constexpr int kRows = tagged_sweep_rows(NV);
    const int lg = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const uint32_t cur = (scalars[3] != 0.f ? 1u : 0u) ^ (uint32_t)(chain_i & 1);     // the copy this step reads (scalars[3]: as the chain began)
    const int64_t t = *step + chain_i + 1;                                            // global_step stands still during a chain
    const float lr_t = adam_lr_t(kc.lr, ln_beta1, ln_beta2, t);
    if ((int)blockIdx.x >= chunk_blocks) {
        // ---- sweep: a lane group per table row (R's rows first, then C's), kRows rows in flight; the batch's rows belong
        // to the chunk groups
        const int sb0 = blockIdx.x - chunk_blocks, nsb = gridDim.x - chunk_blocks;
        const int Vr = rowside.own_twin, Vc = colside.own_twin;
        const int total = Vr + Vc, stride = nsb * GPB;
        for (int v0 = sb0 * GPB + grp; v0 < total; v0 += kRows * stride) {
            f4 Wv[kRows][NV], M[kRows][NV], Vv[kRows][NV];
            float bval[kRows], Mb[kRows], Vb[kRows];
            uint32_t mk[kRows];
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const int v = v0 + r * stride;
                const bool live = v < total, is_row = v < Vr;
                const int id = live ? (is_row ? v : v - Vr) : 0;
                const AdamSide &sd = is_row ? rowside : colside;
                mk[r] = live ? (sd.mark[id >> 5] >> (id & 31)) & 1u : 1u;
                const int at = id + (cur ? sd.own_twin : 0);
                load_row<LPR, NV>(Wv[r], sd.own, at, d4, lg);
                load_row<LPR, NV>(M[r], sd.S1, id, d4, lg);
                load_row<LPR, NV>(Vv[r], sd.S2, id, d4, lg);
                bval[r] = Mb[r] = Vb[r] = 0.f;
                if (lg == 0) { bval[r] = sd.own_bias[at]; Mb[r] = sd.S1b[id]; Vb[r] = sd.S2b[id]; }
            }
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const int v = v0 + r * stride;
                if (v >= total || mk[r]) continue;
                const bool is_row = v < Vr;
                const int id = is_row ? v : v - Vr;
                const AdamSide &sd = is_row ? rowside : colside;
                const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < NV; ++kk) adam_vec(Wv[r][kk], M[r][kk], Vv[r][kk], zero, lr_t, b1, b2, kc.eps);
                const int to = id + (cur ? 0 : sd.own_twin);
                store_row<LPR, NV>(sd.S1, (size_t)id, d4, lg, M[r]);
                store_row<LPR, NV>(sd.S2, (size_t)id, d4, lg, Vv[r]);
                store_row<LPR, NV>(sd.own, (size_t)to, d4, lg, Wv[r]);
                if (lg == 0) {
                    adam_elem(bval[r], Mb[r], Vb[r], 0.f, lr_t, b1, b2, kc.eps);
                    sd.S1b[id] = Mb[r];
                    sd.S2b[id] = Vb[r];
                    sd.own_bias[to] = bval[r];
                }
            }
        }
        return;
    }
    constexpr int U = PassUnroll<NV>::value;
    constexpr int kRecStride = 4 + 3 * kChunkMax + 4;
    __shared__ __attribute__((aligned(16))) uint32_t fld_raw[GPB * kRecStride];
    uint32_t *rec = fld_raw + (threadIdx.x / LPR) * kRecStride;
    const bool is_row = (int)blockIdx.x < row_blocks;
    const AdamSide &sd = is_row ? rowside : colside;
    const int bid = is_row ? blockIdx.x : blockIdx.x - row_blocks;
    const int nblk = is_row ? row_blocks : chunk_blocks - row_blocks;
    // the group's first record is requested before anything else (as in tagged_step_kernel)
    constexpr int kPre = (1 + 3 * kChunkMax / 4 + LPR - 1) / LPR;
    u32x4 pre[kPre];       // (a native vector type: an array of the struct uint4 went through scratch at kPre = 4)
    {
        const int jf = bid * GPB + grp;
        const uint4 *rp = reinterpret_cast<const uint4 *>(sd.crec) + (size_t)(jf < sd.cap_chunks ? jf : sd.cap_chunks - 1) * rec_stride_q(sd.capP);
        const int rq0 = 1 + 3 * sd.capP / 4;
#pragma unroll
        for (int x = 0; x < kPre; ++x) {
            const int f = lg + x * LPR;
            pre[x] = reinterpret_cast<const u32x4 *>(rp)[rec_gq(f < rq0 ? f : 0)];
        }
    }
    const int n_chunks = sd.n_host >= 0 ? sd.n_host : counts[sd.count_index];
    float part[kPartials] = {0.f, 0.f, 0.f, 0.f};
    const int capP = sd.capP;
    const int rq = 1 + 3 * capP / 4;
    const int sq = rec_stride_q(capP);
    // ---- the global bias of THIS step: derived by every workgroup from the record the step before left (tagged_step_kernel)
    __shared__ float s_g[3];
    if (prev) {
        float tot[kPartials];
        sum_blockpart(prev + kChainHead, prev_blocks, tot);
        if (threadIdx.x == 0) {
            const float g0 = prev[0];
            const float dg = tot[3] + 2.0f * kc.m * kc.l2 * g0;
            float gn = g0, Mg = prev[1], Vg = prev[2];
            adam_elem(gn, Mg, Vg, dg, adam_lr_t(kc.lr, ln_beta1, ln_beta2, t - 1), b1, b2, kc.eps);
            s_g[0] = gn; s_g[1] = Mg; s_g[2] = Vg;
        }
    } else if (threadIdx.x == 0) {
        s_g[0] = scalars[0]; s_g[1] = scalars[1]; s_g[2] = scalars[2];
    }
    __syncthreads();
    const float g = s_g[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) { mine[0] = g; mine[1] = s_g[1]; mine[2] = s_g[2]; }
    float *blockpart = mine + kChainHead;
    const uint32_t other_add = cur ? (uint32_t)sd.other_twin : 0u;

    for (int j0 = bid * GPB + grp; j0 < n_chunks; j0 += nblk * GPB) {
        {
            uint4 *lrec = reinterpret_cast<uint4 *>(rec);
            if (j0 == bid * GPB + grp) {
#pragma unroll
                for (int x = 0; x < kPre; ++x) {
                    const int f = lg + x * LPR;
                    if (f < rq) reinterpret_cast<u32x4 *>(lrec)[f] = pre[x];
                }
            } else {
                const uint4 *rp = reinterpret_cast<const uint4 *>(sd.crec) + (size_t)j0 * sq;
                for (int f = lg; f < rq; f += LPR) lrec[f] = rp[rec_gq(f)];
            }
        }
        uint4 hdr = *reinterpret_cast<const uint4 *>(rec);
        if (!(hdr.w >> 31)) continue;                           // the group that holds the id's first chunk does the whole id
        const int32_t u = (int32_t)hdr.x;
        const int chunks = 1 + (int)(hdr.w & 0x7fffffffu);
        const int32_t own_at = u + (cur ? sd.own_twin : 0);
        f4 r[NV], M[NV], Vv[NV], G[NV];
        load_row<LPR, NV>(r, sd.own, own_at, d4, lg);
        const float own_b = sd.own_bias[own_at];
        load_row<LPR, NV>(M, sd.S1, u, d4, lg);
        load_row<LPR, NV>(Vv, sd.S2, u, d4, lg);
        float Mb = sd.S1b[u], Vb = sd.S2b[u];
        const float bg = own_b + g;
#pragma unroll
        for (int k = 0; k < NV; ++k) G[k] = f4{0.f, 0.f, 0.f, 0.f};
        float Gb = 0.f;
        int pairs = 0;
        for (int ch = 0; ch < chunks; ++ch) {
            const int j = j0 + ch;
            if (ch > 0) {
                const uint4 *rp = reinterpret_cast<const uint4 *>(sd.crec) + (size_t)j * sq;
                uint4 *lrec = reinterpret_cast<uint4 *>(rec);
                for (int f = lg; f < rq; f += LPR) lrec[f] = rp[rec_gq(f)];
                hdr = *reinterpret_cast<const uint4 *>(rec);
            }
            const int n = (int)hdr.y;
            pairs += n;
            f4 acc[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
            float se_c = 0.f, cc_sum = 0.f, bsq = 0.f, ed = 0.f;
            for (int q0 = 0; q0 < n; q0 += U) {
                int32_t col[U];
                float w2[U], yq[U];
#pragma unroll
                for (int a4 = 0; a4 < U; a4 += 4) {
                    const int rp0 = rec_pair(q0 + a4);
                    const uint4 pc = *reinterpret_cast<const uint4 *>(&rec[rp0]);
                    const uint4 pw = *reinterpret_cast<const uint4 *>(&rec[rp0 + kRecPad]);
                    const uint4 py = *reinterpret_cast<const uint4 *>(&rec[rp0 + 2 * kRecPad]);
                    col[a4] = (int32_t)(pc.x + other_add); col[a4 + 1] = (int32_t)(pc.y + other_add);
                    col[a4 + 2] = (int32_t)(pc.z + other_add); col[a4 + 3] = (int32_t)(pc.w + other_add);
                    const float sc2 = 2.0f * inv_batch;
                    w2[a4] = sc2 * __uint_as_float(pw.x); w2[a4 + 1] = sc2 * __uint_as_float(pw.y);
                    w2[a4 + 2] = sc2 * __uint_as_float(pw.z); w2[a4 + 3] = sc2 * __uint_as_float(pw.w);
                    yq[a4] = __uint_as_float(py.x); yq[a4 + 1] = __uint_as_float(py.y);
                    yq[a4 + 2] = __uint_as_float(py.z); yq[a4 + 3] = __uint_as_float(py.w);
                }
                f4 c[U][NV];
                float bcv[U];
#pragma unroll
                for (int a = 0; a < U; ++a) {
                    load_row_fast<LPR, NV, FULL>(c[a], sd.other, col[a], d4, lg);
                    bcv[a] = 0.f;
                    if (lg == 0) bcv[a] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(sd.other_bias) + (uint32_t)col[a] * 4u);
                }
                float dp[U], cc[U];
#pragma unroll
                for (int a = 0; a < U; ++a) {
                    dp[a] = 0.f; cc[a] = 0.f;
#pragma unroll
                    for (int k = 0; k < NV; ++k) { dp[a] += dot4(r[k], c[a][k]); cc[a] += dot4(c[a][k], c[a][k]); }
                    dp[a] += bcv[a];
                }
#pragma unroll
                for (int a = 0; a < U; ++a) dp[a] = dpp_add<0xB1>(dp[a]);
#pragma unroll
                for (int a = 0; a < U; ++a) dp[a] = dpp_add<0x4E>(dp[a]);
#pragma unroll
                for (int a = 0; a < U; ++a) dp[a] = dpp_add<0x141>(dp[a]);
                if (LPR >= 16) {
#pragma unroll
                    for (int a = 0; a < U; ++a) dp[a] = dpp_add<0x140>(dp[a]);
                }
                if (LPR >= 32) {
#pragma unroll
                    for (int a = 0; a < U; ++a) dp[a] += __shfl_xor(dp[a], 16, 64);
                }
                if (LPR >= 64) {
#pragma unroll
                    for (int a = 0; a < U; ++a) dp[a] += __shfl_xor(dp[a], 32, 64);
                }
#pragma unroll
                for (int a = 0; a < U; ++a) {
                    const float valid = (q0 + a < n) ? 1.0f : 0.f;
                    float e;
                    if (head == GLOVE_HEAD_REGRESSION) {
                        const float diff = (dp[a] + bg) - yq[a];
                        e = w2[a] * diff;
                        ed += e * diff;
                    } else {
                        const float p = dp[a] + bg;
                        const float en = expf(-fabsf(p));
                        const float s = (p >= 0.f ? 1.0f : en) / (1.0f + en);
                        const float lse = log1pf(en);
                        const float wn = 2.0f * inv_batch * neg_factor * valid * yq[a];
                        e = 0.5f * (w2[a] * (s - 1.0f) + wn * s);
                        ed += w2[a] * (fmaxf(-p, 0.f) + lse) + wn * (fmaxf(p, 0.f) + lse);
                    }
#pragma unroll
                    for (int k = 0; k < NV; ++k) acc[k] += e * c[a][k];
                    se_c += e;
                    cc_sum += valid * cc[a];
                    bsq += valid * bcv[a] * bcv[a];
                }
            }
            if (is_row) {
                float rr = 0.f;
#pragma unroll
                for (int k = 0; k < NV; ++k) rr += dot4(r[k], r[k]);
                part[1] += cc_sum + (float)n * rr;
                if (lg == 0) {
                    part[0] += ed;
                    part[2] += bsq + (float)n * own_b * own_b;
                    part[3] += se_c;
                }
            }
            if (ch == 0) {
#pragma unroll
                for (int k = 0; k < NV; ++k) G[k] = acc[k];
                Gb = se_c;
            } else {
#pragma unroll
                for (int k = 0; k < NV; ++k) G[k] += acc[k];
                Gb += se_c;
            }
        }
        {
            float bval = own_b;
            const float cnt = (float)pairs;
            const float kcn = kc.kappa * cnt;
#pragma unroll
            for (int k = 0; k < NV; ++k) G[k] += kcn * r[k];
            Gb += kc.kappa_b * cnt * bval;
#pragma unroll
            for (int k = 0; k < NV; ++k) adam_vec(r[k], M[k], Vv[k], G[k], lr_t, b1, b2, kc.eps);
            store_row<LPR, NV>(sd.S1, (size_t)u, d4, lg, M);
            store_row<LPR, NV>(sd.S2, (size_t)u, d4, lg, Vv);
            const int32_t new_at = u + (cur ? 0 : sd.own_twin);
            store_row<LPR, NV>(sd.own, (size_t)new_at, d4, lg, r);
            if (lg == 0) {
                adam_elem(bval, Mb, Vb, Gb, lr_t, b1, b2, kc.eps);
                sd.S1b[u] = Mb;
                sd.S2b[u] = Vb;
                sd.own_bias[new_at] = bval;
            }
        }
    }
    if (is_row) {
        part[0] *= 0.5f / inv_batch;
        block_partials_store(part, blockpart);
    }
}

// The end of a chain of n one-launch Adam steps: loss, the global bias the next step begins with, which copy is current, global_step
__global__ __launch_bounds__(kBlock) void tagged_adam_flush_kernel(float *__restrict__ scalars, int64_t *__restrict__ step,
                                                                   const float *__restrict__ last, int nblocks, int n, StepConsts k,
                                                                   float b1, float b2, double ln_beta1, double ln_beta2,
                                                                   float *__restrict__ loss_out)
{
    float tot[kPartials];
    sum_blockpart(last + kChainHead, nblocks, tot);
    if (threadIdx.x == 0) {
        const float g = last[0];
        float loss, L, reg;
        loss_from_partials(tot, k, g, loss, L, reg);
        const float dg = tot[3] + 2.0f * k.m * k.l2 * g;
        float gn = g, Mg = last[1], Vg = last[2];
        adam_elem(gn, Mg, Vg, dg, adam_lr_t(k.lr, ln_beta1, ln_beta2, *step + n), b1, b2, k.eps);
        scalars[0] = gn; scalars[1] = Mg; scalars[2] = Vg;
        if (n & 1) scalars[3] = scalars[3] != 0.f ? 0.f : 1.0f;
        if (loss_out) { loss_out[0] = loss; loss_out[1] = L; loss_out[2] = reg; loss_out[3] = tot[3]; }
        *step += n;
    }
}

// Twinned tables whose second copies are current as a whole (the one-launch Adam step, scalars[3] != 0): every row home
template <int LPR, int NV>
__global__ __launch_bounds__(kBlock) void twin_home_kernel(float *__restrict__ W, float *__restrict__ bias, const float *__restrict__ scalars,
                                                           int V, int d4)
{
    if (scalars[3] == 0.f) return;
    constexpr int GPB = kBlock / LPR;
    const int lg = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    for (int u = blockIdx.x * GPB + grp; u < V; u += gridDim.x * GPB) {
        f4 v[NV];
        load_row<LPR, NV>(v, W, u + V, d4, lg);
        store_row<LPR, NV>(W, (size_t)u, d4, lg, v);
        if (lg == 0) bias[u] = bias[u + V];
    }
}

__global__ void twin_home_done_kernel(float *scalars) { if (threadIdx.x == 0 && blockIdx.x == 0) scalars[3] = 0.f; }

// Step-tagged twinned tables back to the plain form: rows whose current copy is the second one are copied home, tags cleared
template <int LPR, int NV>
__global__ __launch_bounds__(kBlock) void untag_kernel(float *__restrict__ W, float *__restrict__ bias, uint64_t *__restrict__ tag,
                                                       int V, int d4)
{
    constexpr int GPB = kBlock / LPR;
    const int lg = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    for (int u = blockIdx.x * GPB + grp; u < V; u += gridDim.x * GPB) {
        const uint64_t tg = tag[u];
        if (tg == 0) continue;
        if (tg & 1ull) {
            f4 v[NV];
            load_row<LPR, NV>(v, W, u + V, d4, lg);
            store_row<LPR, NV>(W, (size_t)u, d4, lg, v);
            if (lg == 0) bias[u] = bias[u + V];
        }
        if (lg == 0) tag[u] = 0;
    }
}

// Twinned row table back to its plain form: every row whose current copy is the second one is copied into the first,
// all versions become 0.  Everything but the fused twin step expects this form.
template <int LPR, int NV>
__global__ __launch_bounds__(kBlock) void canonicalize_kernel(float *__restrict__ R, float *__restrict__ br,
                                                              uint8_t *__restrict__ ver, int V_row, int d4)
{
    constexpr int GPB = kBlock / LPR;
    const int lg = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    for (int u = blockIdx.x * GPB + grp; u < V_row; u += gridDim.x * GPB) {
        if (!ver[u]) continue;
        f4 v[NV];
        load_row<LPR, NV>(v, R, u + V_row, d4, lg);
        store_row<LPR, NV>(R, (size_t)u, d4, lg, v);
        if (lg == 0) { br[u] = br[u + V_row]; ver[u] = 0; }
    }
}

// rows[i] = W[ids[i]] (d floats each), biases[i] = bias[ids[i]]: what the owner of a table shard sends to the ranks
// whose batches touch those rows
template <int LPR, int NV>
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(
    const float *__restrict__ W, const float *__restrict__ bias, const int32_t *__restrict__ ids, int n, int d4,
    float *__restrict__ rows, float *__restrict__ biases)
{
    constexpr int GPB = kBlock / LPR;
    const int lg = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    for (int i = blockIdx.x * GPB + grp; i < n; i += gridDim.x * GPB) {
        const int32_t id = ids[i];
        f4 v[NV];
        load_row<LPR, NV>(v, W, id, d4, lg);
        store_row<LPR, NV>(rows, (size_t)i, d4, lg, v);
        if (lg == 0) biases[i] = bias[id];
    }
}

// ------------------------------------------------------------------------------------------
// Dense sweeps over [R | C | br | bc] consuming (and zeroing) G_flat.
// ------------------------------------------------------------------------------------------
struct DenseSeg { float *W, *S1, *S2, *G; int64_t n; };
struct DenseSegs { DenseSeg s[4]; };

__global__ __launch_bounds__(kBlock) void dense_adagrad_kernel(
    DenseSegs segs, StepConsts k, float *__restrict__ scalars, float *__restrict__ tail,
    float *__restrict__ loss_out, int do_scalars)
{
    // segments 0,1 = [V,d] tables (float4 body); 2,3 = bias vectors, whose G pointers are only 16-B
    // aligned when V % 4 == 0 and which are tiny: swept scalar by the same launch
    const DenseSeg sg = segs.s[blockIdx.y];
    const int64_t n4 = blockIdx.y < 2 ? sg.n / 4 : 0;
    f4 *W4 = reinterpret_cast<f4 *>(sg.W), *A4 = reinterpret_cast<f4 *>(sg.S1), *G4 = reinterpret_cast<f4 *>(sg.G);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        const f4 gv = G4[i];
        if (gv.x == 0.f && gv.y == 0.f && gv.z == 0.f && gv.w == 0.f) continue;  // untouched: exact no-op
        f4 a = A4[i], wv = W4[i];
        adagrad_vec(wv, a, gv, k.lr, k.eps);
        A4[i] = a; W4[i] = wv; G4[i] = f4{0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < sg.n; i += (int64_t)gridDim.x * kBlock) {
        const float gv = sg.G[i];
        if (gv != 0.f) { adagrad_elem(sg.W[i], sg.S1[i], gv, k.lr, k.eps); sg.G[i] = 0.f; }
    }
    if (do_scalars && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        const float g = scalars[0];
        const float tot[kPartials] = {tail[1], tail[2], tail[3], tail[0]};
        float loss, L, reg;
        loss_from_partials(tot, k, g, loss, L, reg);
        const float dg = tail[0] + 2.0f * k.m * k.l2 * g;
        adagrad_elem(scalars[0], scalars[1], dg, k.lr, k.eps);
        if (loss_out) { loss_out[0] = loss; loss_out[1] = L; loss_out[2] = reg; loss_out[3] = tail[0]; }
        tail[0] = tail[1] = tail[2] = tail[3] = 0.f;
    }
}

__global__ __launch_bounds__(kBlock) void dense_adam_kernel(
    DenseSegs segs, StepConsts k, float b1, float b2, double ln_beta1, double ln_beta2, const int64_t *__restrict__ step,
    float *__restrict__ scalars, float *__restrict__ tail, float *__restrict__ loss_out, int do_scalars)
{
    const DenseSeg sg = segs.s[blockIdx.y];
    // t = global_step after rowpass advanced it; lr_t = lr sqrt(1-b2^t)/(1-b1^t)  (Keras legacy Adam)
    // 1 - beta^t = -expm1(t ln beta): the logs come from the host in fp64, so each wave pays two fp64
    // multiplies and two expm1f instead of two fp64 pow() (~290 fp64 instructions, a third of this kernel's time)
    const float lr_t = adam_lr_t(k.lr, ln_beta1, ln_beta2, *step);
    const int64_t n4 = blockIdx.y < 2 ? sg.n / 4 : 0;
    f4 *W4 = reinterpret_cast<f4 *>(sg.W), *M4 = reinterpret_cast<f4 *>(sg.S1), *V4 = reinterpret_cast<f4 *>(sg.S2),
       *G4 = reinterpret_cast<f4 *>(sg.G);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        const f4 gv = G4[i];
        f4 m = M4[i], v = V4[i], wv = W4[i];
        adam_vec(wv, m, v, gv, lr_t, b1, b2, k.eps);
        M4[i] = m; V4[i] = v; W4[i] = wv;
        if (gv.x != 0.f || gv.y != 0.f || gv.z != 0.f || gv.w != 0.f) G4[i] = f4{0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < sg.n; i += (int64_t)gridDim.x * kBlock) {
        adam_elem(sg.W[i], sg.S1[i], sg.S2[i], sg.G[i], lr_t, b1, b2, k.eps);
        sg.G[i] = 0.f;
    }
    if (do_scalars && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        const float g = scalars[0];
        const float tot[kPartials] = {tail[1], tail[2], tail[3], tail[0]};
        float loss, L, reg;
        loss_from_partials(tot, k, g, loss, L, reg);
        const float dg = tail[0] + 2.0f * k.m * k.l2 * g;
        adam_elem(scalars[0], scalars[1], scalars[2], dg, lr_t, b1, b2, k.eps);
        if (loss_out) { loss_out[0] = loss; loss_out[1] = L; loss_out[2] = reg; loss_out[3] = tail[0]; }
        tail[0] = tail[1] = tail[2] = tail[3] = 0.f;
    }
}


// ------------------------------------------------------------------------------------------
// The other Keras optimizers `tf.keras.optimizers.get(name)` resolves (reference src/models/train_utils.py:13-16), as apply
// epilogues on the same traversal as AdagradApply.  Semantics: include/glove_hip.h glove_hyper.optimizer; restated in
// oracle/glove_ref.py (_sgd, _rmsprop_dense_decay, _adamax).
// ------------------------------------------------------------------------------------------
template <int LPR, int NV, int OPT>
struct SparseOptApply {
    SideBufs rs, cs;
    float *S2_R, *S2_C, *S2_br, *S2_bc;
    int d4, lg;
    OptConsts o;
    struct State { f4 A[NV], B[NV]; float Ab, Bb; };
    __device__ void prefetch(bool is_row, int32_t id, State &st) const
    {
        const SideBufs &sb = is_row ? rs : cs;
        if (OPT == GLOVE_OPT_SGD && o.momentum == 0.f) return;
        load_row<LPR, NV>(st.A, sb.S1, id, d4, lg);
        st.Ab = sb.S1b[id];
        if (OptSlots<OPT>::two) {
            load_row<LPR, NV>(st.B, is_row ? S2_R : S2_C, id, d4, lg);
            st.Bb = (is_row ? S2_br : S2_bc)[id];
        }
    }
    __device__ static void one(float &w, float &a, float &b, float g, const OptConsts &o) { OptElem<OPT>::one(w, a, b, g, o); }
    __device__ void finish(bool is_row, int32_t id, int32_t wid, int q, f4 (&G)[NV], f4 (&Wv)[NV], float Gb, float bval, State &st) const
    {
        const SideBufs &sb = is_row ? rs : cs;
        const bool slots = !(OPT == GLOVE_OPT_SGD && o.momentum == 0.f);
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) {
            float w[4] = {Wv[kk].x, Wv[kk].y, Wv[kk].z, Wv[kk].w}, a[4] = {st.A[kk].x, st.A[kk].y, st.A[kk].z, st.A[kk].w};
            float b[4] = {st.B[kk].x, st.B[kk].y, st.B[kk].z, st.B[kk].w};
            const float g[4] = {G[kk].x, G[kk].y, G[kk].z, G[kk].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) one(w[i], a[i], b[i], g[i], o);
            Wv[kk] = f4{w[0], w[1], w[2], w[3]};
            st.A[kk] = f4{a[0], a[1], a[2], a[3]};
            st.B[kk] = f4{b[0], b[1], b[2], b[3]};
        }
        if (slots) store_row<LPR, NV>(sb.S1, (size_t)id, d4, lg, st.A);
        if (OptSlots<OPT>::two) store_row<LPR, NV>(is_row ? S2_R : S2_C, (size_t)id, d4, lg, st.B);
        store_row<LPR, NV>(sb.W, (size_t)wid, d4, lg, Wv);
        if (lg == 0) {
            one(bval, st.Ab, st.Bb, Gb, o);
            if (slots) sb.S1b[id] = st.Ab;
            if (OptSlots<OPT>::two) (is_row ? S2_br : S2_bc)[id] = st.Bb;
            sb.bias[wid] = bval;
        }
    }
};

template <int LPR, int NV, int OPT>
__global__ __launch_bounds__(kBlock) void apply_sparse_opt_kernel(
    IdWork wk, SideBufs rs, SideBufs cs, float *S2_R, float *S2_C, float *S2_br, float *S2_bc, int d4, StepConsts k, OptConsts o,
    double ln_beta1, const int64_t *__restrict__ step, float *__restrict__ scalars, const float *__restrict__ blockpart,
    int nblocks_rowpass, float *__restrict__ loss_out)
{
    // t = global_step after rowpass advanced it (Adamax: lr_t = lr / (1 - beta1^t))
    if (OPT == GLOVE_OPT_ADAMAX) o.lr_t = o.lr / -expm1f((float)((double)(*step) * ln_beta1));
    const bool scalar_duty = for_each_id<LPR, NV>(wk, rs, cs, d4, k,
                                                  SparseOptApply<LPR, NV, OPT>{rs, cs, S2_R, S2_C, S2_br, S2_bc, d4, (int)(threadIdx.x % LPR), o});
    if (scalar_duty) {
        float tot[kPartials];
        sum_blockpart(blockpart, nblocks_rowpass, tot);
        if (threadIdx.x == 0) {
            const float g = scalars[0];
            float loss, L, reg;
            loss_from_partials(tot, k, g, loss, L, reg);
            const float dg = tot[3] + 2.0f * k.m * k.l2 * g;
            float gn = g, a = scalars[1], b = scalars[2];
            SparseOptApply<LPR, NV, OPT>::one(gn, a, b, dg, o);
            scalars[0] = gn; scalars[1] = a; scalars[2] = b;
            if (loss_out) { loss_out[0] = loss; loss_out[1] = L; loss_out[2] = reg; loss_out[3] = tot[3]; }
        }
    }
}

// Keras-legacy RMSprop over [R | C | br | bc], consuming (and zeroing) G_flat: every rms entry decays, entries with a gradient move
__global__ __launch_bounds__(kBlock) void dense_rmsprop_kernel(
    DenseSegs segs, StepConsts k, float rho, float *__restrict__ scalars, float *__restrict__ tail,
    float *__restrict__ loss_out, int do_scalars)
{
#pragma clang fp contract(off)
    const DenseSeg sg = segs.s[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < sg.n; i += (int64_t)gridDim.x * kBlock) {
        const float gv = sg.G[i];
        const float a = rho * sg.S1[i] + (1.0f - rho) * gv * gv;
        sg.S1[i] = a;
        if (gv != 0.f) {
            sg.W[i] -= k.lr * gv / (sqrtf(a) + k.eps);
            sg.G[i] = 0.f;
        }
    }
    if (do_scalars && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        const float g = scalars[0];
        const float tot[kPartials] = {tail[1], tail[2], tail[3], tail[0]};
        float loss, L, reg;
        loss_from_partials(tot, k, g, loss, L, reg);
        const float dg = tail[0] + 2.0f * k.m * k.l2 * g;
        const float a = rho * scalars[1] + (1.0f - rho) * dg * dg;
        scalars[1] = a;
        scalars[0] = g - k.lr * dg / (sqrtf(a) + k.eps);
        if (loss_out) { loss_out[0] = loss; loss_out[1] = L; loss_out[2] = reg; loss_out[3] = tail[0]; }
        tail[0] = tail[1] = tail[2] = tail[3] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// Keras-legacy Adam in ONE launch behind the passes (single GPU, batches that touch a minority of the rows:
// the reference's default, 1,024 pairs against a 10^4-row vocabulary).  The passes left mark[id] = 1 for
// every id of the batch (in the bias segments of G_flat, which is otherwise unused on this path).
//   * the leading workgroups are the sparse apply: per distinct id, sum of the chunk partials -> m, v, W
//     (same traversal and summation order as the Adagrad apply and the dense-gradient kernel);
//   * the remaining workgroups sweep ALL rows, one lane group per row: a marked row belongs to the apply part
//     (its mark is cleared, nothing else), an unmarked row takes the G = 0 update m *= b1, v *= b2,
//     W -= lr_t m / (sqrt(v) + eps) — Keras' sparse Adam decays every row of the table, a11.
// The two parts touch disjoint rows, so they run side by side; only a row's own sweep group reads or clears
// its mark.  Results are bit-identical to dense_grad + dense_adam; G_flat is all zero again afterwards.
// ------------------------------------------------------------------------------------------
template <int LPR, int NV>
struct AdamApply {
    SideBufs rs, cs;
    float *S2_R, *S2_C, *S2_br, *S2_bc;
    int d4, lg;
    float lr_t, b1, b2, eps;
    struct State { f4 M[NV], V[NV]; float Mb, Vb; };
    __device__ void prefetch(bool is_row, int32_t id, State &st) const
    {
        const SideBufs &sb = is_row ? rs : cs;
        load_row<LPR, NV>(st.M, sb.S1, id, d4, lg);
        load_row<LPR, NV>(st.V, is_row ? S2_R : S2_C, id, d4, lg);
        st.Mb = sb.S1b[id];
        st.Vb = (is_row ? S2_br : S2_bc)[id];
    }
    __device__ void finish(bool is_row, int32_t id, int32_t wid, int q, f4 (&G)[NV], f4 (&Wv)[NV], float Gb, float bval, State &st) const
    {
        const SideBufs &sb = is_row ? rs : cs;
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) adam_vec(Wv[kk], st.M[kk], st.V[kk], G[kk], lr_t, b1, b2, eps);
        store_row<LPR, NV>(sb.S1, (size_t)id, d4, lg, st.M);
        store_row<LPR, NV>(is_row ? S2_R : S2_C, (size_t)id, d4, lg, st.V);
        store_row<LPR, NV>(sb.W, (size_t)id, d4, lg, Wv);
        if (lg == 0) {
            adam_elem(bval, st.Mb, st.Vb, Gb, lr_t, b1, b2, eps);
            sb.S1b[id] = st.Mb;
            (is_row ? S2_br : S2_bc)[id] = st.Vb;
            sb.bias[id] = bval;
        }
    }
};

template <int LPR, int NV>
struct NadamApply {
    SideBufs rs, cs;
    float *S2_R, *S2_C, *S2_br, *S2_bc;
    int d4, lg;
    NadamConsts k;
    struct State { f4 M[NV], V[NV]; float Mb, Vb; };
    __device__ void prefetch(bool is_row, int32_t id, State &st) const
    {
        const SideBufs &sb = is_row ? rs : cs;
        load_row<LPR, NV>(st.M, sb.S1, id, d4, lg);
        load_row<LPR, NV>(st.V, is_row ? S2_R : S2_C, id, d4, lg);
        st.Mb = sb.S1b[id];
        st.Vb = (is_row ? S2_br : S2_bc)[id];
    }
    __device__ void finish(bool is_row, int32_t id, int32_t wid, int q, f4 (&G)[NV], f4 (&Wv)[NV], float Gb, float bval, State &st) const
    {
        const SideBufs &sb = is_row ? rs : cs;
#pragma unroll
        for (int kk = 0; kk < NV; ++kk) {
            float w[4] = {Wv[kk].x, Wv[kk].y, Wv[kk].z, Wv[kk].w}, m[4] = {st.M[kk].x, st.M[kk].y, st.M[kk].z, st.M[kk].w};
            float v[4] = {st.V[kk].x, st.V[kk].y, st.V[kk].z, st.V[kk].w};
            const float g[4] = {G[kk].x, G[kk].y, G[kk].z, G[kk].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) nadam_elem(w[i], m[i], v[i], g[i], k);
            Wv[kk] = f4{w[0], w[1], w[2], w[3]};
            st.M[kk] = f4{m[0], m[1], m[2], m[3]};
            st.V[kk] = f4{v[0], v[1], v[2], v[3]};
        }
        store_row<LPR, NV>(sb.S1, (size_t)id, d4, lg, st.M);
        store_row<LPR, NV>(is_row ? S2_R : S2_C, (size_t)id, d4, lg, st.V);
        store_row<LPR, NV>(sb.W, (size_t)id, d4, lg, Wv);
        if (lg == 0) {
            nadam_elem(bval, st.Mb, st.Vb, Gb, k);
            sb.S1b[id] = st.Mb;
            (is_row ? S2_br : S2_bc)[id] = st.Vb;
            sb.bias[id] = bval;
        }
    }
};

template <int LPR, int NV>
__global__ __launch_bounds__(kBlock) void nadam_fused_kernel(
    IdWork wk, SideBufs rs, SideBufs cs, float *S2_R, float *S2_C, float *S2_br, float *S2_bc, int d4, StepConsts k,
    float b1, float b2, double ln_beta2, const int64_t *__restrict__ step,
    float *__restrict__ scalars, const float *__restrict__ blockpart, int nblocks_rowpass,
    float *__restrict__ mark_rows, float *__restrict__ mark_cols, int V_row, int V, int apply_blocks,
    float *__restrict__ loss_out)
{
#pragma clang fp contract(off)
    constexpr int GPB = kBlock / LPR;
    const int lg = threadIdx.x % LPR;
    const NadamConsts nk = nadam_consts(k.lr, k.eps, b1, b2, ln_beta2, *step, scalars);
    if ((int)blockIdx.x >= apply_blocks) {
        // ---- sweep (as adam_fused_kernel's): an unmarked row's m and v decay, the row itself stays; a marked row belongs to the apply part
        const int grp = threadIdx.x / LPR;
        const int sb0 = blockIdx.x - apply_blocks, nsb = gridDim.x - apply_blocks;
        const int total = V_row + V, stride = nsb * GPB;
        for (int v0 = sb0 * GPB + grp; v0 < total; v0 += kSweepRows * stride) {
            f4 M[kSweepRows][NV], Vv[kSweepRows][NV];
            float mk[kSweepRows], Mb[kSweepRows], Vb[kSweepRows];
#pragma unroll
            for (int r = 0; r < kSweepRows; ++r) {
                const int v = v0 + r * stride;
                const bool live = v < total;
                const bool is_row = v < V_row;
                const int id = live ? (is_row ? v : v - V_row) : 0;
                const SideBufs &sb = is_row ? rs : cs;
                mk[r] = live ? (is_row ? mark_rows : mark_cols)[id] : 1.0f;
                load_row<LPR, NV>(M[r], sb.S1, id, d4, lg);
                load_row<LPR, NV>(Vv[r], is_row ? S2_R : S2_C, id, d4, lg);
                Mb[r] = Vb[r] = 0.f;
                if (lg == 0) { Mb[r] = sb.S1b[id]; Vb[r] = (is_row ? S2_br : S2_bc)[id]; }
            }
#pragma unroll
            for (int r = 0; r < kSweepRows; ++r) {
                const int v = v0 + r * stride;
                if (v >= total) continue;
                const bool is_row = v < V_row;
                const int id = is_row ? v : v - V_row;
                const SideBufs &sb = is_row ? rs : cs;
                float *S2 = is_row ? S2_R : S2_C, *S2b = is_row ? S2_br : S2_bc;
                if (mk[r] != 0.f) {
                    if (lg == 0) (is_row ? mark_rows : mark_cols)[id] = 0.f;
                    continue;
                }
#pragma unroll
                for (int kk = 0; kk < NV; ++kk) { M[r][kk] = b1 * M[r][kk]; Vv[r][kk] = b2 * Vv[r][kk]; }
                store_row<LPR, NV>(sb.S1, (size_t)id, d4, lg, M[r]);
                store_row<LPR, NV>(S2, (size_t)id, d4, lg, Vv[r]);
                if (lg == 0) { sb.S1b[id] = b1 * Mb[r]; S2b[id] = b2 * Vb[r]; }
            }
        }
        return;
    }
    const bool scalar_duty = for_each_id<LPR, NV>(wk, rs, cs, d4, k, NadamApply<LPR, NV>{rs, cs, S2_R, S2_C, S2_br, S2_bc, d4, lg, nk},
                                                  apply_blocks);
    if (scalar_duty) {
        float tot[kPartials];
        sum_blockpart(blockpart, nblocks_rowpass, tot);
        if (threadIdx.x == 0) {
            const float g = scalars[0];
            float loss, L, reg;
            loss_from_partials(tot, k, g, loss, L, reg);
            const float dg = tot[3] + 2.0f * k.m * k.l2 * g;
            nadam_elem(scalars[0], scalars[1], scalars[2], dg, nk);
            scalars[4 + (int)(*step & 1)] = nk.sched_new;
            if (loss_out) { loss_out[0] = loss; loss_out[1] = L; loss_out[2] = reg; loss_out[3] = tot[3]; }
        }
    }
}

template <int LPR, int NV>
__global__ __launch_bounds__(kBlock) void adam_fused_kernel(
    IdWork wk, SideBufs rs, SideBufs cs, float *S2_R, float *S2_C, float *S2_br, float *S2_bc, int d4, StepConsts k,
    float b1, float b2, double ln_beta1, double ln_beta2, const int64_t *__restrict__ step,
    float *__restrict__ scalars, const float *__restrict__ blockpart, int nblocks_rowpass,
    float *__restrict__ mark_rows, float *__restrict__ mark_cols, int V_row, int V, int apply_blocks,
    float *__restrict__ loss_out)
{
    constexpr int GPB = kBlock / LPR;
    const int lg = threadIdx.x % LPR;
    const float lr_t = adam_lr_t(k.lr, ln_beta1, ln_beta2, *step);
    if ((int)blockIdx.x >= apply_blocks) {
        // ---- sweep: a lane group per table row (R's rows first, then C's), kSweepRows rows in flight per group so
        // that the whole sweep is resident in one round next to the register-heavier apply part
        const int grp = threadIdx.x / LPR;
        const int sb0 = blockIdx.x - apply_blocks, nsb = gridDim.x - apply_blocks;
        const int total = V_row + V, stride = nsb * GPB;
        for (int v0 = sb0 * GPB + grp; v0 < total; v0 += kSweepRows * stride) {
            f4 Wv[kSweepRows][NV], M[kSweepRows][NV], Vv[kSweepRows][NV];
            float mk[kSweepRows], bval[kSweepRows], Mb[kSweepRows], Vb[kSweepRows];
            // everything is requested at once; what a marked row loaded is simply dropped
#pragma unroll
            for (int r = 0; r < kSweepRows; ++r) {
                const int v = v0 + r * stride;
                const bool live = v < total;
                const bool is_row = v < V_row;
                const int id = live ? (is_row ? v : v - V_row) : 0;
                const SideBufs &sb = is_row ? rs : cs;
                mk[r] = live ? (is_row ? mark_rows : mark_cols)[id] : 1.0f;
                load_row<LPR, NV>(Wv[r], sb.W, id, d4, lg);
                load_row<LPR, NV>(M[r], sb.S1, id, d4, lg);
                load_row<LPR, NV>(Vv[r], is_row ? S2_R : S2_C, id, d4, lg);
                bval[r] = Mb[r] = Vb[r] = 0.f;
                if (lg == 0) { bval[r] = sb.bias[id]; Mb[r] = sb.S1b[id]; Vb[r] = (is_row ? S2_br : S2_bc)[id]; }
            }
#pragma unroll
            for (int r = 0; r < kSweepRows; ++r) {
                const int v = v0 + r * stride;
                if (v >= total) continue;
                const bool is_row = v < V_row;
                const int id = is_row ? v : v - V_row;
                const SideBufs &sb = is_row ? rs : cs;
                float *S2 = is_row ? S2_R : S2_C, *S2b = is_row ? S2_br : S2_bc;
                if (mk[r] != 0.f) {                              // the apply part owns this row
                    if (lg == 0) (is_row ? mark_rows : mark_cols)[id] = 0.f;
                    continue;
                }
                const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < NV; ++kk) adam_vec(Wv[r][kk], M[r][kk], Vv[r][kk], zero, lr_t, b1, b2, k.eps);
                store_row<LPR, NV>(sb.S1, (size_t)id, d4, lg, M[r]);
                store_row<LPR, NV>(S2, (size_t)id, d4, lg, Vv[r]);
                store_row<LPR, NV>(sb.W, (size_t)id, d4, lg, Wv[r]);
                if (lg == 0) {
                    adam_elem(bval[r], Mb[r], Vb[r], 0.f, lr_t, b1, b2, k.eps);
                    sb.S1b[id] = Mb[r];
                    S2b[id] = Vb[r];
                    sb.bias[id] = bval[r];
                }
            }
        }
        return;
    }
    // ---- apply: the ids of the batch (for_each_id sees a grid of apply_blocks workgroups)
    const bool scalar_duty = for_each_id<LPR, NV>(wk, rs, cs, d4, k,
                                                  AdamApply<LPR, NV>{rs, cs, S2_R, S2_C, S2_br, S2_bc, d4, lg, lr_t, b1, b2, k.eps},
                                                  apply_blocks);
    if (scalar_duty) {
        float tot[kPartials];
        sum_blockpart(blockpart, nblocks_rowpass, tot);
        if (threadIdx.x == 0) {
            const float g = scalars[0];
            float loss, L, reg;
            loss_from_partials(tot, k, g, loss, L, reg);
            const float dg = tot[3] + 2.0f * k.m * k.l2 * g;
            adam_elem(scalars[0], scalars[1], scalars[2], dg, lr_t, b1, b2, k.eps);
            if (loss_out) { loss_out[0] = loss; loss_out[1] = L; loss_out[2] = reg; loss_out[3] = tot[3]; }
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static inline RowShape pass_shape(int d4) { return pick_pass_shape(d4); }

static int check_common(const glove_plan *p, const glove_tables *t, const glove_hyper *h, const void *ws)
{
    if (!p || !t || !h || !ws) return GLOVE_E_BADARG;
    if (p->B < 0 || p->cap_chunks < 0 || t->V <= 0 || t->d <= 0 || (t->d % 4) != 0) return GLOVE_E_BADARG;
    if (!p->counts || !t->R || !t->C || !t->br || !t->bc || !t->scalars || !t->step) return GLOVE_E_BADARG;
    if (p->B > 0 && (!p->r_chunk_id || !p->r_chunk_start || !p->r_uniq_slot ||
                     !p->heavy || p->heavy_chunks < 1 || !p->r_uniq_rec || !p->c_uniq_rec || !p->c_chunk_id || !p->c_chunk_start || !p->c_uniq_slot))
        return GLOVE_E_BADARG;
    // the pair fields: in arrays of the plan's own, or (a plan of a dealt epoch, glove_plan_build_sorted) in its chunk records only
    const bool own_pairs = p->r_partner && p->r_w && p->r_y && p->c_partner && p->c_w && p->c_y;
    if (p->B > 0 && !own_pairs && !(p->r_crec && p->c_crec)) return GLOVE_E_BADARG;
    const RowShape shape = pick_row_shape(t->d / 4);
    if (shape.lpr == 0 || pass_shape(t->d / 4).lpr == 0 || p->chunk_cap <= 0) return GLOVE_E_BADARG;
    if ((uint64_t)t->V * (uint64_t)t->d * 4u >= (1ull << 32) || t->V_row < 0 || t->V_row > t->V) return GLOVE_E_BADARG;   // 32-bit row offsets
    if (t->R_ver && 2ull * (uint64_t)(t->V_row > 0 ? t->V_row : t->V) * (uint64_t)t->d * 4u >= (1ull << 32)) return GLOVE_E_BADARG;
    if ((t->R_tag != nullptr) != (t->C_tag != nullptr) || (t->R_tag && t->R_ver)) return GLOVE_E_BADARG;   // tags: both tables, not beside R_ver
    if (t->R_tag && 2ull * (uint64_t)t->V * (uint64_t)t->d * 4u >= (1ull << 32)) return GLOVE_E_BADARG;
    if (t->d_model < 0 || t->d_model > t->d) return GLOVE_E_BADARG;
    if (p->chunk_cap > kChunkMax) return GLOVE_E_BADARG;
    if (h->head != GLOVE_HEAD_REGRESSION && h->head != GLOVE_HEAD_LOGISTIC) return GLOVE_E_BADARG;
    return 0;
}

static inline int sides_of(const glove_hyper *h) { return (h->sides & 3) ? (h->sides & 3) : 3; }
static inline int v_row(const glove_tables *t) { return t->V_row > 0 ? t->V_row : t->V; }

struct GradLayout { int64_t G_R, G_br, G_C, G_bc, tail, total; };
static GradLayout grad_layout(int32_t Vr, int32_t V, int32_t d)
{
    GradLayout L;
    auto up4 = [](int64_t x) { return (x + 3) / 4 * 4; };
    L.G_R = 0;
    L.G_br = (int64_t)Vr * d;
    L.G_C = up4(L.G_br + Vr);
    L.G_bc = L.G_C + (int64_t)V * d;
    L.tail = up4(L.G_bc + V);
    L.total = L.tail + 8;
    return L;
}

static IdWork id_work(const glove_plan *p)
{
    IdWork w;
    w.counts = p->counts;
    w.nu_r_host = p->host_counts[1];
    w.nu_c_host = p->host_counts[3];
    w.n_heavy_host = p->host_counts[4];
    w.heavy = p->heavy;
    w.heavy_blocks = p->host_counts[4] >= 0 ? p->host_counts[4] : p->cap_heavy;
    w.heavy_chunks = p->heavy_chunks;
    w.sides = 3;
    w.pre_r = w.pre_c = kFuseNone;
    w.per = 1;
    w.gpb = 0;
    w.work = nullptr;
    w.flips = 0;
    return w;
}

static StepConsts make_consts(const glove_tables *t, const glove_hyper *h)
{
    StepConsts k;
    const float dm = (float)(t->d_model > 0 ? t->d_model : t->d);      // the reference's embedding size
    k.kappa = 2.0f * h->reg_mult * h->l2_reg / dm * h->inv_batch;
    k.kappa_b = 2.0f * h->reg_mult * h->l2_reg * h->inv_batch;
    k.lr = h->learning_rate;
    k.eps = h->epsilon;
    k.l2 = h->l2_reg;
    k.m = h->reg_mult;
    k.inv_batch = h->inv_batch;
    k.inv_d = 1.0f / dm;
    return k;
}

static inline int rowpass_blocks(const glove_plan *p, int lpr) { return blocks_for(p->cap_chunks, kBlock / lpr); }

// FUSE passes: consecutive chunks per lane group.  Up to 6 (12 until round 5, see below): a heavy id then leaves one partial row per 12 chunks instead
// of one per chunk, and an id of a few chunks is usually applied by the pass itself (an id whose chunks straddle two
// groups goes through partial rows and the apply launch) — but the ids of the Zipf head fill consecutive FULL chunks, so
// a head group's per x chunk_cap pairs are one serial chain of partner-row trips, the launch's critical path once per is
// large.  One-process A/B of the whole step on resident plans, B = 1 M, chunk_cap 32 (tools/ab_kernels.py, builds with
// a forced per): V = 400 k, d = 300: per 4 / 8 / 12 / 16 / 20 / 23 / 27 / 32 / 48 = 620 / 613 / 609 / 640 / 699 / 772 /
// 868 / 979 / 1281 us; V = 2 M, d = 128: 516 / 509 / 520 / 536 / 541 / 572 / - / 565 / 765 (builds differ by +-1.5 % whatever
// they hold: 12 and the former rule's 12 / 18 are not told apart).  Fewer when the side has too few
// chunks to fill the chip with such groups (V = 400 k at B = 131,072: per 1 or 2 156 us, per 4 165 us; V = 50 k: 103 /
// 103 / 106), more only when the plan has more chunks than kMaxPassBlocks workgroups of such groups cover (the loss
// partials are kept per workgroup).
// Solid workgroups (sidepass_kernel, Slots) pay where the head of the batch leaves hundreds of partial rows: from 131,072 chunks a
// side (V = 400 k, d = 300, one process, same plans: B = 1 M 565.3 -> 557.7 us per step; at B = 131,072 — 50 k chunks, the id of
// rank 1 in 50 runs — the head workgroups' epilogue is on the passes' critical path: 139.3 -> 142.4).  Pass and apply launch ask here.
static bool solid_groups(const glove_plan *p)
{
    const int64_t r = most_chunks(p, true), c = most_chunks(p, false);
    return (r > c ? r : c) >= 131072;
}

static int fuse_per(const glove_plan *p, int lpr)
{
    const int64_t nr = most_chunks(p, true), nc = most_chunks(p, false);
    const int64_t side = nr > nc ? nr : nc;
    int per = (int)((side + 16383) / 16384);              // <= 2,048 workgroups up to 196 k chunks a side
    // (round 5, the passes compiled for one head with staged version lookups — shorter per-chunk chains, more waves: the same
    // A/B, one process, us per step: V = 400 k, d = 300: per 3 / 4 / 5 / 6 / 8 / 10 / 12 / 15 = 558 / 546 / 543 / 540 / 545* / 551* / 550 / 595*;
    // V = 2 M, d = 128: 344.6 / 344.5 / 344.4 / 343.1 / 344* / 348* / 348.5 / 431* (* scaled from another box's run against 12): 6)
    per = per < 1 ? 1 : per > 6 ? 6 : per;
    const int64_t groups = (int64_t)kMaxPassBlocks * (kBlock / lpr);
    // (`side` is the chunk count itself when the host knows it — also for a staging plan whose counts were read back: its
    // capacity is the worst case, V = 2 M at B = 1 M ran 16 chunks per group instead of 12 for nothing, 530 against 511 us)
    const int need = (int)((side + groups - 1) / groups);
    return need > per ? need : per;
}
static inline int fusepass_blocks(const glove_plan *p, int lpr, int per, bool row)
{
    const int64_t per_block = (int64_t)per * (kBlock / lpr);
    const int64_t b = (most_chunks(p, row) + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : b > kMaxPassBlocks ? kMaxPassBlocks : b);
}

static SideBufs side_bufs(const glove_plan *p, const StepWs &w, const glove_tables *t, bool row)
{
    SideBufs s;
    s.uniq_rec = row ? p->r_uniq_rec : p->c_uniq_rec;
    s.gp = row ? w.gp_r : w.gp_c;
    s.gb = row ? w.gb_r : w.gb_c;
    s.W = row ? t->R : t->C;
    s.S1 = row ? t->s1_R : t->s1_C;
    s.bias = row ? t->br : t->bc;
    s.S1b = row ? t->s1_br : t->s1_bc;
    s.ver = nullptr;            // set by the twin step alone: every other caller sees a canonical table
    s.twin = 0;
    return s;
}

}  // namespace glove

using namespace glove;

extern "C" {

int glove_abi_version(void) { return GLOVE_ABI_VERSION; }

// Every entry point but the twin form of the Adagrad step reads and writes rows 0 .. V_row-1 of a twinned row table: it first
// brings the table back to that form (one small launch over V_row version bytes plus the rows whose second copy was
// current; nothing without a twin).  A caller that mixes step forms, or steps and anything else, on one twinned table
// therefore never reads a stale copy.
static int plain_table(const glove_tables *t, void *stream) { return t && (t->R_ver || t->R_tag) ? glove_canonicalize_f32(t, stream) : 0; }
#ifdef GLOVE_STAMPS
int glove_debug_set_stamps(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)); }
#endif

size_t glove_step_workspace_bytes(int64_t B, int32_t cap_chunks, int32_t d)
{
    if (B < 0 || cap_chunks < 0 || d <= 0) return 0;
    return carve_step_ws(nullptr, B, cap_chunks, d).bytes;
}


size_t glove_dense_grad_layout(int32_t V_row, int32_t V, int32_t d, int64_t *offs)
{
    const GradLayout L = grad_layout(V_row > 0 ? V_row : V, V, d);
    if (offs) { offs[0] = L.G_R; offs[1] = L.G_br; offs[2] = L.G_C; offs[3] = L.G_bc; offs[4] = L.tail; }
    return (size_t)L.total;
}

static PassSide pass_side(const glove_plan *p, const glove_tables *t, const StepWs &w, bool row, bool want_e = false,
                          int fuse = kFuseNone, bool twin = false)
{
    PassSide sd;
    sd.partner = row ? p->r_partner : p->c_partner;
    sd.w = row ? p->r_w : p->c_w;
    sd.y = row ? p->r_y : p->c_y;
    sd.chunk_id = row ? p->r_chunk_id : p->c_chunk_id;
    sd.chunk_start = row ? p->r_chunk_start : p->c_chunk_start;
    sd.own = row ? t->R : t->C;
    sd.other = row ? t->C : t->R;
    sd.own_bias = row ? t->br : t->bc;
    sd.other_bias = row ? t->bc : t->br;
    sd.gp = row ? w.gp_r : w.gp_c;
    sd.gb = row ? w.gb_r : w.gb_c;
    sd.e_out = row && want_e ? w.e : nullptr;
    sd.mark = nullptr;
    sd.n_host = p->host_counts[row ? 0 : 2];
    sd.count_index = row ? 0 : 2;
    sd.crec = row ? p->r_crec : p->c_crec;
    sd.chunk_hw = row ? p->r_chunk_hw : p->c_chunk_hw;
    sd.capP = rec_cap(p->chunk_cap);
    sd.fuse = fuse;
    sd.own_out = row ? t->R : t->C;
    sd.own_bias_out = row ? t->br : t->bc;
    sd.S1 = row ? t->s1_R : t->s1_C;
    sd.S1b = row ? t->s1_br : t->s1_bc;
    // the row table is the one that may be twinned: own of the row side, partner of the col side
    sd.own_ver = twin && row ? t->R_ver : nullptr;
    sd.other_ver = twin && !row ? t->R_ver : nullptr;
    sd.own_twin = sd.other_twin = v_row(t);
    return sd;
}

// which: 1 = row side, 2 = col side, 3 = both in one launch
static int launch_passes(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws, size_t ws_bytes,
                         void *stream, int which, float *mark_rows = nullptr, float *mark_cols = nullptr,
                         bool want_e = false, int fuse_r = kFuseNone, int fuse_c = kFuseNone, bool twin = false,
                         float *packed = nullptr, bool list_tail = false)
{
    if (int rc = check_common(p, t, h, ws)) return rc;
    const bool fuse = fuse_r != kFuseNone || fuse_c != kFuseNone || twin;
    if ((twin || fuse_r == kFuseTwin) && !t->R_ver) return GLOVE_E_BADARG;
    if (fuse_c == kFuseTwin) return GLOVE_E_BADARG;          // only the row table has a twin
    const bool applies = twin || (fuse_r != kFuseNone && fuse_r != kFusePack) || (fuse_c != kFuseNone && fuse_c != kFusePack);
    if (applies && (!t->s1_R || !t->s1_C || !t->s1_br || !t->s1_bc)) return GLOVE_E_BADARG;
    // in place is only legal for a side whose table no concurrent chunk gathers from: one side per launch
    if ((fuse_r == kFuseInPlace && (which & 2)) || (fuse_c == kFuseInPlace && (which & 1))) return GLOVE_E_BADARG;
    const StepWs w = carve_step_ws(ws, p->B, p->cap_chunks, t->d);
    if (w.bytes > ws_bytes) return GLOVE_E_WORKSPACE;
    const int d4 = t->d / 4;
    const RowShape shape = pass_shape(d4);
    const int per = fuse ? fuse_per(p, shape.lpr) : 1;
    const int row_blocks = !(which & 1) ? 0 : fuse ? fusepass_blocks(p, shape.lpr, per, true) : rowpass_blocks(p, shape.lpr);
    const int nb = row_blocks + (!(which & 2) ? 0 : fuse ? fusepass_blocks(p, shape.lpr, per, false) : rowpass_blocks(p, shape.lpr));
    PassSide rs = pass_side(p, t, w, true, want_e, fuse_r, twin), cs = pass_side(p, t, w, false, false, fuse_c, twin);
    rs.mark = mark_rows;
    cs.mark = mark_cols;
    const bool pack = fuse_r == kFusePack || fuse_c == kFusePack;
    if (pack) {
        // the packing build neither applies nor reads plain arrays: records, and no other fuse mode beside it
        if (!packed || p->host_counts[1] < 0 || !p->r_crec || !p->c_crec || want_e || applies) return GLOVE_E_BADARG;
        rs.own_out = cs.own_out = packed;
        rs.own_twin = 1;                                                  // behind the header
        cs.own_twin = 1 + (fuse_r == kFusePack ? p->host_counts[1] : 0);  // and behind the row side's ids when both sides pack
    }
    const StepConsts kc = make_consts(t, h);
    hipStream_t st = (hipStream_t)stream;
    // the twin form's col-side launch: behind the pass, the workgroups that list the ids the apply launch has to visit
    ListTail tail{p->counts, p->r_uniq_rec, p->c_uniq_rec, p->host_counts[1], p->host_counts[3], p->heavy_chunks, per, -1};
    int nb_launch = nb;
    if (list_tail) {
        if (!twin || which != 2 || fuse_c != kFuseInPlace || !w.work || sides_of(h) != 3) return GLOVE_E_BADARG;
        const int64_t ids = tail.nu_r_host >= 0 && tail.nu_c_host >= 0 ? (int64_t)tail.nu_r_host + tail.nu_c_host : 2 * (int64_t)p->cap_uniq;
        tail.first_block = nb;
        nb_launch = nb + (int)((ids + kBlock - 1) / kBlock);
    }
#define ARGS p->counts, rs, cs, row_blocks, t->scalars, t->step, d4, h->inv_batch, w.blockpart, (int)h->head, h->neg_factor, kc, per, w.work, tail
    const bool solid = fuse && !pack && solid_groups(p);        // (the applying run-merged passes: FUSE == 1)
    // the twin form's streaming cache policy: row tables beyond the Infinity Cache's reach (on the 61 MB table of V = 50 k, d = 300
    // it costs 3 - 8 %: the next launch finds the finished rows in the caches there)
    const bool streaming = (size_t)v_row(t) * (size_t)t->d * 4 >= ((size_t)128 << 20);
    // (the diagnostic row pass stores e by pair position, which the records do not carry: it reads the plain arrays)
    const bool rec = p->r_crec != nullptr && p->c_crec != nullptr && !want_e;
    if (!rec && p->B > 0 && !p->r_partner) return GLOVE_E_BADARG;     // (the diagnostic pass needs pair arrays)
// (K: one launch of the build <..., HEAD, SIDE>, in its solid form where the batch calls for it — FUSE == 1 only)
#define K(LPR, NV, FULL, REC, FUSE, HEAD, SIDE, GRID)                                                                          \
    do {                                                                                                                       \
        if (FUSE == 1 && solid)                                                                                                \
            hipLaunchKernelGGL((sidepass_kernel<LPR, NV, FULL, REC, FUSE, HEAD, SIDE, FUSE == 1>), dim3(GRID), dim3(kBlock), 0, st, ARGS); \
        else                                                                                                                   \
            hipLaunchKernelGGL((sidepass_kernel<LPR, NV, FULL, REC, FUSE, HEAD, SIDE, false>), dim3(GRID), dim3(kBlock), 0, st, ARGS); \
    } while (0)
#define LAUNCH(LPR, NV, FULL, REC, FUSE)                                                                                       \
    do {                                                                                                                       \
        /* every pass compiled for the regression head (the same bits are asked of all of them on ids one chunk holds: one     \
         * epilogue, one set of contraction decisions); the logistic heads keep the run-time branch (compiled alone their      \
         * d = 300 build spills) */                                                                                             \
        /* (the side-specialised builds on tables the caches hold — three-launch form and twin form alike: there the col side's     \
         * fourth wave pays, V = 50 k, d = 300 96.8 -> 93.4 us per step; on tables beyond them, bound by bandwidth, it costs:        \
         * V = 400 k 577.6 -> 580.0, V = 2 M 361.5 -> 364.2) */ \
        if (h->head == GLOVE_HEAD_REGRESSION && FUSE == 1 && which == 1 && !(twin && streaming))                                \
            K(LPR, NV, FULL, REC, FUSE, GLOVE_HEAD_REGRESSION, (FUSE == 1 ? 1 : -1), nb);                                        \
        else if (h->head == GLOVE_HEAD_REGRESSION && FUSE == 1 && which == 2 && !(twin && streaming))                           \
            K(LPR, NV, FULL, REC, FUSE, GLOVE_HEAD_REGRESSION, (FUSE == 1 ? 0 : -1), nb_launch);                                 \
        else if (h->head == GLOVE_HEAD_REGRESSION && FUSE == 1 && twin)         /* (tables beyond the caches: the streaming build) */ \
            K(LPR, NV, FULL, REC, FUSE, GLOVE_HEAD_REGRESSION, (FUSE == 1 ? -2 : -1), nb_launch);                                \
        else if (h->head == GLOVE_HEAD_REGRESSION)                                                                              \
            K(LPR, NV, FULL, REC, FUSE, GLOVE_HEAD_REGRESSION, -1, nb);                                                          \
        else if (FUSE == 1 && twin)                  /* (the logistic heads: one twin build, streaming) */                      \
            K(LPR, NV, FULL, REC, FUSE, -1, (FUSE == 1 ? -2 : -1), nb_launch);                                                   \
        else                                                                                                                   \
            K(LPR, NV, FULL, REC, FUSE, -1, -1, nb);                                                                             \
    } while (0)
#define CALL(LPR, NV)                                                                   \
    if (LPR * NV == d4) {                                                               \
        if (rec) { if (pack) LAUNCH(LPR, NV, true, true, 2); else if (fuse) LAUNCH(LPR, NV, true, true, 1); else LAUNCH(LPR, NV, true, true, 0); } \
        else { if (fuse) LAUNCH(LPR, NV, true, false, 1); else LAUNCH(LPR, NV, true, false, 0); }        \
    } else {                                                                            \
        if (rec) { if (pack) LAUNCH(LPR, NV, false, true, 2); else if (fuse) LAUNCH(LPR, NV, false, true, 1); else LAUNCH(LPR, NV, false, true, 0); } \
        else { if (fuse) LAUNCH(LPR, NV, false, false, 1); else LAUNCH(LPR, NV, false, false, 0); }      \
    }
    GLOVE_DISPATCH_PASS_SHAPE(shape, CALL);
#undef CALL
#undef LAUNCH
#undef K
#undef ARGS
    return (int)hipGetLastError();
}

int glove_passes_f32(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws, size_t ws_bytes,
                     void *stream)
{
    if (int rc = plain_table(t, stream)) return rc;
    return launch_passes(p, t, h, ws, ws_bytes, stream, 3);
}

int glove_rowpass_f32(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws, size_t ws_bytes,
                      void *stream)
{
    if (int rc = plain_table(t, stream)) return rc;
    return launch_passes(p, t, h, ws, ws_bytes, stream, 1, nullptr, nullptr, true);
}

int glove_colpass_f32(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws, size_t ws_bytes,
                      void *stream)
{
    if (int rc = plain_table(t, stream)) return rc;
    return launch_passes(p, t, h, ws, ws_bytes, stream, 2);
}

static int launch_apply_adagrad(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws,
                                size_t ws_bytes, float *loss_out, void *stream, int pre_r, int pre_c, bool listed = false)
{
    if (int rc = check_common(p, t, h, ws)) return rc;
    if (!t->s1_R || !t->s1_C || !t->s1_br || !t->s1_bc) return GLOVE_E_BADARG;
    const StepWs w = carve_step_ws(ws, p->B, p->cap_chunks, t->d);
    if (w.bytes > ws_bytes) return GLOVE_E_WORKSPACE;
    const int d4 = t->d / 4;
    const RowShape shape = pick_row_shape(d4);
    IdWork wk = id_work(p);
    wk.sides = sides_of(h);
    wk.pre_r = pre_r;
    wk.pre_c = pre_c;
    const bool fused = pre_r != kFuseNone || pre_c != kFuseNone;
    wk.per = fused ? fuse_per(p, pass_shape(d4).lpr) : 1;
    wk.gpb = fused && solid_groups(p) ? kBlock / pass_shape(d4).lpr : 0;
    const int nb = wk.heavy_blocks + blocks_for(2 * (int64_t)p->cap_uniq, kBlock / shape.lpr) + 1;
    // workgroups of the row-side pass, whose loss partials the scalar duty sums (the first launch of the step)
    const int nb_row = fused ? fusepass_blocks(p, pass_shape(d4).lpr, wk.per, true) : rowpass_blocks(p, pass_shape(d4).lpr);
    const StepConsts k = make_consts(t, h);
    SideBufs rs = side_bufs(p, w, t, true);
    const SideBufs cs = side_bufs(p, w, t, false);
    if (pre_r == kFuseTwin) {
        if (!t->R_ver) return GLOVE_E_BADARG;
        rs.ver = t->R_ver;
        rs.twin = v_row(t);
    }
    hipStream_t st = (hipStream_t)stream;
    const bool short_list = (pre_r == kFuseTwin && pre_c == kFuseInPlace && wk.sides == 3) ||
                            (pre_r == kFuseInPlace && wk.sides == 1);        // glove_rowside_step_adagrad_f32
    if (listed) {
        // the col-side launch drew the list (ListTail); the finished row ids' version flips are this launch's (for_each_id)
        if (!(pre_r == kFuseTwin && pre_c == kFuseInPlace && wk.sides == 3) || !w.work) return GLOVE_E_BADARG;
        wk.work = w.work;
        wk.flips = 1;
    } else if (short_list && w.work) {
        // sort the ids out first (triage_kernel, a thread per id): the launch below then walks the list of those that
        // still need it.  Only where that list is short — the twin form, whose finished ids need a version flip at most
        // (V = 400 k, d = 300: apply 62 -> 12 us + 5 us of triage), and the in-place row side of the sharded forms,
        // whose finished ids need nothing.  Behind the slot form every row id still needs its
        // copy, nearly all ids go onto the list and its one counter becomes the bottleneck (V = 2 M: 51 us of triage)
        wk.work = w.work;
        // (a plan refilled on the device: a thread per id the arrays can hold, the kernel reads the counts)
        const int64_t ids = wk.nu_r_host >= 0 && wk.nu_c_host >= 0 ? (int64_t)wk.nu_r_host + wk.nu_c_host : 2 * (int64_t)p->cap_uniq;
        const int nbt = (int)((ids + kBlock - 1) / kBlock);
        if (nbt > 0) hipLaunchKernelGGL(triage_kernel, dim3(nbt), dim3(kBlock), 0, st, wk, rs, cs, w.work);
    }
#define CALL(LPR, NV)                                                                                          \
    hipLaunchKernelGGL((apply_adagrad_kernel<LPR, NV>), dim3(nb), dim3(kBlock), 0, st, wk, rs, cs, d4,        \
                       k, t->scalars, w.blockpart, nb_row, loss_out)
    GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
    return (int)hipGetLastError();
}

int glove_apply_adagrad_f32(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws,
                            size_t ws_bytes, float *loss_out, void *stream)
{
    if (int rc = plain_table(t, stream)) return rc;
    return launch_apply_adagrad(p, t, h, ws, ws_bytes, loss_out, stream, kFuseNone, kFuseNone);
}

static int launch_dense_grad(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws, size_t ws_bytes,
                             float *G_flat, void *stream)
{
    if (int rc = check_common(p, t, h, ws)) return rc;
    if (!G_flat) return GLOVE_E_BADARG;
    const StepWs w = carve_step_ws(ws, p->B, p->cap_chunks, t->d);
    if (w.bytes > ws_bytes) return GLOVE_E_WORKSPACE;
    const int d4 = t->d / 4;
    const RowShape shape = pick_row_shape(d4);
    IdWork wk = id_work(p);
    wk.sides = sides_of(h);
    const int nb = wk.heavy_blocks + blocks_for(2 * (int64_t)p->cap_uniq, kBlock / shape.lpr) + 1;
    const int nb_row = rowpass_blocks(p, pass_shape(d4).lpr);
    const StepConsts k = make_consts(t, h);
    const SideBufs rs = side_bufs(p, w, t, true), cs = side_bufs(p, w, t, false);
    const GradLayout L = grad_layout(v_row(t), t->V, t->d);
    float *G_R = G_flat + L.G_R, *G_C = G_flat + L.G_C, *G_br = G_flat + L.G_br, *G_bc = G_flat + L.G_bc,
          *tail = G_flat + L.tail;
    hipStream_t st = (hipStream_t)stream;
#define CALL(LPR, NV)                                                                                       \
    hipLaunchKernelGGL((dense_grad_kernel<LPR, NV>), dim3(nb), dim3(kBlock), 0, st, wk, rs, cs, d4,        \
                       k, G_R, G_C, G_br, G_bc, tail, w.blockpart, nb_row)
    GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
    return (int)hipGetLastError();
}

int glove_dense_grad_f32(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws, size_t ws_bytes,
                         float *G_flat, void *stream)
{
    if (int rc = plain_table(t, stream)) return rc;
    return launch_dense_grad(p, t, h, ws, ws_bytes, G_flat, stream);
}

static int dense_common(const glove_tables *t, const glove_hyper *h, float *G_flat, bool adam, DenseSegs &segs,
                        float *&tail, int &nbx, int &sides)
{
    if (!t || !h || !G_flat || t->V <= 0 || t->d <= 0 || (t->d % 4) != 0 || t->d_model < 0 || t->d_model > t->d)
        return GLOVE_E_BADARG;
    if (!t->R || !t->C || !t->br || !t->bc || !t->scalars || !t->step) return GLOVE_E_BADARG;
    if (!t->s1_R || !t->s1_C || !t->s1_br || !t->s1_bc) return GLOVE_E_BADARG;
    if (adam && (!t->s2_R || !t->s2_C || !t->s2_br || !t->s2_bc)) return GLOVE_E_BADARG;
    const int32_t Vr = v_row(t);
    const GradLayout L = grad_layout(Vr, t->V, t->d);
    tail = G_flat + L.tail;
    sides = sides_of(h);
    // segment order = grid.y: 0 R, 1 C (float4 bodies), 2 br, 3 bc (scalar); a side that is not covered gets n = 0
    const int64_t nr = (sides & 1) ? 1 : 0, nc = (sides & 2) ? 1 : 0;
    segs.s[0] = {t->R, t->s1_R, t->s2_R, G_flat + L.G_R, nr * Vr * t->d};
    segs.s[1] = {t->C, t->s1_C, t->s2_C, G_flat + L.G_C, nc * t->V * t->d};
    segs.s[2] = {t->br, t->s1_br, t->s2_br, G_flat + L.G_br, nr * Vr};
    segs.s[3] = {t->bc, t->s1_bc, t->s2_bc, G_flat + L.G_bc, nc * t->V};
    nbx = blocks_for((int64_t)(Vr > t->V ? Vr : t->V) * t->d / 4, kBlock);
    return 0;
}

int glove_dense_adagrad_f32(const glove_tables *t, const glove_hyper *h, float *G_flat, float *loss_out, void *stream)
{
    DenseSegs segs; float *tail; int nbx, sides;
    if (int rc = plain_table(t, stream)) return rc;
    if (int rc = dense_common(t, h, G_flat, false, segs, tail, nbx, sides)) return rc;
    const StepConsts k = make_consts(t, h);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dense_adagrad_kernel, dim3(nbx, 4), dim3(kBlock), 0, st, segs, k, t->scalars, tail, loss_out,
                       (sides & 2) ? 1 : 0);
    return (int)hipGetLastError();
}

int glove_dense_adam_f32(const glove_tables *t, const glove_hyper *h, float *G_flat, float *loss_out, void *stream)
{
    DenseSegs segs; float *tail; int nbx, sides;
    if (int rc = plain_table(t, stream)) return rc;
    if (h && h->optimizer == GLOVE_OPT_RMSPROP) {
        // the other dense-decay optimizer of the Keras set: its whole rms slot decays every step, entries with a gradient move —
        // the sweep of glove_step_sparse_f32, here on a G_flat that holds the ranks' summed gradients (data-parallel step)
        if (int rc = dense_common(t, h, G_flat, false, segs, tail, nbx, sides)) return rc;
        if (!(h->rho > 0.f && h->rho < 1.f) || sides != 3) return GLOVE_E_BADARG;
        hipLaunchKernelGGL(dense_rmsprop_kernel, dim3(nbx, 4), dim3(kBlock), 0, (hipStream_t)stream, segs, make_consts(t, h), h->rho,
                           t->scalars, tail, loss_out, 1);
        return (int)hipGetLastError();
    }
    if (int rc = dense_common(t, h, G_flat, true, segs, tail, nbx, sides)) return rc;
    if (!(h->beta1 > 0.0 && h->beta1 < 1.0 && h->beta2 > 0.0 && h->beta2 < 1.0)) return GLOVE_E_BADARG;
    const StepConsts k = make_consts(t, h);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dense_adam_kernel, dim3(nbx, 4), dim3(kBlock), 0, st, segs, k, (float)h->beta1, (float)h->beta2,
                       log((double)(float)h->beta1), log((double)(float)h->beta2), t->step, t->scalars, tail, loss_out, (sides & 2) ? 1 : 0);
    return (int)hipGetLastError();
}

static_assert(GLOVE_PACKED_ENTRY_FLOATS(0) == (size_t)kPackExtra, "the header's entry size is the kernels'");

int glove_pack_grad_f32(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws, size_t ws_bytes,
                        float *packed, int64_t capacity_entries, void *stream)
{
    if (int rc = check_common(p, t, h, ws)) return rc;
    if (!packed) return GLOVE_E_BADARG;
    if (int rc = plain_table(t, stream)) return rc;
    const StepWs w = carve_step_ws(ws, p->B, p->cap_chunks, t->d);
    if (w.bytes > ws_bytes) return GLOVE_E_WORKSPACE;
    IdWork wk = id_work(p);
    wk.sides = sides_of(h);
    // header + every distinct id of the selected sides must fit (the plan's capacity bounds the counts not yet known here)
    const int64_t nr = (wk.sides & 1) ? (p->host_counts[1] >= 0 ? p->host_counts[1] : p->cap_uniq) : 0;
    const int64_t nc = (wk.sides & 2) ? (p->host_counts[3] >= 0 ? p->host_counts[3] : p->cap_uniq) : 0;
    if (1 + nr + nc > capacity_entries) return GLOVE_E_WORKSPACE;
    const int d4 = t->d / 4;
    const RowShape shape = pick_row_shape(d4);
    const int nb = wk.heavy_blocks + blocks_for(2 * (int64_t)p->cap_uniq, kBlock / shape.lpr) + 1;
    const int nb_row = rowpass_blocks(p, pass_shape(d4).lpr);
    const StepConsts k = make_consts(t, h);
    const SideBufs rs = side_bufs(p, w, t, true), cs = side_bufs(p, w, t, false);
    hipStream_t st = (hipStream_t)stream;
#define CALL(LPR, NV)                                                                                       \
    hipLaunchKernelGGL((pack_grad_kernel<LPR, NV>), dim3(nb), dim3(kBlock), 0, st, wk, rs, cs, d4, k, packed, \
                       w.blockpart, nb_row)
    GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
    return (int)hipGetLastError();
}

// The passes can write packed-list entries themselves when the plan has chunk records (run-merged schedule) and its
// counts are known on the host (the col side's entries start behind the row side's).
static bool packing_ok(const glove_plan *p)
{
    return p->r_crec && p->c_crec && p->host_counts[1] >= 0 && p->host_counts[3] >= 0;
}

int glove_passes_packing_f32(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws, size_t ws_bytes,
                             float *packed, int64_t capacity_entries, void *stream)
{
    if (!p || !h) return GLOVE_E_BADARG;
    if (int rc = plain_table(t, stream)) return rc;
    const int sides = sides_of(h);
    if (!packing_ok(p)) return launch_passes(p, t, h, ws, ws_bytes, stream, sides);
    if (!packed) return GLOVE_E_BADARG;
    const int64_t nr = (sides & 1) ? p->host_counts[1] : 0, nc = (sides & 2) ? p->host_counts[3] : 0;
    if (1 + nr + nc > capacity_entries) return GLOVE_E_WORKSPACE;
    return launch_passes(p, t, h, ws, ws_bytes, stream, sides, nullptr, nullptr, false, (sides & 1) ? kFusePack : kFuseNone,
                         (sides & 2) ? kFusePack : kFuseNone, false, packed);
}

int glove_pack_rest_f32(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws, size_t ws_bytes,
                        float *packed, int64_t capacity_entries, void *stream)
{
    if (!p) return GLOVE_E_BADARG;
    if (!packing_ok(p)) return glove_pack_grad_f32(p, t, h, ws, ws_bytes, packed, capacity_entries, stream);
    if (int rc = check_common(p, t, h, ws)) return rc;
    if (!packed) return GLOVE_E_BADARG;
    if (int rc = plain_table(t, stream)) return rc;
    const StepWs w = carve_step_ws(ws, p->B, p->cap_chunks, t->d);
    if (w.bytes > ws_bytes) return GLOVE_E_WORKSPACE;
    IdWork wk = id_work(p);
    wk.sides = sides_of(h);
    const int64_t nr = (wk.sides & 1) ? p->host_counts[1] : 0, nc = (wk.sides & 2) ? p->host_counts[3] : 0;
    if (1 + nr + nc > capacity_entries) return GLOVE_E_WORKSPACE;
    const int d4 = t->d / 4;
    const RowShape shape = pick_row_shape(d4);
    const int lpr = pass_shape(d4).lpr;
    wk.pre_r = (wk.sides & 1) ? kFusePack : kFuseNone;
    wk.pre_c = (wk.sides & 2) ? kFusePack : kFuseNone;
    wk.per = fuse_per(p, lpr);
    const int nb = wk.heavy_blocks + blocks_for(2 * (int64_t)p->cap_uniq, kBlock / shape.lpr) + 1;
    // the loss partials: of the packing row pass (its grid), or folded by glove_rowside_step_adagrad_f32 (any count reads them)
    const int nb_row = fusepass_blocks(p, lpr, wk.per, true);
    const StepConsts k = make_consts(t, h);
    const SideBufs rs = side_bufs(p, w, t, true), cs = side_bufs(p, w, t, false);
    hipStream_t st = (hipStream_t)stream;
#define CALL(LPR, NV)                                                                                       \
    hipLaunchKernelGGL((pack_grad_kernel<LPR, NV>), dim3(nb), dim3(kBlock), 0, st, wk, rs, cs, d4, k, packed, \
                       w.blockpart, nb_row)
    GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
    return (int)hipGetLastError();
}

int glove_loss_partials_f32(const glove_plan *p, const glove_tables *t, void *ws, size_t ws_bytes, float *out4, void *stream)
{
    if (!p || !t || !ws || !out4 || t->d <= 0 || (t->d % 4) != 0) return GLOVE_E_BADARG;
    const StepWs w = carve_step_ws(ws, p->B, p->cap_chunks, t->d);
    if (w.bytes > ws_bytes) return GLOVE_E_WORKSPACE;
    const int lpr = pass_shape(t->d / 4).lpr;
    if (lpr == 0) return GLOVE_E_BADARG;
    // the grid of the row pass that left them (folded by glove_rowside_step_adagrad_f32: any count reads the same)
    const int nb_row = packing_ok(p) ? fusepass_blocks(p, lpr, fuse_per(p, lpr), true) : rowpass_blocks(p, lpr);
    hipLaunchKernelGGL(loss_partials_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, w.blockpart, nb_row, out4);
    return (int)hipGetLastError();
}

static int packed_common(const glove_tables *t, float *G_flat, int32_t *mark, DenseViews &dv)
{
    if (!t || !G_flat || !mark || t->V <= 0 || t->d <= 0 || (t->d % 4) != 0) return GLOVE_E_BADARG;
    const int32_t Vr = v_row(t);
    const GradLayout L = grad_layout(Vr, t->V, t->d);
    dv = {G_flat + L.G_R, G_flat + L.G_br, G_flat + L.G_C, G_flat + L.G_bc, mark, (int)Vr};
    return pick_row_shape(t->d / 4).lpr == 0 ? GLOVE_E_BADARG : 0;
}

static bool to_list(const glove_packed_list *in, PackedList &out)
{
    if (!in || !in->entries || (!in->header && in->n < 0) || in->side < -1 || in->side > 1) return false;
    out = {in->entries, in->ids, in->header, (int)in->n, (int)in->side};
    return true;
}

int glove_combine_packed_f32(const glove_packed_list *list, int32_t tag, const glove_tables *t, float *G_flat,
                             int32_t *mark, int64_t capacity_entries, void *stream)
{
    DenseViews dv;
    if (int rc = packed_common(t, G_flat, mark, dv)) return rc;
    PackedList pl;
    if (!to_list(list, pl) || tag < 0 || capacity_entries < 0) return GLOVE_E_BADARG;
    const int64_t n = pl.header ? capacity_entries : pl.n_host;
    if (n == 0) return 0;
    const int d4 = t->d / 4;
    const RowShape shape = pick_row_shape(d4);
    const int nb = blocks_for(n, kBlock / shape.lpr);
    hipStream_t st = (hipStream_t)stream;
#define CALL(LPR, NV) hipLaunchKernelGGL((combine_packed_kernel<LPR, NV>), dim3(nb), dim3(kBlock), 0, st, pl, (int)tag, dv, d4)
    GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
    return (int)hipGetLastError();
}

int glove_count_packed_f32(const glove_packed_list *lists, int32_t n_lists, const glove_tables *t, float *G_flat,
                           int32_t *mark, int64_t capacity_entries, void *stream)
{
    DenseViews dv;
    if (int rc = packed_common(t, G_flat, mark, dv)) return rc;
    if (!lists || n_lists < 1 || n_lists > kMarkCountMask || capacity_entries < 0) return GLOVE_E_BADARG;
    const int d4 = t->d / 4;
    hipStream_t st = (hipStream_t)stream;
    for (int32_t first = 0; first < n_lists; first += 8) {
        PackedLists pls;
        pls.n = n_lists - first < 8 ? n_lists - first : 8;
        int64_t n_max = 0;
        for (int i = 0; i < pls.n; ++i) {
            if (!to_list(lists + first + i, pls.l[i])) return GLOVE_E_BADARG;
            const int64_t n = pls.l[i].header ? capacity_entries : pls.l[i].n_host;
            n_max = n > n_max ? n : n_max;
        }
        if (n_max > 0)
            hipLaunchKernelGGL(count_packed_kernel, dim3(blocks_for(n_max, kBlock), pls.n), dim3(kBlock), 0, st, pls, dv, d4);
    }
    return (int)hipGetLastError();
}

int glove_apply_packed_adagrad_f32(const glove_packed_list *lists, int32_t n_lists, const glove_tables *t,
                                   const glove_hyper *h, float *G_flat, int32_t *mark, const float *tail,
                                   float *loss_out, int64_t capacity_entries, void *stream)
{
    DenseViews dv;
    if (int rc = packed_common(t, G_flat, mark, dv)) return rc;
    if (!h || !lists || n_lists < 1 || capacity_entries < 0) return GLOVE_E_BADARG;
    if (!t->R || !t->C || !t->br || !t->bc || !t->scalars || !t->s1_R || !t->s1_C || !t->s1_br || !t->s1_bc) return GLOVE_E_BADARG;
    // glove_hyper.optimizer: Adagrad, or one of the per-row Keras optimizers (only touched rows move: the exchange is the same)
    const int opt = h->optimizer;
    const bool two_slots = opt == GLOVE_OPT_ADAMAX || opt == GLOVE_OPT_ADADELTA || opt == GLOVE_OPT_FTRL || opt == GLOVE_OPT_NADAM;
    if (opt != GLOVE_OPT_ADAGRAD && opt != GLOVE_OPT_SGD && !two_slots) return GLOVE_E_BADARG;
    if (two_slots && (!t->s2_R || !t->s2_C || !t->s2_br || !t->s2_bc)) return GLOVE_E_BADARG;
    if ((opt == GLOVE_OPT_ADAMAX || opt == GLOVE_OPT_NADAM) &&
        (!(h->beta1 > 0.0 && h->beta1 < 1.0) || !(h->beta2 > 0.0 && h->beta2 < 1.0) || !t->step)) return GLOVE_E_BADARG;
    if (opt == GLOVE_OPT_NADAM && sides_of(h) != 3) return GLOVE_E_BADARG;                 // (m and v of BOTH tables decay every step)
    if (opt == GLOVE_OPT_ADADELTA && !(h->rho > 0.f && h->rho < 1.f)) return GLOVE_E_BADARG;
    if (opt == GLOVE_OPT_FTRL && !(h->learning_rate > 0.f)) return GLOVE_E_BADARG;
    if (opt == GLOVE_OPT_SGD && !(h->momentum >= 0.f && h->momentum < 1.f)) return GLOVE_E_BADARG;
    const OptConsts o = {h->learning_rate, h->epsilon, h->momentum, 0.f, (float)h->beta1, (float)h->beta2, h->nesterov ? 1 : 0, h->rho};
    const double ln_b1 = opt == GLOVE_OPT_ADAMAX ? log((double)(float)h->beta1) : opt == GLOVE_OPT_NADAM ? log((double)(float)h->beta2) : 0.0;
    const SlotTwo s2 = {t->s2_R, t->s2_C, t->s2_br, t->s2_bc};
    if (int rc = plain_table(t, stream)) return rc;
    const int d4 = t->d / 4;
    const RowShape shape = pick_row_shape(d4);
    const StepConsts k = make_consts(t, h);
    SideBufs rs = {nullptr, nullptr, nullptr, t->R, t->s1_R, t->br, t->s1_br, nullptr, 0};
    SideBufs cs = {nullptr, nullptr, nullptr, t->C, t->s1_C, t->bc, t->s1_bc, nullptr, 0};
    hipStream_t st = (hipStream_t)stream;
    // the scalar work (global bias, loss) goes with the col side, like everywhere else; without an explicit tail it
    // sums the lists' headers in list order
    const int do_scalars = (sides_of(h) & 2) ? 1 : 0;
    float *scratch = nullptr;
    if (!tail && n_lists > 8 && do_scalars) {
        // a launch sees eight lists: sum the loss partials of all headers first (list order, so the sum is repeatable)
        scratch = G_flat + grad_layout(v_row(t), t->V, t->d).tail + 4;
        for (int32_t first = 0; first < n_lists; first += 8) {
            PackedLists pls;
            pls.n = n_lists - first < 8 ? n_lists - first : 8;
            for (int i = 0; i < pls.n; ++i)
                if (!to_list(lists + first + i, pls.l[i])) return GLOVE_E_BADARG;
            hipLaunchKernelGGL(sum_headers_kernel, dim3(1), dim3(64), 0, st, pls, scratch, first == 0 ? 1 : 0);
        }
        tail = scratch;
    }
    if (opt == GLOVE_OPT_NADAM) {
        // the rows nobody touched: m and v decay (the marks are still whole: the apply launches below clear them as they go)
        const int nbd = blocks_for((int64_t)v_row(t) + t->V, kBlock / shape.lpr);
#define CALL(LPR, NV)                                                                                               \
        hipLaunchKernelGGL((nadam_decay_unmarked_kernel<LPR, NV>), dim3(nbd), dim3(kBlock), 0, st, dv, rs, cs, s2, d4,   \
                           (float)h->beta1, (float)h->beta2, (int)t->V)
        GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
    }
    for (int32_t first = 0; first < n_lists; first += 8) {
        PackedLists pls;
        pls.n = n_lists - first < 8 ? n_lists - first : 8;
        int64_t n_max = 0;
        for (int i = 0; i < pls.n; ++i) {
            if (!to_list(lists + first + i, pls.l[i])) return GLOVE_E_BADARG;
            const int64_t n = pls.l[i].header ? capacity_entries : pls.l[i].n_host;
            n_max = n > n_max ? n : n_max;
        }
        const int nbx = blocks_for(n_max > 0 ? n_max : 1, kBlock / shape.lpr);
        const int scal = first == 0 ? do_scalars : 0;
#define LAUNCH_OPT(LPR, NV, OPT)                                                                                     \
        hipLaunchKernelGGL((apply_packed_kernel<LPR, NV, OPT>), dim3(nbx, pls.n), dim3(kBlock), 0, st, pls, dv, rs, cs, s2, \
                           d4, k, o, ln_b1, (const int64_t *)t->step, (int)first, tail, t->scalars, loss_out, scal)
#define CALL(LPR, NV)                                                                                               \
        if (opt == GLOVE_OPT_ADAGRAD) LAUNCH_OPT(LPR, NV, GLOVE_OPT_ADAGRAD);                                       \
        else if (opt == GLOVE_OPT_SGD) LAUNCH_OPT(LPR, NV, GLOVE_OPT_SGD);                                          \
        else if (opt == GLOVE_OPT_ADAMAX) LAUNCH_OPT(LPR, NV, GLOVE_OPT_ADAMAX);                                    \
        else if (opt == GLOVE_OPT_ADADELTA) LAUNCH_OPT(LPR, NV, GLOVE_OPT_ADADELTA);                                \
        else if (opt == GLOVE_OPT_NADAM) LAUNCH_OPT(LPR, NV, GLOVE_OPT_NADAM);                                      \
        else LAUNCH_OPT(LPR, NV, GLOVE_OPT_FTRL)
        GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef LAUNCH_OPT
#undef CALL
    }
    if (scratch) {
        hipError_t e = zero_words(scratch, 4, st);      // the tail is all zero between steps
        if (e != hipSuccess) return (int)e;
    }
    return (int)hipGetLastError();
}

int glove_gather_rows_f32(const float *W, const float *bias, const int32_t *ids, int32_t n, int32_t d, float *rows,
                          float *biases, void *stream)
{
    if (n < 0 || d <= 0 || (d % 4) != 0) return GLOVE_E_BADARG;
    if (n == 0) return 0;
    if (!W || !bias || !ids || !rows || !biases) return GLOVE_E_BADARG;
    const int d4 = d / 4;
    const RowShape shape = pick_row_shape(d4);
    if (shape.lpr == 0) return GLOVE_E_BADARG;
    const int nb = blocks_for(n, kBlock / shape.lpr);
    hipStream_t st = (hipStream_t)stream;
#define CALL(LPR, NV) hipLaunchKernelGGL((gather_rows_kernel<LPR, NV>), dim3(nb), dim3(kBlock), 0, st, W, bias, ids, (int)n, d4, rows, biases)
    GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
    return (int)hipGetLastError();
}

// Forms 1 / 3 / 4 on the same resident plans, one process, interleaved rounds (tools/ab_step_forms.py, round 4; us per step):
//   V = 50 k,  d = 300 (table 61 MB):  146 MB of touched rows  71.0 /  70.7 /  72.8    221 MB 105.9 / 101.0 / 103.2    311 MB 175.3 / 152.5 / 153.0
//   V = 400 k, d = 300 (486 MB):       226 MB 102.5 / 104.8 /  98.7                    388 MB 178.2 / 163.6 / 151.9    649 MB 333.0 / 277.1 / 247.5
//   V = 2 M,   d = 128 (1 GB):         209 MB 128.6 / 122.9 / 120.6                    368 MB 220.3 / 199.4 / 188.8
//   V = 10 k,  d = 64, B = 1 M (20 MB touched): 55.2 / 82.3 / 87.7
// A tie at 146 MB, 4 - 6 % for the fused forms from 209 MB on (the twin form on tables beyond the Infinity Cache, the
// three-launch form on the 61 MB table): the switch sat between (192 MB).
// Round 5 (the passes compiled for one head, staged version lookups, six chunks per lane group; same tool, forms 1 / 3 / 4):
//   V = 50 k,  d = 300:  56 MB 32.5 / 39.1 / 40.8     75 MB 41.1 / 41.6 / 42.8     92 MB 46.2 / 44.3 / 46.5    122 MB 57.6 / 54.4 / 58.5   148 MB 69.2 / 62.5 / 67.5
//   V = 400 k, d = 300:  77 MB 42.2 / 44.2 / 46.5    137 MB 60.3 / 58.0 / 60.9    241 MB 98.3 / 93.5 / 89.6
//   V = 2 M,   d = 128:  35 MB 26.0 / 31.8 / 35.9     64 MB 42.2 / 40.2 / 41.8    117 MB 67.5 / 63.9 / 63.1    209 MB 114.3 / 114.4 / 104.5
// The fused forms now pay from 90 MB on: the switch is GLOVE_FUSED_STEP_BYTES = 96 MB.

// Which form a sparse Adagrad step takes (glove_hyper.step_form; see include/glove_hip.h).
static int pick_step_form(const glove_plan *p, const glove_tables *t, const glove_hyper *h)
{
    if (!p || !t || !h) return GLOVE_STEP_TWO_LAUNCH;
    if (sides_of(h) != 3) return GLOVE_STEP_TWO_LAUNCH;
    // the fused forms read the id layout from the chunk records, or from the run words of a plan that keeps pair arrays instead
    const bool records = p->r_crec && p->c_crec;
    const bool run_words = p->r_chunk_hw && p->c_chunk_hw && p->r_partner && p->r_w && p->r_y && p->c_partner && p->c_w && p->c_y;
    if (!records && !run_words) return GLOVE_STEP_TWO_LAUNCH;
    if (!records && h->step_form == GLOVE_STEP_TAGGED) return GLOVE_STEP_TWO_LAUNCH;        // (the one-launch form walks records)
    if (h->step_form != GLOVE_STEP_AUTO) return h->step_form;
    // the latency-bound regime on step-tagged tables: everything in one launch
    // ... unless the batch is known to hold heavy ids (more than plan.heavy_chunks chunks: a frequent token with hundreds of the
    // batch's pairs): the tagged form walks ALL chunks of an id in one lane group, one record trip after the other, and beyond
    // heavy_chunks its summation order is no longer the two-launch form's.  (A plan refilled on the device keeps its counts
    // there: host_counts[4] < 0, judged by the batch size alone.)
    if (t->R_tag && t->C_tag && p->B <= 2048 && records && p->host_counts[4] <= 0) return GLOVE_STEP_TAGGED;     // (measured against the two-launch form, 64 staging plans: 512 / 1,024 / 2,048 pairs -16 / -16 / -13 %, 4,096 +12 %)
    // the fused forms pay off once the touched rows and their partials no longer live in the caches; the id counts
    // are known on the host for a plan whose build has been synchronised (a resident plan)
    // (a plan refilled on the device every step — a reshuffled epoch — is judged by the most ids its batch can hold)
    const int64_t most = (int64_t)v_row(t) + t->V < 2 * p->B ? (int64_t)v_row(t) + t->V : 2 * p->B;
    const int64_t ids = p->host_counts[1] >= 0 && p->host_counts[3] >= 0 ? (int64_t)p->host_counts[1] + p->host_counts[3] : most;
    // (V = 50 k, d = 300, B = 131,072: 48 k ids = 230 MB per step, all of it living in the Infinity Cache: two launches
    // 106 us, fused 103 - 111; V = 400 k at B = 131,072, 336 MB: 172 against 156; V = 50 k at B = 1 M, 432 MB: 378 against 314)
    if (ids * t->d * 16 < (int64_t)GLOVE_FUSED_STEP_BYTES) return GLOVE_STEP_TWO_LAUNCH;        // (the table of measurements above)
    return t->R_ver ? GLOVE_STEP_FUSED_TWIN : GLOVE_STEP_FUSED_THREE_LAUNCH;
}

// n consecutive tagged steps as chains: one launch per step, the global bias handed from step to step through chain records in
// the workspace (the tagged form keeps nothing else there: no partial rows), one epilogue launch per chain.
static int launch_tagged_chain(const glove_plan *const *plans, int n, const glove_tables *t, const glove_hyper *h, void *ws,
                               size_t ws_bytes, float *loss_out, void *stream)
{
    if (!plans || n < 1 || !t || !h || !ws) return GLOVE_E_BADARG;
    const int d4 = t->d / 4;
    int most_blocks = 1;
    for (int i = 0; i < n; ++i) {
        const glove_plan *p = plans[i];
        if (int rc = check_common(p, t, h, ws)) return rc;
        if (!t->R_tag || !t->C_tag || !p->r_crec || !p->c_crec) return GLOVE_E_BADARG;
        const int rb = rowpass_blocks(p, pass_shape(d4).lpr);
        most_blocks = rb > most_blocks ? rb : most_blocks;
    }
    if (!t->s1_R || !t->s1_C || !t->s1_br || !t->s1_bc || sides_of(h) != 3) return GLOVE_E_BADARG;
    const size_t rec_floats = (size_t)kChainHead + (size_t)kPartials * most_blocks;
    const size_t fit = ws_bytes / (rec_floats * sizeof(float));
    if (fit < 1) return GLOVE_E_WORKSPACE;
    const RowShape shape = pass_shape(d4);
    const int Vr = v_row(t);
    const StepConsts kc = make_consts(t, h);
    hipStream_t st = (hipStream_t)stream;
    float *recs = (float *)ws;
    for (int i0 = 0; i0 < n; i0 += (int)fit) {
        const int m = n - i0 < (int)fit ? n - i0 : (int)fit;
        int prev_blocks = 0;
        for (int i = 0; i < m; ++i) {
            const glove_plan *p = plans[i0 + i];
            const int row_blocks = rowpass_blocks(p, shape.lpr), nb = 2 * row_blocks;     // the classic passes' grid and chunk assignment
            OneSide rs, cs;
            rs.crec = p->r_crec; rs.own = t->R; rs.own_bias = t->br; rs.other = t->C; rs.other_bias = t->bc;
            rs.S1 = t->s1_R; rs.S1b = t->s1_br; rs.own_tag = t->R_tag; rs.other_tag = t->C_tag; rs.own_twin = Vr; rs.other_twin = t->V;
            rs.n_host = p->host_counts[0]; rs.count_index = 0; rs.capP = rec_cap(p->chunk_cap);
            cs.crec = p->c_crec; cs.own = t->C; cs.own_bias = t->bc; cs.other = t->R; cs.other_bias = t->br;
            cs.S1 = t->s1_C; cs.S1b = t->s1_bc; cs.own_tag = t->C_tag; cs.other_tag = t->R_tag; cs.own_twin = t->V; cs.other_twin = Vr;
            cs.n_host = p->host_counts[2]; cs.count_index = 2; cs.capP = rs.capP;
            rs.cap_chunks = cs.cap_chunks = p->cap_chunks > 0 ? p->cap_chunks : 1;
            const float *prev = i > 0 ? recs + (size_t)(i - 1) * rec_floats : nullptr;
            float *mine = recs + (size_t)i * rec_floats;
#define CALL(LPR, NV)                                                                                                           \
            if (LPR * NV == d4)                                                                                                 \
                hipLaunchKernelGGL((tagged_step_kernel<LPR, NV, true>), dim3(nb), dim3(kBlock), 0, st, p->counts, rs, cs, row_blocks, \
                                   (const float *)t->scalars, (const int64_t *)t->step, d4, h->inv_batch, (int)h->head,        \
                                   h->neg_factor, kc, prev, prev_blocks, mine, i);                                             \
            else                                                                                                                \
                hipLaunchKernelGGL((tagged_step_kernel<LPR, NV, false>), dim3(nb), dim3(kBlock), 0, st, p->counts, rs, cs, row_blocks, \
                                   (const float *)t->scalars, (const int64_t *)t->step, d4, h->inv_batch, (int)h->head,        \
                                   h->neg_factor, kc, prev, prev_blocks, mine, i)
            GLOVE_DISPATCH_PASS_SHAPE(shape, CALL);
#undef CALL
            prev_blocks = row_blocks;
        }
        hipLaunchKernelGGL(tagged_flush_kernel, dim3(1), dim3(kBlock), 0, st, t->scalars, t->step,
                           (const float *)(recs + (size_t)(m - 1) * rec_floats), prev_blocks, m, kc, i0 + m == n ? loss_out : nullptr);
    }
    return (int)hipGetLastError();
}

int glove_step_adagrad_f32(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws, size_t ws_bytes,
                           float *loss_out, void *stream)
{
    const int form = pick_step_form(p, t, h);
    if (form == GLOVE_STEP_TAGGED) return launch_tagged_chain(&p, 1, t, h, ws, ws_bytes, loss_out, stream);
    // only the twin form follows the version bytes of a twinned row table: every other form first brings it home
    // (a step of another form behind a twin step would otherwise read and write copy 0 of rows whose current copy is the second)
    if (form != GLOVE_STEP_FUSED_TWIN)
        if (int rc = plain_table(t, stream)) return rc;
    switch (form) {
    case GLOVE_STEP_FUSED_ONE_PASS:
        // (tests / comparisons) both sides in one launch: neither table may change under the other side's gathers, so both put
        // their finished rows into the slots and the apply launch moves them
        if (int rc = launch_passes(p, t, h, ws, ws_bytes, stream, 3, nullptr, nullptr, false, kFuseSlot, kFuseSlot)) return rc;
        return launch_apply_adagrad(p, t, h, ws, ws_bytes, loss_out, stream, kFuseSlot, kFuseSlot);
    case GLOVE_STEP_FUSED_THREE_LAUNCH:
        // row side first (C is read only); then the col side alone may update C in place while it gathers the
        // still unchanged R; the apply launch moves the row side's finished rows and does the multi-chunk ids
        if (int rc = launch_passes(p, t, h, ws, ws_bytes, stream, 1, nullptr, nullptr, false, kFuseSlot, kFuseNone)) return rc;
        if (int rc = launch_passes(p, t, h, ws, ws_bytes, stream, 2, nullptr, nullptr, false, kFuseNone, kFuseInPlace)) return rc;
        return launch_apply_adagrad(p, t, h, ws, ws_bytes, loss_out, stream, kFuseSlot, kFuseInPlace);
    case GLOVE_STEP_FUSED_TWIN:
        // as the three-launch form, but the row side writes its new rows into the other copy of the twinned row table:
        // the col side keeps gathering the old rows from the current copy, and the apply launch only flips versions
        if (!t->R_ver) return GLOVE_E_BADARG;
        if (int rc = launch_passes(p, t, h, ws, ws_bytes, stream, 1, nullptr, nullptr, false, kFuseTwin, kFuseNone, true)) return rc;
        {
            // (the list of ids for the apply launch rides in the col-side launch when both sides step and the workspace has the list)
            const StepWs w = carve_step_ws(ws, p->B, p->cap_chunks, t->d);
            const bool tail = w.work != nullptr && sides_of(h) == 3;
            if (int rc = launch_passes(p, t, h, ws, ws_bytes, stream, 2, nullptr, nullptr, false, kFuseNone, kFuseInPlace, true, nullptr, tail)) return rc;
            return launch_apply_adagrad(p, t, h, ws, ws_bytes, loss_out, stream, kFuseTwin, kFuseInPlace, tail);
        }
    case GLOVE_STEP_TWO_LAUNCH:
        break;
    default:
        return GLOVE_E_BADARG;
    }
    if (int rc = launch_passes(p, t, h, ws, ws_bytes, stream, 3)) return rc;
    return launch_apply_adagrad(p, t, h, ws, ws_bytes, loss_out, stream, kFuseNone, kFuseNone);
}

int glove_rowside_step_adagrad_f32(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws, size_t ws_bytes,
                                   void *stream)
{
    if (!p || !t || !h || sides_of(h) != 1) return GLOVE_E_BADARG;          // hyper.sides = 1: this is the row side's step
    if (int rc = plain_table(t, stream)) return rc;
    if (!p->r_crec || !p->c_crec) {
        // no chunk records: the row pass stores its partial rows, the apply launch does every row id
        if (int rc = launch_passes(p, t, h, ws, ws_bytes, stream, 1)) return rc;
        return launch_apply_adagrad(p, t, h, ws, ws_bytes, nullptr, stream, kFuseNone, kFuseNone);
    }
    if (int rc = launch_passes(p, t, h, ws, ws_bytes, stream, 1, nullptr, nullptr, false, kFuseInPlace, kFuseNone)) return rc;
    if (int rc = launch_apply_adagrad(p, t, h, ws, ws_bytes, nullptr, stream, kFuseInPlace, kFuseNone)) return rc;
    const StepWs w = carve_step_ws(ws, p->B, p->cap_chunks, t->d);
    const int lpr = pass_shape(t->d / 4).lpr;
    const int nb_fused = fusepass_blocks(p, lpr, fuse_per(p, lpr), true);
    hipLaunchKernelGGL(fold_blockpart_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, w.blockpart, nb_fused);
    return (int)hipGetLastError();
}

int glove_canonicalize_f32(const glove_tables *t, void *stream)
{
    if (!t || !t->R || !t->br || t->V <= 0 || t->d <= 0 || (t->d % 4) != 0) return GLOVE_E_BADARG;
    if (!t->R_ver && !t->R_tag) return 0;
    const int d4 = t->d / 4;
    const RowShape shape = pick_row_shape(d4);
    if (shape.lpr == 0) return GLOVE_E_BADARG;
    const int Vr = v_row(t);
    const int nb = blocks_for(Vr, kBlock / shape.lpr);
    hipStream_t st = (hipStream_t)stream;
    if (t->R_tag) {                             // step-tagged twins of both tables (the one-launch step)
        if (!t->C_tag || !t->C || !t->bc || t->R_ver) return GLOVE_E_BADARG;
        const int nbc = blocks_for(t->V, kBlock / shape.lpr);
#define CALL(LPR, NV)                                                                                                  \
        hipLaunchKernelGGL((untag_kernel<LPR, NV>), dim3(nb), dim3(kBlock), 0, st, t->R, t->br, t->R_tag, Vr, d4);     \
        hipLaunchKernelGGL((untag_kernel<LPR, NV>), dim3(nbc), dim3(kBlock), 0, st, t->C, t->bc, t->C_tag, (int)t->V, d4)
        GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
        // the one-launch Adam step flips the tables as a whole (scalars[3]) instead of tagging rows
        if (t->scalars) {
#define CALL(LPR, NV)                                                                                                  \
            hipLaunchKernelGGL((twin_home_kernel<LPR, NV>), dim3(nb), dim3(kBlock), 0, st, t->R, t->br, (const float *)t->scalars, Vr, d4);  \
            hipLaunchKernelGGL((twin_home_kernel<LPR, NV>), dim3(nbc), dim3(kBlock), 0, st, t->C, t->bc, (const float *)t->scalars, (int)t->V, d4)
            GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
            hipLaunchKernelGGL(twin_home_done_kernel, dim3(1), dim3(64), 0, st, t->scalars);
        }
        return (int)hipGetLastError();
    }
#define CALL(LPR, NV) hipLaunchKernelGGL((canonicalize_kernel<LPR, NV>), dim3(nb), dim3(kBlock), 0, st, t->R, t->br, t->R_ver, Vr, d4)
    GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
    return (int)hipGetLastError();
}

int glove_steps_adagrad_f32(const glove_plan *const *plans, int32_t n, const glove_tables *t, const glove_hyper *h,
                            void *ws, size_t ws_bytes, float *loss_out, void *stream)
{
    if (!plans || n < 0) return GLOVE_E_BADARG;
    // runs of consecutive steps that take the tagged form go out as chains: one launch per step
    for (int32_t i = 0; i < n;) {
        int32_t k = i;
        while (k < n && plans[k] && pick_step_form(plans[k], t, h) == GLOVE_STEP_TAGGED) ++k;
        if (k > i) {
            if (int rc = launch_tagged_chain(plans + i, k - i, t, h, ws, ws_bytes, k == n ? loss_out : nullptr, stream)) return rc;
            i = k;
            continue;
        }
        if (int rc = glove_step_adagrad_f32(plans[i], t, h, ws, ws_bytes, i == n - 1 ? loss_out : nullptr, stream)) return rc;
        ++i;
    }
    return 0;
}

// passes (marking the batch's ids) + adam_fused_kernel: see the kernel's header
static int step_adam_fused(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws, size_t ws_bytes,
                           float *G_flat, float *loss_out, void *stream, bool nadam = false)
{
    if (int rc = check_common(p, t, h, ws)) return rc;
    if (!G_flat || !t->s1_R || !t->s1_C || !t->s1_br || !t->s1_bc || !t->s2_R || !t->s2_C || !t->s2_br || !t->s2_bc)
        return GLOVE_E_BADARG;
    if (!(h->beta1 > 0.0 && h->beta1 < 1.0 && h->beta2 > 0.0 && h->beta2 < 1.0)) return GLOVE_E_BADARG;
    const int32_t Vr = v_row(t);
    const GradLayout L = grad_layout(Vr, t->V, t->d);
    float *mark_rows = G_flat + L.G_br, *mark_cols = G_flat + L.G_bc;
    if (int rc = launch_passes(p, t, h, ws, ws_bytes, stream, 3, mark_rows, mark_cols)) return rc;
    const StepWs w = carve_step_ws(ws, p->B, p->cap_chunks, t->d);
    const int d4 = t->d / 4;
    const RowShape shape = pick_row_shape(d4);
    IdWork wk = id_work(p);
    wk.sides = 3;
    // a multiple of 8 workgroups in front: workgroups b and b + 8 share an XCD, so every step's sweep then finds
    // the rows it wrote last step in the same L2s, whatever the batch's id count (5.3 vs 11 us per launch)
    const int apply_blocks = (wk.heavy_blocks + blocks_for(2 * (int64_t)p->cap_uniq, kBlock / shape.lpr) + 1 + 7) & ~7;
    const int nb = apply_blocks + blocks_for(((int64_t)Vr + t->V + kSweepRows - 1) / kSweepRows, kBlock / shape.lpr);
    const int nb_row = rowpass_blocks(p, pass_shape(d4).lpr);
    const StepConsts k = make_consts(t, h);
    const SideBufs rs = side_bufs(p, w, t, true), cs = side_bufs(p, w, t, false);
    hipStream_t st = (hipStream_t)stream;
#define CALL(LPR, NV)                                                                                          \
    if (nadam)                                                                                                 \
        hipLaunchKernelGGL((nadam_fused_kernel<LPR, NV>), dim3(nb), dim3(kBlock), 0, st, wk, rs, cs, t->s2_R, t->s2_C, \
                           t->s2_br, t->s2_bc, d4, k, (float)h->beta1, (float)h->beta2, log((double)(float)h->beta2), \
                           t->step, t->scalars, w.blockpart, nb_row, mark_rows, mark_cols, (int)Vr, (int)t->V, \
                           apply_blocks, loss_out);                                                            \
    else                                                                                                       \
        hipLaunchKernelGGL((adam_fused_kernel<LPR, NV>), dim3(nb), dim3(kBlock), 0, st, wk, rs, cs, t->s2_R, t->s2_C, \
                           t->s2_br, t->s2_bc, d4, k, (float)h->beta1, (float)h->beta2, log((double)(float)h->beta1), log((double)(float)h->beta2),  \
                           t->step, t->scalars, w.blockpart, nb_row, mark_rows, mark_cols, (int)Vr, (int)t->V, \
                           apply_blocks, loss_out)
    GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
    return (int)hipGetLastError();
}

// Whether an Adam step takes the one-launch form (tagged_adam_kernel): twinned tables (glove_tables.R_tag / C_tag are the
// sign; the tags themselves stay zero), a plan with chunk records and id bitmaps, a small batch that touches a minority of the rows
static bool adam_one_launch(const glove_plan *p, const glove_tables *t, const glove_hyper *h)
{
    if (!p || !t || !h || sides_of(h) != 3) return false;
    if (!t->R_tag || !t->C_tag || !p->r_crec || !p->c_crec || !p->r_mark || !p->c_mark) return false;
    if (h->step_form != GLOVE_STEP_AUTO && h->step_form != GLOVE_STEP_TAGGED) return false;
    if (h->step_form == GLOVE_STEP_AUTO && p->host_counts[4] > 0) return false;     // heavy ids: as pick_step_form
    return p->B <= 2048 && 2 * p->B <= (int64_t)v_row(t) + t->V;
}

// n consecutive one-launch Adam steps: chain records in the workspace as launch_tagged_chain keeps them, one epilogue per chain
static int launch_adam_chain(const glove_plan *const *plans, int n, const glove_tables *t, const glove_hyper *h, void *ws,
                             size_t ws_bytes, float *loss_out, void *stream)
{
    if (!plans || n < 1 || !t || !h || !ws) return GLOVE_E_BADARG;
    const int d4 = t->d / 4;
    int most_blocks = 1;
    for (int i = 0; i < n; ++i) {
        const glove_plan *p = plans[i];
        if (int rc = check_common(p, t, h, ws)) return rc;
        if (!adam_one_launch(p, t, h)) return GLOVE_E_BADARG;
        const int rb = rowpass_blocks(p, pass_shape(d4).lpr);
        most_blocks = rb > most_blocks ? rb : most_blocks;
    }
    if (!t->s1_R || !t->s1_C || !t->s1_br || !t->s1_bc || !t->s2_R || !t->s2_C || !t->s2_br || !t->s2_bc) return GLOVE_E_BADARG;
    if (!(h->beta1 > 0.0 && h->beta1 < 1.0 && h->beta2 > 0.0 && h->beta2 < 1.0)) return GLOVE_E_BADARG;
    const size_t rec_floats = (size_t)kChainHead + (size_t)kPartials * most_blocks;
    const size_t fit = ws_bytes / (rec_floats * sizeof(float));
    if (fit < 1) return GLOVE_E_WORKSPACE;
    const RowShape shape = pass_shape(d4);
    const int Vr = v_row(t);
    const StepConsts kc = make_consts(t, h);
    const float b1 = (float)h->beta1, b2 = (float)h->beta2;
    const double ln_b1 = log((double)b1), ln_b2 = log((double)b2);
    hipStream_t st = (hipStream_t)stream;
    float *recs = (float *)ws;
    // the sweep's part of the grid: as adam_fused_kernel's; a multiple of 8 workgroups in front of it, so that every step's sweep
    // finds the rows it wrote last step in the same L2s
    const int sweep_rows = tagged_sweep_rows(shape.nv);
    const int sweep_blocks = blocks_for(((int64_t)Vr + t->V + sweep_rows - 1) / sweep_rows, kBlock / shape.lpr);
    for (int i0 = 0; i0 < n; i0 += (int)fit) {
        const int m = n - i0 < (int)fit ? n - i0 : (int)fit;
        int prev_blocks = 0;
        for (int i = 0; i < m; ++i) {
            const glove_plan *p = plans[i0 + i];
            const int row_blocks = rowpass_blocks(p, shape.lpr), chunk_blocks = (2 * row_blocks + 7) & ~7;
            const int nb = chunk_blocks + sweep_blocks;
            AdamSide rs, cs;
            rs.crec = p->r_crec; rs.own = t->R; rs.own_bias = t->br; rs.other = t->C; rs.other_bias = t->bc;
            rs.S1 = t->s1_R; rs.S1b = t->s1_br; rs.S2 = t->s2_R; rs.S2b = t->s2_br; rs.mark = p->r_mark; rs.own_twin = Vr; rs.other_twin = t->V;
            rs.n_host = p->host_counts[0]; rs.count_index = 0; rs.capP = rec_cap(p->chunk_cap);
            cs.crec = p->c_crec; cs.own = t->C; cs.own_bias = t->bc; cs.other = t->R; cs.other_bias = t->br;
            cs.S1 = t->s1_C; cs.S1b = t->s1_bc; cs.S2 = t->s2_C; cs.S2b = t->s2_bc; cs.mark = p->c_mark; cs.own_twin = t->V; cs.other_twin = Vr;
            cs.n_host = p->host_counts[2]; cs.count_index = 2; cs.capP = rs.capP;
            rs.cap_chunks = cs.cap_chunks = p->cap_chunks > 0 ? p->cap_chunks : 1;
            const float *prev = i > 0 ? recs + (size_t)(i - 1) * rec_floats : nullptr;
            float *mine = recs + (size_t)i * rec_floats;
#define CALL(LPR, NV)                                                                                                           \
            if (LPR * NV == d4)                                                                                                 \
                hipLaunchKernelGGL((tagged_adam_kernel<LPR, NV, true>), dim3(nb), dim3(kBlock), 0, st, p->counts, rs, cs, row_blocks, \
                                   chunk_blocks, (const float *)t->scalars, (const int64_t *)t->step, d4, h->inv_batch, (int)h->head, \
                                   h->neg_factor, kc, b1, b2, ln_b1, ln_b2, prev, prev_blocks, mine, i);                        \
            else                                                                                                                \
                hipLaunchKernelGGL((tagged_adam_kernel<LPR, NV, false>), dim3(nb), dim3(kBlock), 0, st, p->counts, rs, cs, row_blocks, \
                                   chunk_blocks, (const float *)t->scalars, (const int64_t *)t->step, d4, h->inv_batch, (int)h->head, \
                                   h->neg_factor, kc, b1, b2, ln_b1, ln_b2, prev, prev_blocks, mine, i)
            GLOVE_DISPATCH_PASS_SHAPE(shape, CALL);
#undef CALL
            prev_blocks = row_blocks;
        }
        hipLaunchKernelGGL(tagged_adam_flush_kernel, dim3(1), dim3(kBlock), 0, st, t->scalars, t->step,
                           (const float *)(recs + (size_t)(m - 1) * rec_floats), prev_blocks, m, kc, b1, b2, ln_b1, ln_b2,
                           i0 + m == n ? loss_out : nullptr);
    }
    return (int)hipGetLastError();
}

int glove_step_adam_f32(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws, size_t ws_bytes,
                        float *G_flat, float *loss_out, void *stream)
{
    if (adam_one_launch(p, t, h)) return launch_adam_chain(&p, 1, t, h, ws, ws_bytes, loss_out, stream);
    // a batch that touches a minority of the rows (the reference's 1,024 pairs): two launches, no gradient buffer
    // traffic; a batch that touches most rows: the dense form, whose sweep then wastes nothing
    if (int rc = plain_table(t, stream)) return rc;
    if (p && t && h && sides_of(h) == 3 && 2 * p->B <= (int64_t)v_row(t) + t->V)
        return step_adam_fused(p, t, h, ws, ws_bytes, G_flat, loss_out, stream);
    if (int rc = launch_passes(p, t, h, ws, ws_bytes, stream, 3)) return rc;
    if (int rc = launch_dense_grad(p, t, h, ws, ws_bytes, G_flat, stream)) return rc;
    return glove_dense_adam_f32(t, h, G_flat, loss_out, stream);
}

int glove_step_sparse_f32(const glove_plan *p, const glove_tables *t, const glove_hyper *h, void *ws, size_t ws_bytes,
                          float *G_flat, float *loss_out, void *stream)
{
    if (!h) return GLOVE_E_BADARG;
    if (h->optimizer == GLOVE_OPT_ADAGRAD) return glove_step_adagrad_f32(p, t, h, ws, ws_bytes, loss_out, stream);
    if (h->optimizer == GLOVE_OPT_ADAM) return glove_step_adam_f32(p, t, h, ws, ws_bytes, G_flat, loss_out, stream);
    if (int rc = plain_table(t, stream)) return rc;
    if (int rc = check_common(p, t, h, ws)) return rc;
    if (sides_of(h) != 3 || !t->s1_R || !t->s1_C || !t->s1_br || !t->s1_bc) return GLOVE_E_BADARG;
    if (h->optimizer == GLOVE_OPT_NADAM)        // passes (marking the batch's ids) + one kernel: apply for the ids, decay of m and v for all other rows
        return step_adam_fused(p, t, h, ws, ws_bytes, G_flat, loss_out, stream, true);
    hipStream_t st = (hipStream_t)stream;
    const StepConsts k = make_consts(t, h);
    if (h->optimizer == GLOVE_OPT_RMSPROP) {
        if (!G_flat || !(h->rho > 0.f && h->rho < 1.f)) return GLOVE_E_BADARG;
        if (int rc = launch_passes(p, t, h, ws, ws_bytes, stream, 3)) return rc;
        if (int rc = launch_dense_grad(p, t, h, ws, ws_bytes, G_flat, stream)) return rc;
        DenseSegs segs; float *tail; int nbx, sides;
        if (int rc = dense_common(t, h, G_flat, false, segs, tail, nbx, sides)) return rc;
        hipLaunchKernelGGL(dense_rmsprop_kernel, dim3(nbx, 4), dim3(kBlock), 0, st, segs, k, h->rho, t->scalars, tail, loss_out, 1);
        return (int)hipGetLastError();
    }
    const bool two_slots = h->optimizer == GLOVE_OPT_ADAMAX || h->optimizer == GLOVE_OPT_ADADELTA || h->optimizer == GLOVE_OPT_FTRL;
    if (h->optimizer != GLOVE_OPT_SGD && !two_slots) return GLOVE_E_BADARG;
    if (two_slots && (!t->s2_R || !t->s2_C || !t->s2_br || !t->s2_bc)) return GLOVE_E_BADARG;
    if (h->optimizer == GLOVE_OPT_ADAMAX && (!(h->beta1 > 0.0 && h->beta1 < 1.0) || !(h->beta2 > 0.0 && h->beta2 < 1.0))) return GLOVE_E_BADARG;
    if (h->optimizer == GLOVE_OPT_ADADELTA && !(h->rho > 0.f && h->rho < 1.f)) return GLOVE_E_BADARG;
    if (h->optimizer == GLOVE_OPT_FTRL && !(h->learning_rate > 0.f)) return GLOVE_E_BADARG;
    if (h->optimizer == GLOVE_OPT_SGD && !(h->momentum >= 0.f && h->momentum < 1.f)) return GLOVE_E_BADARG;
    if (int rc = launch_passes(p, t, h, ws, ws_bytes, stream, 3)) return rc;
    const StepWs w = carve_step_ws(ws, p->B, p->cap_chunks, t->d);
    const int d4 = t->d / 4;
    const RowShape shape = pick_row_shape(d4);
    IdWork wk = id_work(p);
    const int nb = wk.heavy_blocks + blocks_for(2 * (int64_t)p->cap_uniq, kBlock / shape.lpr) + 1;
    const int nb_row = rowpass_blocks(p, pass_shape(d4).lpr);
    const SideBufs rs = side_bufs(p, w, t, true), cs = side_bufs(p, w, t, false);
    const OptConsts o = {h->learning_rate, h->epsilon, h->momentum, 0.f, (float)h->beta1, (float)h->beta2, h->nesterov ? 1 : 0, h->rho};
    const double ln_b1 = h->optimizer == GLOVE_OPT_ADAMAX ? log((double)(float)h->beta1) : 0.0;
#define LAUNCH_OPT(LPR, NV, OPT)                                                                                           \
        hipLaunchKernelGGL((apply_sparse_opt_kernel<LPR, NV, OPT>), dim3(nb), dim3(kBlock), 0, st, wk, rs, cs, t->s2_R, t->s2_C, \
                           t->s2_br, t->s2_bc, d4, k, o, ln_b1, (const int64_t *)t->step, t->scalars, (const float *)w.blockpart, nb_row, loss_out)
#define CALL(LPR, NV)                                                                                                       \
    if (h->optimizer == GLOVE_OPT_SGD) LAUNCH_OPT(LPR, NV, GLOVE_OPT_SGD);                                                  \
    else if (h->optimizer == GLOVE_OPT_ADAMAX) LAUNCH_OPT(LPR, NV, GLOVE_OPT_ADAMAX);                                       \
    else if (h->optimizer == GLOVE_OPT_ADADELTA) LAUNCH_OPT(LPR, NV, GLOVE_OPT_ADADELTA);                                   \
    else LAUNCH_OPT(LPR, NV, GLOVE_OPT_FTRL)
    GLOVE_DISPATCH_ROW_SHAPE(shape, CALL);
#undef CALL
#undef LAUNCH_OPT
    return (int)hipGetLastError();
}

int glove_steps_adam_f32(const glove_plan *const *plans, int32_t n, const glove_tables *t, const glove_hyper *h,
                         void *ws, size_t ws_bytes, float *G_flat, float *loss_out, void *stream)
{
    if (!plans || n < 0) return GLOVE_E_BADARG;
    // runs of consecutive steps that take the one-launch form go out as chains
    for (int32_t i = 0; i < n;) {
        int32_t k = i;
        while (k < n && adam_one_launch(plans[k], t, h)) ++k;
        if (k > i) {
            if (int rc = launch_adam_chain(plans + i, k - i, t, h, ws, ws_bytes, k == n ? loss_out : nullptr, stream)) return rc;
            i = k;
            continue;
        }
        if (int rc = glove_step_adam_f32(plans[i], t, h, ws, ws_bytes, G_flat, i == n - 1 ? loss_out : nullptr, stream)) return rc;
        ++i;
    }
    return 0;
}


}  // extern "C"
