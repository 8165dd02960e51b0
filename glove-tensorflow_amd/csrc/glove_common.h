// Shared device helpers for the gfx950 GloVe kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/glove_hip.h"

namespace glove {

using f4 = float __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;        // 4 waves per workgroup
constexpr int kMaxBlocks = 2048;   // 256 CUs x 8 workgroups: grid-stride beyond that
// Workgroups of a fused (run-merged) pass, whose lane groups take `per` consecutive chunks each instead of striding:
// enough of them that `per` (glove_step.hip fuse_per, where the measurements are) stays at 12 up to 786 k chunks a side.
constexpr int kMaxPassBlocks = 8192;
constexpr int kPartials = 4;       // per-block loss partials: sum w diff^2, sum |r|^2+|c|^2, sum b^2, sum e

// ---- diagnostic build only (-DGLOVE_STAMPS): per-wave wall-clock stamps (s_memrealtime, 100 MHz)
#ifdef GLOVE_STAMPS
extern __device__ unsigned long long *g_stamps;     // [waves][8], set by glove_debug_set_stamps
__device__ inline void stamp(int slot)
{
    if ((threadIdx.x & 63) == 0 && g_stamps)
        g_stamps[((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 8 + slot] = __builtin_amdgcn_s_memrealtime();
}
#define GLOVE_STAMP(slot) stamp(slot)
#define GLOVE_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define GLOVE_STAMP(slot) ((void)0)
#define GLOVE_DRAIN() ((void)0)
#endif

__host__ __device__ inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Small device buffers are zeroed by a KERNEL, never by hipMemsetAsync: captured into a hipGraph, a memset node of a few
// words came back wrong on replay (ROCm 7.2: counts[4..7] of a plan held host-pointer-like garbage after the second
// replay, tools/dbg_dyn_graph.py) — the cause of the GPU memory faults of the replayed dynamic index build.
static __global__ void zero_words_kernel(uint32_t *p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
inline hipError_t zero_words(void *p, int n_words, hipStream_t st)
{
    hipLaunchKernelGGL(zero_words_kernel, dim3((n_words + 63) / 64), dim3(64), 0, st, (uint32_t *)p, n_words);
    return hipGetLastError();
}

// Up to kPlanSetMax plans handed to one launch by value (the one-workgroup index build of consecutive small batches)
constexpr int kPlanSetMax = 8;
struct PlanSet { glove_plan p[kPlanSetMax]; };

// ---- per-chunk records (glove_plan.r_crec / c_crec) --------------------------------------------------
// A record holds 4 + 3 * capP dwords, capP = rec_cap(chunk_cap): the header, then capP / kRecPad blocks of kRecPad pairs,
// each block {partner[8] | w[8] | y[8]} — so that what a chunk of n pairs needs is the PREFIX of 4 + 24 ceil(n / 8)
// dwords (a reader that knows n, or finds it in the header after a first 112-byte read, fetches no padding).
// In memory a record starts on a 128-byte line of its own: header + block 0 + 16 bytes of padding fill line 0, the other
// blocks follow packed, the stride is rounded up to whole lines (rec_stride_q).  Most chunks of a big batch hold a few pairs:
// their reader touches ONE line (1.75 on average when records were 400 bytes apart), and their writer stores a WHOLE line
// (fill_records; partial-line stores made the memory side fetch every line first: 80 us for 2 x 373 k chunks at V = 400 k,
// B = 1 M).  In LDS the pass kernels keep the packed image: logical float4 f of a record sits at float4 rec_gq(f) in memory.
// INVARIANT: a trip of the pass kernel reads U <= kRecPad consecutive pair slots of each field starting at a multiple
// of U below the chunk's pair count, i.e. inside one block; the slots behind the chunk's pairs in the chunk's last
// block are VALID partner ids with weight 0 (fill_records replays pair 0 there).  A record sized for the
// bare chunk_cap (or rounded to 4 only) lets the 8-slot trips of the d <= 32 shapes read the head of the w field as
// partner ids — float bit patterns used as row numbers, a load far outside the table: the GPU fault behind the
// aborted run of (B = 3000, V = 40, d = 16, chunk_cap = 2) while the records were being written.
constexpr int kRecPad = 8;
__host__ __device__ inline int rec_cap(int chunk_cap) { return (chunk_cap + kRecPad - 1) / kRecPad * kRecPad; }
// dword of pair q's partner id inside a record (its weight: + kRecPad, its value: + 2 kRecPad)
__host__ __device__ inline int rec_pair(int q) { return 4 + 3 * kRecPad * (q / kRecPad) + q % kRecPad; }
// float4 per record in memory (a multiple of 8 = whole 128-byte lines) / where logical float4 f of a record lies
__host__ __device__ inline int rec_stride_q(int capP) { return (8 + 6 * (capP / kRecPad - 1) + 7) / 8 * 8; }
__host__ __device__ inline int rec_gq(int f) { return f < 7 ? f : f + 1; }

// ---- cross-lane sums ------------------------------------------------------------------
// Butterfly all-reduce inside an aligned group of LPR lanes.  Strides 1,2 use quad_perm DPP;
// 4 and 8 use row_half_mirror / row_mirror (valid as butterfly steps because the lanes of
// each already-reduced sub-group hold the same value); 16 and 32 cross DPP rows.
template <int CTRL>
__device__ inline float dpp_add(float v)
{
    int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
    return v + __int_as_float(x);
}

template <int LPR>
__device__ inline float group_sum(float v)
{
    static_assert(LPR == 8 || LPR == 16 || LPR == 32 || LPR == 64, "group width");
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    if (LPR >= 16) v = dpp_add<0x140>(v);   // row_mirror
    if (LPR >= 32) v += __shfl_xor(v, 16, 64);
    if (LPR >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

__device__ inline float wave_sum(float v) { return group_sum<64>(v); }

// ---- keyed bijection of [0, n): a Feistel network over the b = ceil(log2 n) bits of the position — halves of floor(b / 2)
// (high) and ceil(b / 2) (low) bits, four rounds of a multiply-xorshift mix that alternately rewrite the high half from the low
// one and the low half from the high one, one 32-bit round key each — with cycle walking: positions that land outside [0, n)
// go round again (2^b < 2 n: fewer than two rounds of it on average; a balanced network over an even number of bits would
// waste up to four).  What an epoch's reshuffle of a resident nonzero stream is drawn from (the deal of glove_epoch.hip,
// glove_epoch_deal): no sort of n random keys, no index array.  oracle/glove_ref.py:feistel_walk restates it.
__host__ __device__ inline uint32_t feistel_mix(uint32_t x, uint32_t k)
{
    // one multiply per round: the deal's histogram pass is bound by this arithmetic (25 M pairs x 2 orders x 4 rounds x ~1.3
    // walks: 290 us per epoch with a two-multiply finaliser per round)
    // (mixing checked on the CPU restatement: |corr(position, seat)| < 0.01, neighbours share a batch as often as chance)
    x += k;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15;
    return x;
}

// (32-bit arithmetic: glove_epoch_deal takes n < 2^31, so bits <= 31 and both halves fit 16 bits — the 64-bit form cost the
// deal's histogram pass twice the integer instructions per round)
// one trip through the network (a bijection of [0, 2^bits))
__host__ __device__ inline uint32_t feistel_once(uint32_t x, int bits, uint4 key)
{
    const int hb = bits / 2, lb = bits - hb;
    const uint32_t hmask = (1u << hb) - 1u, lmask = (1u << lb) - 1u;
    uint32_t H = x >> lb, L = x & lmask;
    H ^= feistel_mix(L, key.x) & hmask;
    L ^= feistel_mix(H, key.y) & lmask;
    H ^= feistel_mix(L, key.z) & hmask;
    L ^= feistel_mix(H, key.w) & lmask;
    return (H << lb) | L;
}

__host__ __device__ inline uint32_t feistel_walk(uint32_t x, uint32_t n, int bits, uint4 key)
{
    do x = feistel_once(x, bits, key); while (x >= n);
    return x;
}

// b with 2^b >= n (b >= 2), or -1 when n is beyond 2^62
inline int feistel_bits(int64_t n)
{
    int b = 2;
    while (b < 62 && (1ull << b) < (uint64_t)n) ++b;
    return (1ull << b) < (uint64_t)n ? -1 : b;
}

__device__ inline int wave_sum_int(int v)
{
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) v += __shfl_xor(v, dlt, 64);
    return v;
}

// lanes of the wave that hold the same digit as this one (valid lanes only); db ballots
__device__ inline unsigned long long digit_peers(int digit, int db, bool valid)
{
    unsigned long long peers = __ballot(valid);
    for (int b = 0; b < db; ++b) {
        const bool bit = (digit >> b) & 1;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// The horizontal sum of four products, association and fusing spelled out: left to the compiler one build makes it a chain of
// fmas and another (the passes compiled for one loss head, whose body is a single basic block) packs the products two to a
// v_pk_mul_f32 and adds them unfused — a last-bit difference in the dot product between kernels that are asked for the same bits
// (ids one chunk holds: every step form, the packing passes, the one-launch forms).
__device__ inline float dot4(const f4 a, const f4 b)
{
    return __builtin_fmaf(a.w, b.w, __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)));
}

// Per-workgroup reduction of kPartials running sums into out[blockIdx.x][kPartials].
// Fixed order (lane butterfly, then waves 0..3) => bitwise repeatable for a fixed grid.
__device__ inline void block_partials_store(float (&acc)[kPartials], float *out)
{
    __shared__ float red[kBlock / 64][kPartials];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kPartials; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < kPartials; ++i) red[wave][i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < kPartials) {
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < kBlock / 64; ++wv) s += red[wv][threadIdx.x];
        out[(size_t)blockIdx.x * kPartials + threadIdx.x] = s;
    }
}

// ---- step workspace layout ----------------------------------------------------------------
struct StepWs {
    float *e;        // [B]            per-pair error coefficient e_i, row-sorted order (diagnostics / eval)
    float *gp_r;     // [cap_chunks*d] row-side chunk partial gradient rows
    float *gp_c;     // [cap_chunks*d] col-side
    float *gb_r;     // [cap_chunks]   row-side chunk partial bias gradients (sum e)
    float *gb_c;     // [cap_chunks]
    float *blockpart;  // [kMaxPassBlocks*kPartials]
    int32_t *work;     // [4 + 2*cap_chunks] fused step forms: count + positions of the ids the apply launch still has to do
    size_t bytes;
};

inline StepWs carve_step_ws(void *ws, int64_t B, int32_t cap_chunks, int32_t d)
{
    StepWs s;
    size_t off = 0;
    char *base = (char *)ws;
    auto take = [&](size_t nfloats) {
        float *p = (float *)(base + off);
        off += align_up(nfloats * sizeof(float), 256);
        return p;
    };
    s.e = take((size_t)B);
    s.gp_r = take((size_t)cap_chunks * d);
    s.gp_c = take((size_t)cap_chunks * d);
    s.gb_r = take((size_t)cap_chunks);
    s.gb_c = take((size_t)cap_chunks);
    s.blockpart = take((size_t)kMaxPassBlocks * kPartials);
    s.work = (int32_t *)take((size_t)4 + 2 * (size_t)cap_chunks);
    s.bytes = off;
    return s;
}

inline int blocks_for(int64_t items, int per_block)
{
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}

// The most chunks a side can have: the count itself for a resident plan; for a plan refilled on the device (its counts
// are not read back) one chunk per distinct id plus one per full chunk_cap pairs — far fewer than the B the arrays are
// sized for, and `per` below must not grow with the capacity: the ids of the Zipf head sit in consecutive chunks, so a
// group's `per` full chunks are the launch's critical path (V = 400 k, B = 1 M: per 62 instead of 23 ran a pass in 800
// instead of 290 us).
inline int64_t most_chunks(const glove_plan *p, bool row)
{
    const int32_t known = p->host_counts[row ? 0 : 2];
    if (known >= 0) return known;
    const int64_t bound = (int64_t)p->cap_uniq + p->B / (p->chunk_cap > 0 ? p->chunk_cap : 1) + 1;
    return bound < p->cap_chunks ? bound : p->cap_chunks;
}
// (lanes per row, float4 per lane) covering d4 = d/4 float4 per embedding row with the least
// idle lanes; a wave64 holds 64/LPR rows at a time.
struct RowShape { int lpr, nv; };
inline RowShape pick_row_shape(int d4)
{
    const RowShape cands[] = {{16, 1}, {32, 1}, {16, 2}, {64, 1}, {32, 2}, {32, 3}, {64, 2}, {64, 3}, {64, 4}};
    RowShape best = {0, 0};
    int best_slots = 1 << 30;
    for (const RowShape &c : cands) {
        const int slots = c.lpr * c.nv;
        if (slots >= d4 && slots < best_slots) { best = c; best_slots = slots; }
    }
    return best;
}

// Shapes of the gather passes (rowpass / colpass): narrow groups (8 lanes = one 128-B line per
// load at d = 64) put 8 chunks on a wave, which divides the per-pair bookkeeping instructions.
inline RowShape pick_pass_shape(int d4)
{
    if (d4 <= 8) return RowShape{8, 1};
    if (d4 <= 16) return RowShape{8, 2};      // d = 64: measured faster than 16 x 1 (A/B, same process)
    return pick_row_shape(d4);                // wider rows: 16-lane x 5 measured slower than 32 x 3 at d = 300
}

#define GLOVE_DISPATCH_PASS_SHAPE(shape, CALL)                                  \
    do {                                                                        \
        const int key_ = (shape).lpr * 8 + (shape).nv;                          \
        switch (key_) {                                                         \
        case 8 * 8 + 1: { CALL(8, 1); } break;                                  \
        case 8 * 8 + 2: { CALL(8, 2); } break;                                  \
        case 16 * 8 + 2: { CALL(16, 2); } break;                                \
        case 32 * 8 + 1: { CALL(32, 1); } break;                                \
        case 32 * 8 + 2: { CALL(32, 2); } break;                                \
        case 32 * 8 + 3: { CALL(32, 3); } break;                                \
        case 64 * 8 + 1: { CALL(64, 1); } break;                                \
        case 64 * 8 + 2: { CALL(64, 2); } break;                                \
        case 64 * 8 + 3: { CALL(64, 3); } break;                                \
        case 64 * 8 + 4: { CALL(64, 4); } break;                                \
        default: return GLOVE_E_BADARG;                                         \
        }                                                                       \
    } while (0)

#define GLOVE_DISPATCH_ROW_SHAPE(shape, CALL)                                   \
    do {                                                                        \
        const int key_ = (shape).lpr * 8 + (shape).nv;                          \
        switch (key_) {                                                         \
        case 16 * 8 + 1: { CALL(16, 1); } break;                                \
        case 16 * 8 + 2: { CALL(16, 2); } break;                                \
        case 32 * 8 + 1: { CALL(32, 1); } break;                                \
        case 32 * 8 + 2: { CALL(32, 2); } break;                                \
        case 32 * 8 + 3: { CALL(32, 3); } break;                                \
        case 64 * 8 + 1: { CALL(64, 1); } break;                                \
        case 64 * 8 + 2: { CALL(64, 2); } break;                                \
        case 64 * 8 + 3: { CALL(64, 3); } break;                                \
        case 64 * 8 + 4: { CALL(64, 4); } break;                                \
        default: return GLOVE_E_BADARG;                                         \
        }                                                                       \
    } while (0)

}  // namespace glove
