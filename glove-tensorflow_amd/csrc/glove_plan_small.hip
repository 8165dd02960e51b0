// Single-workgroup index build for small batches (B <= kSmallPlanMax).
//
// glove_plan_build's general path is a chain of dependent launches: launch-bound (~50 us) for the reference's default batch
// of 1,024 nonzeros, where one step of the kernels takes ~10 us.  A caller that hands over a fresh batch every step (the
// reference's input_fn does: data_utils.py:12-21) needs the index in a few microseconds, so batches up to 4,096 pairs
// are indexed by ONE workgroup entirely in LDS.  The two sides are independent (the col side sorts the batch in arrival
// order, like the row side): half of the workgroup's waves — a team — builds the row side while the other half builds the
// col side, in step through the same barriers: per team one stable radix sort by id (hand-written, the ballot-rank scheme
// of glove_plan.hip's tiled sort; passes of up to 8 bits: two for a 10^4-id vocabulary), the pair fields pulled through the
// permutation, and three scans over the team's threads that number the chunks and ids and close the id records.  (Earlier
// forms: rocPRIM's block sort, six scans and a binary search per side 21 us at B = 1,024 and 60 at 4,096; a bitonic sort
// of 64-bit (id, position) keys before it 110 us at B = 4,096; one side after the other on all 16 waves 17 and 43.)  The
// result is identical to the general path and to oracle/glove_ref.py:build_plan.
#include "glove_common.h"

namespace glove {

// ---- diagnostic build only (-DGLOVE_STAMPS): wall-clock stamps of this kernel's phases, per wave
#ifdef GLOVE_STAMPS
__device__ unsigned long long *g_small_stamps = nullptr;     // [waves][16], set by glove_debug_set_small_stamps
#define SMALL_STAMP(slot)                                                                                          \
    do {                                                                                                           \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                \
        if ((threadIdx.x & 63) == 0 && g_small_stamps) g_small_stamps[(threadIdx.x >> 6) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define SMALL_STAMP(slot) ((void)0)
#endif

// Workgroup size: 16 waves, 8 per side.  A sort pass costs ~2 us whatever the number of ranking rounds per wave (its
// barriers and the column walk of the wave counters are the fixed part), so the rounds are spread over as many waves as
// the workgroup has.
constexpr int kSmallMaxWaves = 16;
constexpr int kSmallDigits = 256;

struct SmallLds {
    int wcnt[kSmallMaxWaves][kSmallDigits];     // sort: per wave running digit counts, then the wave's offset inside the digit
    int dig[kSmallDigits];                   // sort: where the keys of a digit start
    int red[kSmallMaxWaves];                 // sort: wave totals of the digit scan
    int s_max[kSmallMaxWaves], s_sum[kSmallMaxWaves], s_min[kSmallMaxWaves];     // side numbering: wave results of its three scans
    int s_mapped[kSmallMaxWaves];
};

// ---- stable sort of n <= kSmallThreads x E (id, position) pairs by id, in LDS ------------------------------------------
// key / val: this thread's pairs, striped (wave w, round j, lane l holds position w 64 E + j 64 + l).  LSD radix,
// P = ceil(bits / 8) passes of ceil(bits / P)-bit digits.  A pass: the lanes of a round that hold the same digit find each
// other with `db` ballots, the group's first lane fetch-adds the wave's running count of that digit (LDS atomic with
// return) and hands the old value round: a key's stable rank inside its wave.  Thread d turns column d of the wave
// counters into the waves' starting offsets, a scan over the digits gives where each digit starts, every pair moves to
// its place in kbuf / vbuf, the next pass reads its positions from there.  Four barriers per pass.  On return kbuf / vbuf
// hold the pairs in sorted order (positions behind n: untouched).
// A team = the T threads (T / 64 waves) that build one side; `tid` / `wave` count inside the team.  Both teams run the same
// code in step: the workgroup barriers below are met by everybody.
struct Team { int tid, wave; };

template <int T, int E>
__device__ inline void block_sort_pairs(const Team &tm, uint32_t (&key)[E], int32_t (&val)[E], int n, int bits, uint32_t *kbuf,
                                        int32_t *vbuf, SmallLds &L)
{
    constexpr int NW = T / 64;
    const int lane = threadIdx.x & 63, wave = tm.wave, tid = tm.tid;
    const int P = (bits + 7) / 8, db = (bits + P - 1) / P, nd = 1 << db;
    for (int p = 0; p < P; ++p) {
        const int shift = p * db;
        int rank[E];
        if (p > 0) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const int i = wave * 64 * E + j * 64 + lane;
                key[j] = kbuf[i];
                val[j] = vbuf[i];
            }
        }
        for (int i = lane; i < nd; i += 64) L.wcnt[wave][i] = 0;          // a wave's own counters: LDS ops of one wave are ordered
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const bool valid = wave * 64 * E + j * 64 + lane < n;
            const int digit = (int)(key[j] >> shift) & (nd - 1);
            unsigned long long peers = __ballot(valid);
            for (int b = 0; b < db; ++b) {
                const bool bit = (digit >> b) & 1;
                const unsigned long long m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
            int before = 0;
            if (valid && lane == leader) before = atomicAdd(&L.wcnt[wave][digit], __popcll(peers));
            before = __shfl(before, leader, 64);
            rank[j] = before + __popcll(peers & ((1ull << lane) - 1ull));
        }
        __syncthreads();                                                    // (also: every read of kbuf / vbuf has happened)
        int total = 0;
        if (tid < nd) {
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) {
                const int c = L.wcnt[wv][tid];
                L.wcnt[wv][tid] = total;
                total += c;
            }
        }
        int incl = total;                                                   // exclusive scan over the digits (threads)
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
            const int o = __shfl_up(incl, dlt, 64);
            if (lane >= dlt) incl += o;
        }
        if (lane == 63) L.red[wave] = incl;
        __syncthreads();
        int start = incl - total;
        for (int wv = 0; wv < wave; ++wv) start += L.red[wv];
        if (tid < nd) L.dig[tid] = start;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if (wave * 64 * E + j * 64 + lane >= n) continue;
            const int digit = (int)(key[j] >> shift) & (nd - 1);
            const int dest = L.dig[digit] + L.wcnt[wave][digit] + rank[j];
            kbuf[dest] = key[j];
            vbuf[dest] = val[j];
        }
        __syncthreads();
    }
}

struct SmallSideOut {
    uint32_t *mark;             // bitmap of the side's ids (or nullptr)
    int32_t *chunk_id, *chunk_start, *uniq_slot, *uniq_rec;
    int32_t *crec;              // per-chunk records (or nullptr): words 2, 3 of every record header are written here
    int rec_dwords;             // int32 per record in memory
    uint32_t *chunk_hw;         // per chunk (or nullptr): word 3 on its own (glove_plan.r_chunk_hw)
};

// ids[k] (sorted ids, LDS) -> chunk / id arrays of one side.  Thread t owns positions t E .. t E + E - 1.  Position k opens an
// id where ids[k] != ids[k-1], and a chunk where it opens an id or lies a multiple of `cap` behind the start of its run.
// Three scans over the threads: the start of the run open at a thread's first position (max), the numbers of its first id
// and chunk (sum of both counts, packed), the next id opening behind its last position (min: where its last id's pairs
// end).
template <int T, int E>
__device__ inline void small_side(const Team &tm, const int32_t *ids, int B, int cap, int heavy_chunks, int cap_heavy, int side,
                                  SmallLds &L, const SmallSideOut &o, int32_t *counts, int32_t *heavy)
{
    constexpr int NW = T / 64;
    const int lane = threadIdx.x & 63, wave = tm.wave;
    const int k0 = tm.tid * E;
    int32_t id[E + 1];
    id[0] = (k0 > 0 && k0 <= B) ? ids[k0 - 1] : -1;
#pragma unroll
    for (int e = 0; e < E; ++e) id[e + 1] = k0 + e < B ? ids[k0 + e] : -1;
    unsigned uniq = 0;                                                      // bit e: position k0 + e opens an id
    int last_open = -1, first_open = INT32_MAX;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = k0 + e;
        if (k < B && (k == 0 || id[e + 1] != id[e])) {
            uniq |= 1u << e;
            last_open = k;
            first_open = first_open == INT32_MAX ? k : first_open;
        }
    }
    // ---- scan 1 (max): the run that is open when my first position begins
    int incl = last_open;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const int v = __shfl_up(incl, dlt, 64);
        if (lane >= dlt) incl = v > incl ? v : incl;
    }
    // ---- scan 3 (min, from the right): the first id opening behind my positions
    int sfx = first_open;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const int v = __shfl_down(sfx, dlt, 64);
        if (lane + dlt < 64) sfx = v < sfx ? v : sfx;
    }
    if (lane == 63) L.s_max[wave] = incl;
    if (lane == 0) L.s_min[wave] = sfx;
    __syncthreads();
    int run_start = __shfl_up(incl, 1, 64);
    if (lane == 0) run_start = -1;
    for (int wv = 0; wv < wave; ++wv) run_start = L.s_max[wv] > run_start ? L.s_max[wv] : run_start;
    int next_open = __shfl_down(sfx, 1, 64);
    if (lane == 63) next_open = INT32_MAX;
    for (int wv = wave + 1; wv < NW; ++wv) next_open = L.s_min[wv] < next_open ? L.s_min[wv] : next_open;
    if (next_open > B) next_open = B;                                       // the last id's pairs end with the batch
    // ---- my flags and counts
    unsigned chunk = 0;
    int nu = 0, nc = 0, rs = run_start;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = k0 + e;
        if (k < B) {
            if (uniq >> e & 1) rs = k;
            if ((uniq >> e & 1) || (k - rs) % cap == 0) { chunk |= 1u << e; ++nc; }
            nu += uniq >> e & 1;
        }
    }
    // ---- scan 2 (sum): numbers of my first id and chunk (both counts at most 4,096: packed into one word)
    const int packed = nu << 16 | nc;
    int ps = packed;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const int v = __shfl_up(ps, dlt, 64);
        if (lane >= dlt) ps += v;
    }
    if (lane == 63) L.s_sum[wave] = ps;
    __syncthreads();
    int before = ps - packed;
    for (int wv = 0; wv < wave; ++wv) before += L.s_sum[wv];
    int ui = before >> 16, ci = before & 0xffff;
    int open_ui[E], open_ci[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = k0 + e;
        open_ui[e] = ui;
        open_ci[e] = ci;
        if (k >= B) continue;
        if (uniq >> e & 1) {
            o.uniq_slot[ui++] = ci;
            if (o.mark) atomicOr(o.mark + (id[e + 1] >> 5), 1u << (id[e + 1] & 31));
        }
        if (chunk >> e & 1) { o.chunk_id[ci] = id[e + 1]; o.chunk_start[ci] = k; ++ci; }
        if (k == B - 1) {                                                   // closing entries and totals
            o.chunk_start[ci] = B;
            o.uniq_slot[ui] = ci;
            counts[2 * side] = ci;
            counts[2 * side + 1] = ui;
        }
    }
    // ---- {id, first chunk, chunks, pairs} per id: its pairs end where the next id opens; its chunks restart with it
#pragma unroll
    for (int e = E - 1; e >= 0; --e) {
        if (!(chunk >> e & 1)) continue;                                    // (a position that opens an id opens a chunk)
        const int k = k0 + e;
        const int pairs = next_open - k, chunks = (pairs + cap - 1) / cap;  // from k to the end of its id
        if (o.crec || o.chunk_hw) {
            // words 2, 3 of the chunk's record header {position of its id among the side's ids, first chunk of its id << 31 |
            // chunks of the id behind it}: at hand here, fill_records would bisect uniq_slot for them (ten dependent loads)
            const bool opens = (uniq >> e & 1) != 0;
            const uint32_t hw = (uint32_t)(chunks - 1) | (opens ? 0x80000000u : 0u);
            if (o.crec) *reinterpret_cast<int2 *>(o.crec + (size_t)open_ci[e] * o.rec_dwords + 2) = make_int2(opens ? open_ui[e] : open_ui[e] - 1, (int)hw);
            if (o.chunk_hw) o.chunk_hw[open_ci[e]] = hw;
        }
        if (!(uniq >> e & 1)) continue;
        reinterpret_cast<int4 *>(o.uniq_rec)[open_ui[e]] = make_int4(id[e + 1], open_ci[e], chunks, pairs);
        if (chunks > heavy_chunks) {
            const int slot = atomicAdd(counts + 4, 1);                      // zeroed at the kernel's start
            if (slot < cap_heavy) heavy[slot] = (side << 30) | open_ui[e];
        }
        next_open = k;
    }
    __syncthreads();                                                        // the scan slots are free again
}

// One side of the index, by the TT threads of `tm`'s team (team 0: row side, 1: col side).  kbuf / vbuf / L: the team's LDS;
// rpos: shared, written by the row side for plans that want the links, read by the col side behind a barrier.
template <int TT, int E>
__device__ inline void build_side(int team, const Team &tm, uint32_t *kbuf, int32_t *vbuf, int32_t *rpos, SmallLds &L,
                                  const int32_t *own, const int32_t *other, const float *sw, const float *sy,
                                  int B, int bits, const glove_plan &plan)
{
    const int lane = threadIdx.x & 63;
    // ---- stable sort of the batch by this side's id (own / other / sw / sy: the batch in LDS, ids already mapped into the tables)
    uint32_t key[E];
    int32_t val[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int i = tm.wave * 64 * E + j * 64 + lane;
        key[j] = i < B ? (uint32_t)own[i] : 0u;
        val[j] = i;
    }
    SMALL_STAMP(1);                                                      // ids arrived
    block_sort_pairs<TT, E>(tm, key, val, B, bits, kbuf, vbuf, L);
    SMALL_STAMP(2);                                                      // sort done
    // partner id / w / y pulled through the permutation: the side's pair fields
    {
        int32_t *partner_out = team == 0 ? plan.r_partner : plan.c_partner;
        float *w_out = team == 0 ? plan.r_w : plan.c_w, *y_out = team == 0 ? plan.r_y : plan.c_y;
        int32_t p[E], c[E];
        float wv[E], yv[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = tm.tid * E + e;
            p[e] = k < B ? vbuf[k] : 0;
            c[e] = other[p[e]];
            wv[e] = sw[p[e]];
            yv[e] = sy[p[e]];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = tm.tid * E + e;
            if (k >= B) continue;
            partner_out[k] = c[e]; w_out[k] = wv[e]; y_out[k] = yv[e];
            if (team == 0 && plan.c_perm) rpos[p[e]] = k;
        }
    }
    __syncthreads();                                                     // (counts[4] = 0 is also ordered before the appends)
    SMALL_STAMP(3);                                                      // pair fields gathered and stored
    if (team == 1 && plan.c_perm) {                                      // the optional links between the two orders (glove_hip.h)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int q = tm.tid * E + e;
            if (q >= B) continue;
            const int rp = rpos[vbuf[q]];
            plan.c_perm[q] = rp;
            plan.r_to_c[rp] = q;
        }
    }
    const int rd = 4 * rec_stride_q(rec_cap(plan.chunk_cap));
    const SmallSideOut so = team == 0 ? SmallSideOut{plan.r_mark, plan.r_chunk_id, plan.r_chunk_start, plan.r_uniq_slot, plan.r_uniq_rec, plan.r_crec, rd, plan.r_chunk_hw}
                                      : SmallSideOut{plan.c_mark, plan.c_chunk_id, plan.c_chunk_start, plan.c_uniq_slot, plan.c_uniq_rec, plan.c_crec, rd, plan.c_chunk_hw};
    small_side<TT, E>(tm, reinterpret_cast<const int32_t *>(kbuf), B, plan.chunk_cap, plan.heavy_chunks, plan.cap_heavy, team, L,
                      so, plan.counts, plan.heavy);
    SMALL_STAMP(4);                                                      // side numbered and stored
}

template <int T, int E, bool TEAMS>
constexpr size_t small_lds_bytes()
{
    return TEAMS ? (size_t)9 * (T / 2) * E * 4 + 2 * sizeof(SmallLds) : (size_t)7 * T * E * 4 + sizeof(SmallLds);
}

// T threads.  TEAMS: two teams of T / 2 build the two sides at the same time, E pairs per thread of a team (B <= (T / 2) E:
// up to 2,048 pairs — beyond, twice the ranking rounds per wave cost more than the second sort's fixed part saves: 4,096
// pairs 57 against 52 us); otherwise all T threads build one side after the other (B <= T E).
__device__ inline const glove_plan &pick_plan(const glove_plan &a) { return a; }
__device__ inline const glove_plan &pick_plan(const PlanSet &a) { return a.p[blockIdx.x]; }

template <int T, int E, bool TEAMS, class Plans>
__global__ __launch_bounds__(T) void plan_small_kernel(
    const int32_t *__restrict__ row, const int32_t *__restrict__ col, const float *__restrict__ w,
    const float *__restrict__ y, int B, int V, int bits, Plans set)
{
    // workgroup j indexes batch j of the stream into plan j (glove_plan_build_many; one plan — the bare struct: glove_plan_build)
    const glove_plan &plan = pick_plan(set);
    row += (size_t)blockIdx.x * B; col += (size_t)blockIdx.x * B; w += (size_t)blockIdx.x * B; y += (size_t)blockIdx.x * B;
    constexpr int TT = TEAMS ? T / 2 : T, np = TT * E, NT = TEAMS ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *kbuf = reinterpret_cast<uint32_t *>(smem);                 // [NT][np] sorted ids of a team's sort
    int32_t *vbuf = reinterpret_cast<int32_t *>(kbuf + NT * np);         // [NT][np] their arrival indices
    int32_t *rpos = vbuf + NT * np;                                      // [np] row-sorted position by arrival index (plans with links)
    int32_t *srow = rpos + np, *scol = srow + np;                        // [np] the batch: ids mapped into the tables,
    float *sw = reinterpret_cast<float *>(scol + np), *sy = sw + np;     // [np] weights and values, by arrival index
    SmallLds *L = reinterpret_cast<SmallLds *>(sy + np);                 // [NT]
    // every word of `counts` is written by this kernel: [4] (heavy ids) before anybody appends behind it, the rest at the end
    if (threadIdx.x == 0) plan.counts[4] = 0;
    if (plan.r_mark) {                                                   // the id bitmaps start at zero (small_side sets the bits behind the barriers below)
        const int wr = ((plan.V_row > 0 ? plan.V_row : V) + 31) / 32, wc = (V + 31) / 32;
        for (int i = threadIdx.x; i < wr; i += T) plan.r_mark[i] = 0u;
        for (int i = threadIdx.x; i < wc; i += T) plan.c_mark[i] = 0u;
    }
    SMALL_STAMP(0);
    // ---- the batch into LDS (coalesced; both sides sort and gather from there); ids outside their table count as id 0 (the
    // reference's unknown-token id, estimator.py:26-28; see glove_plan.hip)
    int mapped = 0;
    {
        const uint32_t Vr = (uint32_t)(plan.V_row > 0 ? plan.V_row : V);
        for (int i = threadIdx.x; i < B; i += T) {
            uint32_t r = (uint32_t)row[i], c = (uint32_t)col[i];
            if (r >= Vr) { r = 0; ++mapped; }
            if (c >= (uint32_t)V) { c = 0; ++mapped; }
            srow[i] = (int32_t)r; scol[i] = (int32_t)c; sw[i] = w[i]; sy[i] = y[i];
        }
    }
    __syncthreads();
    if (TEAMS) {
        const int team = threadIdx.x / TT;                               // 0: row side, 1: col side (wave-uniform)
        const Team tm = {(int)threadIdx.x % TT, ((int)threadIdx.x % TT) >> 6};
        build_side<TT, E>(team, tm, kbuf + team * np, vbuf + team * np, rpos, L[team], team == 0 ? srow : scol,
                          team == 0 ? scol : srow, sw, sy, B, bits, plan);
    } else {
        const Team tm = {(int)threadIdx.x, (int)threadIdx.x >> 6};
        for (int side = 0; side < 2; ++side)
            build_side<TT, E>(side, tm, kbuf, vbuf, rpos, L[0], side == 0 ? srow : scol, side == 0 ? scol : srow, sw, sy, B, bits, plan);
    }
    // ---- ids mapped to 0, and the spare words
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) mapped += __shfl_xor(mapped, dlt, 64);
    __shared__ int s_mapped[kSmallMaxWaves];
    if (lane == 0) s_mapped[wave] = mapped;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int wv = 0; wv < T / 64; ++wv) total += s_mapped[wv];
        plan.counts[5] = total;
        plan.counts[6] = plan.counts[7] = 0;
    }
    SMALL_STAMP(5);
}

template <int T, int E, bool TEAMS>
static int launch_small(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B, int32_t V,
                        const PlanSet &set, int n, hipStream_t st)
{
    const size_t smem = small_lds_bytes<T, E, TEAMS>();
    int bits = 1;                                                        // ids < 2^bits
    while (bits < 31 && (1u << bits) < (uint32_t)V) ++bits;
    // above the 64 KiB default of dynamic LDS: the limit is raised explicitly
    if (n == 1) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(plan_small_kernel<T, E, TEAMS, glove_plan>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((plan_small_kernel<T, E, TEAMS, glove_plan>), dim3(1), dim3(T), smem, st, row, col, w, y, (int)B, (int)V,
                           bits, set.p[0]);
        return (int)hipGetLastError();
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(plan_small_kernel<T, E, TEAMS, PlanSet>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((plan_small_kernel<T, E, TEAMS, PlanSet>), dim3(n), dim3(T), smem, st, row, col, w, y, (int)B, (int)V,
                       bits, set);
    return (int)hipGetLastError();
}

#ifdef GLOVE_STAMPS
extern "C" int glove_debug_set_small_stamps(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_small_stamps), &p, sizeof(p)); }
#endif

// host side: called from glove_plan_build for B <= kSmallPlanMax (4,096)
int plan_build_small(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B, int32_t V,
                     const PlanSet &set, int n, hipStream_t st)
{
    // 16 waves: 8 per side up to 2,048 pairs, all of them on one side after the other beyond
    if (B <= 1024) return launch_small<1024, 2, true>(row, col, w, y, B, V, set, n, st);
    if (B <= 2048) return launch_small<1024, 4, true>(row, col, w, y, B, V, set, n, st);
    return launch_small<1024, 4, false>(row, col, w, y, B, V, set, n, st);
}

}  // namespace glove
