// Single-workgroup index build for small batches (B <= kSmallPlanMax).
//
// glove_plan_build's general path is ten dependent launches: launch-bound (~80 us) for the reference's default batch of
// 1,024 nonzeros, where one step of the kernels takes ~10 us.  A caller that hands over a fresh batch every step (the
// reference's input_fn does: data_utils.py:12-21) needs the index in a few microseconds, so batches up to 4,096 pairs
// are indexed by ONE workgroup entirely in LDS: two stable radix sorts by id (hand-written, the ballot-rank scheme of
// glove_plan.hip's tiled sort on one workgroup; passes of up to 8 bits: two for a 10^4-id vocabulary) and, per side,
// three scans over the threads that number the chunks and ids and close the id records (the first form — rocPRIM's block
// sort, six scans and a binary search per side — took 21 us at B = 1,024 and 60 at 4,096; a bitonic sort of 64-bit
// (id, position) keys before it 110 us at B = 4,096; this one 17 and 43 in the stamped diagnostic build).  The result is
// identical to the general path and to oracle/glove_ref.py:build_plan.
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

// Workgroup size: 16 waves.  A sort pass costs ~2 us whatever the number of ranking rounds per wave (its barriers and the
// column walk of the wave counters are the fixed part), so the rounds are spread over as many waves as a workgroup has:
// measured per step with the index rebuilt (text8 d = 64): B = 1,024: 16 waves x 1 pair per thread 31.0 us, 4 x 4 33.7;
// B = 4,096: 16 x 4 56.7 us, 4 x 16 85 (rocPRIM's block sort with six scans per side: 36 / 73).
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
template <int T, int E>
__device__ inline void block_sort_pairs(uint32_t (&key)[E], int32_t (&val)[E], int n, int bits, uint32_t *kbuf, int32_t *vbuf,
                                        SmallLds &L)
{
    constexpr int NW = T / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
        if ((int)threadIdx.x < nd) {
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) {
                const int c = L.wcnt[wv][threadIdx.x];
                L.wcnt[wv][threadIdx.x] = total;
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
        if ((int)threadIdx.x < nd) L.dig[threadIdx.x] = start;
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
    int32_t *chunk_id, *chunk_start, *uniq_slot, *uniq_rec;
};

// ids[k] (sorted ids, LDS) -> chunk / id arrays of one side.  Thread t owns positions t E .. t E + E - 1.  Position k opens an
// id where ids[k] != ids[k-1], and a chunk where it opens an id or lies a multiple of `cap` behind the start of its run.
// Three scans over the threads: the start of the run open at a thread's first position (max), the numbers of its first id
// and chunk (sum of both counts, packed), the next id opening behind its last position (min: where its last id's pairs
// end).
template <int T, int E>
__device__ inline void small_side(const int32_t *ids, int B, int cap, int heavy_chunks, int cap_heavy, int side,
                                  SmallLds &L, const SmallSideOut &o, int32_t *counts, int32_t *heavy)
{
    constexpr int NW = T / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k0 = threadIdx.x * E;
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
        if (uniq >> e & 1) o.uniq_slot[ui++] = ci;
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
        if (!(uniq >> e & 1)) continue;
        const int k = k0 + e;
        const int pairs = next_open - k, chunks = (pairs + cap - 1) / cap;
        reinterpret_cast<int4 *>(o.uniq_rec)[open_ui[e]] = make_int4(id[e + 1], open_ci[e], chunks, pairs);
        if (chunks > heavy_chunks) {
            const int slot = atomicAdd(counts + 4, 1);                      // zeroed at the kernel's start
            if (slot < cap_heavy) heavy[slot] = (side << 30) | open_ui[e];
        }
        next_open = k;
    }
    __syncthreads();                                                        // the scan slots are free again
}

template <int T, int E>
constexpr size_t small_lds_bytes() { return (size_t)6 * T * E * 4 + sizeof(SmallLds); }

template <int T, int E>
__global__ __launch_bounds__(T) void plan_small_kernel(
    const int32_t *__restrict__ row, const int32_t *__restrict__ col, const float *__restrict__ w,
    const float *__restrict__ y, int B, int V, int bits, glove_plan plan)
{
    constexpr int np = T * E, NW = T / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *kbuf = reinterpret_cast<uint32_t *>(smem);                 // [np] sorted ids of the sort in flight
    int32_t *vbuf = reinterpret_cast<int32_t *>(kbuf + np);              // [np] their positions before the sort
    int32_t *srow = vbuf + np;                                           // [np] row ids, row-sorted
    int32_t *rpos = srow + np;                                           // [np] row-sorted position of the pair that arrived i-th
    float *sw = reinterpret_cast<float *>(rpos + np);                    // [np] w, row-sorted
    float *sy = sw + np;                                                 // [np]
    SmallLds &L = *reinterpret_cast<SmallLds *>(sy + np);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every word of `counts` is written by this kernel: [4] (heavy ids) before anybody appends behind it, the rest at the end
    if (threadIdx.x == 0) plan.counts[4] = 0;
    const uint32_t Vr = (uint32_t)(plan.V_row > 0 ? plan.V_row : V);
    int mapped = 0;
    SMALL_STAMP(0);

    // ---- row side: stable sort by row id; ids outside [0, V_row) count as id 0 (the reference's unknown-token id,
    // estimator.py:26-28; see glove_plan.hip)
    uint32_t key[E];
    int32_t val[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int i = wave * 64 * E + j * 64 + lane;
        uint32_t r = 0;
        if (i < B) {
            r = (uint32_t)row[i];
            if (r >= Vr) { r = 0; ++mapped; }
        }
        key[j] = r;
        val[j] = i;
    }
    SMALL_STAMP(1);                                                      // row ids arrived
    block_sort_pairs<T, E>(key, val, B, bits, kbuf, vbuf, L);
    SMALL_STAMP(2);                                                      // row sort done
    // col / w / y pulled through the permutation: the row side's pair fields, kept in LDS for the col side
    {
        int32_t p[E], c[E];
        float wv[E], yv[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = threadIdx.x * E + e;
            p[e] = k < B ? vbuf[k] : 0;
            c[e] = col[p[e]];
            wv[e] = w[p[e]];
            yv[e] = y[p[e]];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = threadIdx.x * E + e;
            if (k >= B) continue;
            if ((uint32_t)c[e] >= (uint32_t)V) { c[e] = 0; ++mapped; }
            srow[k] = (int32_t)kbuf[k];
            rpos[p[e]] = k;
            sw[k] = wv[e]; sy[k] = yv[e];
            plan.r_partner[k] = c[e]; plan.r_w[k] = wv[e]; plan.r_y[k] = yv[e];
        }
    }
    __syncthreads();                                                     // (counts[4] = 0 is also ordered before the appends)
    SMALL_STAMP(3);                                                      // pair fields gathered, row arrays stored
    small_side<T, E>(srow, B, plan.chunk_cap, plan.heavy_chunks, plan.cap_heavy, 0, L,
                  SmallSideOut{plan.r_chunk_id, plan.r_chunk_start, plan.r_uniq_slot, plan.r_uniq_rec}, plan.counts, plan.heavy);

    SMALL_STAMP(4);                                                      // row side numbered and stored
    // ---- col side: stable sort of the batch AS IT ARRIVED by col id (like the row side; the tiled builder runs both sorts
    // in the same launches); a pair's row-sorted position links the sides
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int i = wave * 64 * E + j * 64 + lane;
        uint32_t c = i < B ? (uint32_t)col[i] : 0u;
        if (c >= (uint32_t)V) c = 0;                                     // (counted above, when the row side pulled it)
        key[j] = c;
        val[j] = i;
    }
    block_sort_pairs<T, E>(key, val, B, bits, kbuf, vbuf, L);
    SMALL_STAMP(5);                                                      // col sort done
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int j = threadIdx.x * E + e;
        if (j >= B) continue;
        const int p = rpos[vbuf[j]];
        if (plan.c_perm) { plan.c_perm[j] = p; plan.r_to_c[p] = j; }      // (optional links: see glove_hip.h)
        plan.c_partner[j] = srow[p];
        plan.c_w[j] = sw[p];
        plan.c_y[j] = sy[p];
    }
    SMALL_STAMP(6);                                                      // col arrays stored
    small_side<T, E>(reinterpret_cast<const int32_t *>(kbuf), B, plan.chunk_cap, plan.heavy_chunks, plan.cap_heavy, 1, L,
                  SmallSideOut{plan.c_chunk_id, plan.c_chunk_start, plan.c_uniq_slot, plan.c_uniq_rec}, plan.counts, plan.heavy);

    SMALL_STAMP(7);                                                      // col side numbered and stored
    // ---- ids mapped to 0, and the spare words
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) mapped += __shfl_xor(mapped, dlt, 64);
    if (lane == 0) L.s_mapped[wave] = mapped;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int wv = 0; wv < NW; ++wv) total += L.s_mapped[wv];
        plan.counts[5] = total;
        plan.counts[6] = plan.counts[7] = 0;
    }
    SMALL_STAMP(8);
}

template <int T, int E>
static int launch_small(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B, int32_t V,
                        const glove_plan *plan, hipStream_t st)
{
    const size_t smem = small_lds_bytes<T, E>();
    int bits = 1;                                                        // ids < 2^bits
    while (bits < 31 && (1u << bits) < (uint32_t)V) ++bits;
    // above the 64 KiB default of dynamic LDS: the limit is raised explicitly
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(plan_small_kernel<T, E>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((plan_small_kernel<T, E>), dim3(1), dim3(T), smem, st, row, col, w, y, (int)B, (int)V,
                       bits, *plan);
    return (int)hipGetLastError();
}

#ifdef GLOVE_STAMPS
extern "C" int glove_debug_set_small_stamps(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_small_stamps), &p, sizeof(p)); }
#endif

// host side: called from glove_plan_build for B <= kSmallPlanMax (4,096)
int plan_build_small(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t B, int32_t V,
                     const glove_plan *plan, hipStream_t st)
{
    if (B <= 1024) return launch_small<1024, 1>(row, col, w, y, B, V, plan, st);
    if (B <= 2048) return launch_small<1024, 2>(row, col, w, y, B, V, plan, st);
    return launch_small<1024, 4>(row, col, w, y, B, V, plan, st);
}

}  // namespace glove
