// Epochs of a resident nonzero stream.
//
// The reference's input_fn reshuffles the interaction file every epoch (make_csv_dataset(shuffle=True, num_epochs=None),
// reference src/models/data_utils.py:12-21), so a batch's composition is new every epoch and Keras' OptimizerV2 runs
// Unique + UnsortedSegmentSum on every step's gradients.  Sorting every batch by row id and by col id when
// it is used (glove_plan_build) costs two stable sorts per step.  Here the sorting happens ONCE, at load:
//
//   masters   the rank's nonzeros kept in two orders — row-major, sorted by (row id, col id, stream index), and col-major,
//             sorted by (col id, row id, stream index) — plus link[q] = row-major position of the pair at col-major position q
//             (glove_masters_build);
//   deal      per epoch a keyed bijection seats every pair: the pair at row-major position p sits at seat(p) in [0, n) and
//             belongs to batch seat(p) / B.  ONE stable counting-sort pass per order by batch number (two for more than 2,048
//             batches) writes the epoch: batch k of either order occupies positions [k B, (k + 1) B) and is already sorted by
//             its id (glove_epoch_deal).  Both orders hold the same pairs per batch because both take the batch number
//             from the row-major position.  The batch "arrives" in row-major order: the col side's (col id, row id, stream
//             index) order is the stable sort of that arrival order by col id, so oracle/glove_ref.py:build_plan on the
//             row side's pairs gives the same index bit for bit;
//   index     glove_plan_build_sorted (glove_plan.hip) numbers the chunks and ids of any run of consecutive batches in three
//             launches, no sort.
//
// Integer work, bound by HBM: a deal reads and writes 16 B per pair and order (+ 4 B of link).
#include "glove_common.h"

#include <type_traits>

namespace glove {

// ---- one stable counting-sort pass over n items, generic in what an item is ---------------------------------------------
// Tiles of kCsThreads x E consecutive positions, one workgroup each, grid.y = which of up to two independent jobs ("sides").
//   csort_hist     count[side][digit][tile]
//   csort_scan     offs[side][digit][tile] = items of that digit in earlier tiles (a workgroup per digit scans its row);
//                  tot[side][digit]
//   csort_scatter  destination = items of smaller digits + offs + stable rank inside the tile (wave ballots, as the
//                  sort passes of glove_plan.hip).  The tile is first sorted INSIDE LDS (every item to its tile-local
//                  position), then written out position by position: consecutive lanes hold consecutive items of one digit,
//                  i.e. consecutive destinations — coalesced stores whatever the number of digits (storing straight from the
//                  arrival order, a wave's 64 items spread over 24 batches, ran a 25 M-pair deal at 1 TB/s).
// Every word of count / offs / tot that is read has been written by the same pass: nothing relies on zeroed memory.
constexpr int kCsThreads = 256;
constexpr int kCsWaves = kCsThreads / 64;
constexpr int kCsMaxDigits = 2048;              // 11-bit digits
constexpr int kCsMaxBits = 11;

struct CsGeom {
    int64_t n;
    int ntiles;
    int nd[2], db[2];                           // digits per side and their bits
    int stride;                                 // rows per side of the tables (max of nd)
    uint32_t *count, *offs;                     // [2][stride][ntiles]
    uint32_t *tot;                              // [2][stride]
};

template <int E, class Job>
__global__ __launch_bounds__(kCsThreads) void csort_hist(Job job, CsGeom g)
{
    __shared__ int hist[kCsMaxDigits];
    const int side = blockIdx.y, nd = g.nd[side], db = g.db[side];
    for (int i = threadIdx.x; i < nd; i += kCsThreads) hist[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = ((int64_t)blockIdx.x * kCsWaves + wave) * (64 * E);
    // A job whose digits come from seats of the keyed bijection (the deal's first pass): a lane walks ITS E positions one behind
    // the other — a trip through the network per iteration, the next position taken up as soon as a seat lands inside [0, n) —
    // instead of the wave walking position j until its slowest lane has landed: E x 1.34 trips per lane on average (the slowest
    // of 64 lanes about 15 at E = 8, n = 25 M in a domain of 2^25) against E x the slowest of 64 walks (about 4 each: 31).  Same
    // seats.  Positions and seats wait in LDS (a register array indexed by a per-lane counter would live in scratch).
    __shared__ uint32_t walk[Job::kSeats ? E * kCsThreads : 1];
    if (Job::kSeats) {
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int64_t i = base + j * 64 + lane;
            if (i < g.n) { walk[j * kCsThreads + threadIdx.x] = job.position(side, i); cnt = j + 1; }
        }
        int j = 0;
        uint32_t x = cnt > 0 ? walk[threadIdx.x] : 0u;      // (same lane wrote it)
        while (j < cnt) {
            x = job.seat_once(x);
            if (job.seated(x)) {
                walk[j * kCsThreads + threadIdx.x] = x;
                ++j;
                if (j < cnt) x = walk[j * kCsThreads + threadIdx.x];
            }
        }
    }
#pragma unroll 4
    for (int j = 0; j < E; ++j) {
        const int64_t i = base + j * 64 + lane;
        const bool valid = i < g.n;
        int digit = 0;
        if (Job::kSeats) { if (valid) digit = job.digit_of_seat(side, i, walk[j * kCsThreads + threadIdx.x]); }
        else if (valid) digit = job.digit(side, i);
        const unsigned long long peers = digit_peers(digit, db, valid);
        if (valid && lane == __ffsll((long long)peers) - 1) atomicAdd(&hist[digit], __popcll(peers));
    }
    __syncthreads();
    uint32_t *out = g.count + (size_t)side * g.stride * g.ntiles + blockIdx.x;
    for (int i = threadIdx.x; i < nd; i += kCsThreads) out[(size_t)i * g.ntiles] = (uint32_t)hist[i];
}

// exclusive scan of one value per thread over the workgroup; *total (LDS) = the sum.  Called by all kCsThreads threads.
__device__ inline int block_excl_scan(int v, int *wtot /* [kCsWaves] LDS */, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const int o = __shfl_up(incl, dlt, 64);
        if (lane >= dlt) incl += o;
    }
    __syncthreads();                                    // (wtot may still be read from a previous call)
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int start = incl - v;
    total = 0;
#pragma unroll
    for (int wv = 0; wv < kCsWaves; ++wv) {
        if (wv < wave) start += wtot[wv];
        total += wtot[wv];
    }
    return start;
}

// one workgroup per (digit, side): the row of that digit's tile counts -> exclusive prefix, 8 tiles per thread and round
__global__ __launch_bounds__(kCsThreads) void csort_scan(CsGeom g)
{
    __shared__ int wtot[kCsWaves];
    const int side = blockIdx.y, d = blockIdx.x;
    if (d >= g.nd[side]) return;
    const uint32_t *row = g.count + ((size_t)side * g.stride + d) * g.ntiles;
    uint32_t *out = g.offs + ((size_t)side * g.stride + d) * g.ntiles;
    constexpr int kPer = 8;
    int carry = 0;
    for (int base = 0; base < g.ntiles; base += kCsThreads * kPer) {
        int v[kPer], s = 0;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int t = base + threadIdx.x * kPer + k;
            v[k] = t < g.ntiles ? (int)row[t] : 0;
            s += v[k];
        }
        int total;
        int run = carry + block_excl_scan(s, wtot, total);
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int t = base + threadIdx.x * kPer + k;
            if (t < g.ntiles) out[t] = (uint32_t)run;
            run += v[k];
        }
        carry += total;
    }
    if (threadIdx.x == 0) g.tot[(size_t)side * g.stride + d] = (uint32_t)carry;
}

template <int E, class Job>
__global__ __launch_bounds__(kCsThreads) void csort_scatter(Job job, CsGeom g)
{
    using Item = typename Job::Item;
    constexpr int T = kCsThreads * E;
    extern __shared__ __attribute__((aligned(16))) unsigned char cs_smem[];
    const int side = blockIdx.y, nd = g.nd[side], db = g.db[side];
    Item *stage = reinterpret_cast<Item *>(cs_smem);                    // [T] the tile in its sorted order
    uint16_t *sdig = reinterpret_cast<uint16_t *>(stage + T);           // [T] the digit of the item at each tile-local position
    int *wcnt = reinterpret_cast<int *>(sdig + T);                      // [kCsWaves][nd] a wave's running digit counts; then where its items of a digit start
    int *gbase = wcnt + kCsWaves * nd;                                  // [nd] destination of tile-local position 0 if it held this digit
    int *tstart = gbase + nd;                                           // [nd] tile-local start of the digit
    __shared__ int wtot[kCsWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile0 = (int64_t)blockIdx.x * T;
    const int64_t base = tile0 + (int64_t)wave * (64 * E);
    Item it[E];
    int digit[E], rank[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int64_t i = base + j * 64 + lane;
        digit[j] = 0;
        if (i < g.n) digit[j] = job.load(side, i, it[j]);
    }
    // ---- stable rank of every item among the items of its digit in this wave's range (a wave zeroes and uses its own
    // counters: LDS operations of one wave complete in order)
    int *mine = wcnt + wave * nd;
    for (int i = lane; i < nd; i += 64) mine[i] = 0;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const bool valid = base + j * 64 + lane < g.n;
        const unsigned long long peers = digit_peers(digit[j], db, valid);
        const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
        int before = 0;
        if (valid && lane == leader) before = atomicAdd(&mine[digit[j]], __popcll(peers));
        before = __shfl(before, leader, 64);
        rank[j] = before + __popcll(peers & ((1ull << lane) - 1ull));
    }
    __syncthreads();                                    // every wave's running counts are final
    // ---- per digit (thread t: digits t kPer .. t kPer + kPer - 1, rounds of kCsThreads kPer digits): where the waves' items
    // start inside the digit, where the digit starts inside the tile, where it starts in the output
    constexpr int kPer = kCsMaxDigits / kCsThreads;
    {
        const uint32_t *tot = g.tot + (size_t)side * g.stride;
        const uint32_t *offs = g.offs + (size_t)side * g.stride * g.ntiles + blockIdx.x;
        int tile_n[kPer], all_n[kPer], st = 0, sa = 0;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int d = threadIdx.x * kPer + k;
            tile_n[k] = all_n[k] = 0;
            if (d < nd) {
                int run = 0;
#pragma unroll
                for (int wv = 0; wv < kCsWaves; ++wv) {
                    const int c = wcnt[wv * nd + d];
                    wcnt[wv * nd + d] = run;
                    run += c;
                }
                tile_n[k] = run;
                all_n[k] = (int)tot[d];
            }
            st += tile_n[k];
            sa += all_n[k];
        }
        int total;
        int ts = block_excl_scan(st, wtot, total);
        int ds = block_excl_scan(sa, wtot, total);
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int d = threadIdx.x * kPer + k;
            if (d < nd) {
                tstart[d] = ts;
                gbase[d] = ds + (int)offs[(size_t)d * g.ntiles] - ts;
            }
            ts += tile_n[k];
            ds += all_n[k];
        }
    }
    __syncthreads();
    // ---- the tile sorted inside LDS
#pragma unroll
    for (int j = 0; j < E; ++j) {
        if (base + j * 64 + lane >= g.n) continue;
        const int lp = tstart[digit[j]] + mine[digit[j]] + rank[j];
        stage[lp] = it[j];
        sdig[lp] = (uint16_t)digit[j];
    }
    __syncthreads();
    // ---- and out, position by position: coalesced
    const int64_t left = g.n - tile0;
    const int count = left < T ? (int)left : T;
#pragma unroll 4
    for (int k = 0; k < E; ++k) {
        const int lp = k * kCsThreads + threadIdx.x;
        if (lp >= count) break;
        job.store(side, (int64_t)(gbase[sdig[lp]] + lp), stage[lp]);
    }
}

struct CsWs {
    uint32_t *count, *offs, *tot;
    int ntiles, e;
    size_t bytes;
};

// positions per thread of a tile.  8 (2,048 positions: 36 KB of LDS, four workgroups per CU) beats 16 at every size measured
// (25 M pairs in 24 batches: scatter 507 against 614 us; 1.2 M pairs in 9: the whole deal 67 against 90 us)
static int cs_e_for(int stride) { (void)stride; return 8; }

static CsWs carve_cs(void *ws, int64_t n, int stride)
{
    CsWs c;
    char *base = (char *)ws;
    size_t off = 0;
    auto take = [&](size_t bytes) { void *q = base + off; off += align_up(bytes, 256); return q; };
    c.e = cs_e_for(stride);
    const int64_t per_tile = (int64_t)kCsThreads * c.e;
    c.ntiles = (int)((n + per_tile - 1) / per_tile);
    if (c.ntiles < 1) c.ntiles = 1;
    c.count = (uint32_t *)take((size_t)2 * c.ntiles * stride * 4);
    c.offs = (uint32_t *)take((size_t)2 * c.ntiles * stride * 4);
    c.tot = (uint32_t *)take((size_t)2 * stride * 4);
    c.bytes = off;
    return c;
}

template <int E, class Job>
static int launch_cs_pass(const Job &job, const CsGeom &g, hipStream_t st)
{
    const size_t smem = (size_t)kCsThreads * E * (sizeof(typename Job::Item) + 2) + (size_t)(kCsWaves + 2) * g.stride * 4;
    // above the 64 KiB default of dynamic LDS: the limit is raised explicitly
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(csort_scatter<E, Job>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((csort_hist<E, Job>), dim3(g.ntiles, 2), dim3(kCsThreads), 0, st, job, g);
    hipLaunchKernelGGL(csort_scan, dim3(g.stride, 2), dim3(kCsThreads), 0, st, g);
    hipLaunchKernelGGL((csort_scatter<E, Job>), dim3(g.ntiles, 2), dim3(kCsThreads), smem, st, job, g);
    return (int)hipGetLastError();
}

// (the workspace was carved for `stride_carved` digits per row: the tile size follows from it, the rows of this pass may be shorter)
template <class Job>
static int run_cs_pass(const Job &job, const CsWs &c, int64_t n, const int (&db)[2], hipStream_t st)
{
    CsGeom g;
    g.n = n;
    g.ntiles = c.ntiles;
    for (int sd = 0; sd < 2; ++sd) { g.db[sd] = db[sd]; g.nd[sd] = 1 << db[sd]; }
    g.stride = g.nd[0] > g.nd[1] ? g.nd[0] : g.nd[1];
    g.count = c.count; g.offs = c.offs; g.tot = c.tot;
    return c.e == 16 ? launch_cs_pass<16, Job>(job, g, st) : launch_cs_pass<8, Job>(job, g, st);
}

static int ceil_log2_64(int64_t v)
{
    int b = 1;
    while (b < 62 && (1ll << b) < v) ++b;
    return b;
}

// ---- the deal -------------------------------------------------------------------------------------------------------------
struct DealSide {
    const int32_t *id, *partner;                // the master order
    const float *w, *y;
    const int32_t *link;                        // col-major: the pair's row-major position; row-major: nullptr (the position itself)
    int32_t *o_id, *o_partner;                  // the epoch, this order
    float *o_w, *o_y;
    int32_t *cache;                             // [n] batch number by master position: the first pass's histogram leaves it for its scatter
    int4 *tmp_pay;                              // between the two passes of a deal of more than 2,048 batches
    int32_t *tmp_batch;
};

struct DealItemLast { int32_t id, partner, w, y; };
struct DealItemMid { int32_t id, partner, w, y, batch; };

template <bool FIRST, bool LAST>
struct DealJob {
    DealSide s[2];
    uint64_t n;
    uint32_t B;
    int bits;
    uint4 key;
    int shift;
    uint32_t mask;
    int bshift;                                 // log2 B when B is a power of two, else -1
    int nt;                                     // a stream beyond the caches: its one pass per epoch goes by non-temporal loads and stores
    int cache8;                                 // at most 256 batches: the batch numbers between histogram and scatter are bytes (25 M pairs: 656 -> 633 us per epoch)
    using Item = typename std::conditional<LAST, DealItemLast, DealItemMid>::type;

    // (called by the histogram: every position exactly once)
    static constexpr bool kSeats = FIRST;       // the first pass's digits are seats of the bijection: csort_hist walks them lane by lane
    __device__ uint32_t position(int side, int64_t i) const { return s[side].link ? (uint32_t)s[side].link[i] : (uint32_t)i; }
    __device__ uint32_t seat_once(uint32_t x) const { return feistel_once(x, bits, key); }
    __device__ bool seated(uint32_t x) const { return x < (uint32_t)n; }
    __device__ int digit_of_seat(int side, int64_t i, uint32_t seat) const
    {
        const int32_t batch = (int32_t)(bshift >= 0 ? seat >> bshift : seat / B);      // (uniform: a batch size that is a power of two)
        if (cache8) reinterpret_cast<uint8_t *>(s[side].cache)[i] = (uint8_t)batch; else s[side].cache[i] = batch;
        return (int)(((uint32_t)batch >> shift) & mask);
    }
    __device__ int digit(int side, int64_t i) const
    {
        if (FIRST) return digit_of_seat(side, i, feistel_walk(position(side, i), (uint32_t)n, bits, key));
        return (int)(((uint32_t)s[side].tmp_batch[i] >> shift) & mask);
    }
    __device__ int load(int side, int64_t i, Item &it) const
    {
        const DealSide &sd = s[side];
        int32_t batch;
        if (FIRST) {
            if (nt) {
                // (read once per epoch: non-temporal, so that the deal does not push the step's table rows out of the L2s)
                it.id = __builtin_nontemporal_load(sd.id + i); it.partner = __builtin_nontemporal_load(sd.partner + i);
                it.w = __float_as_int(__builtin_nontemporal_load(sd.w + i)); it.y = __float_as_int(__builtin_nontemporal_load(sd.y + i));
                batch = cache8 ? (int32_t)__builtin_nontemporal_load(reinterpret_cast<const uint8_t *>(sd.cache) + i) : __builtin_nontemporal_load(sd.cache + i);
            } else {
                it.id = sd.id[i]; it.partner = sd.partner[i];
                it.w = __float_as_int(sd.w[i]); it.y = __float_as_int(sd.y[i]);
                batch = cache8 ? (int32_t)reinterpret_cast<const uint8_t *>(sd.cache)[i] : sd.cache[i];
            }
        } else {
            const int4 v = sd.tmp_pay[i];
            it.id = v.x; it.partner = v.y; it.w = v.z; it.y = v.w;
            batch = sd.tmp_batch[i];
        }
        set_batch(it, batch);
        return (int)(((uint32_t)batch >> shift) & mask);
    }
    __device__ static void set_batch(DealItemLast &, int32_t) {}
    __device__ static void set_batch(DealItemMid &it, int32_t b) { it.batch = b; }
    __device__ void store(int side, int64_t dest, const DealItemLast &it) const
    {
        const DealSide &sd = s[side];
        if (nt) {
            __builtin_nontemporal_store(it.id, sd.o_id + dest); __builtin_nontemporal_store(it.partner, sd.o_partner + dest);
            __builtin_nontemporal_store(__int_as_float(it.w), sd.o_w + dest); __builtin_nontemporal_store(__int_as_float(it.y), sd.o_y + dest);
        } else {
            sd.o_id[dest] = it.id; sd.o_partner[dest] = it.partner;
            sd.o_w[dest] = __int_as_float(it.w); sd.o_y[dest] = __int_as_float(it.y);
        }
    }
    __device__ void store(int side, int64_t dest, const DealItemMid &it) const
    {
        const DealSide &sd = s[side];
        sd.tmp_pay[dest] = make_int4(it.id, it.partner, it.w, it.y);
        sd.tmp_batch[dest] = it.batch;
    }
};

struct DealPlan { int passes, db[2]; int64_t nb; };
static DealPlan deal_plan(int64_t n, int64_t B)
{
    DealPlan p;
    p.nb = (n + B - 1) / B;                      // the last one may be partial: it is dealt too (nobody steps on it)
    const int bits = ceil_log2_64(p.nb > 1 ? p.nb : 2);
    p.passes = bits <= kCsMaxBits ? 1 : 2;
    p.db[0] = p.passes == 1 ? bits : (bits + 1) / 2;
    p.db[1] = p.passes == 1 ? 0 : bits - p.db[0];
    return p;
}

struct DealWs { CsWs cs; int32_t *cache[2]; int4 *pay[2]; int32_t *batch[2]; size_t bytes; };
static DealWs carve_deal(void *ws, int64_t n, int64_t B)
{
    const DealPlan p = deal_plan(n, B);
    DealWs d;
    const int stride = 1 << (p.db[0] > p.db[1] ? p.db[0] : p.db[1]);
    d.cs = carve_cs(ws, n, stride);
    size_t off = d.cs.bytes;
    char *base = (char *)ws;
    auto take = [&](size_t bytes) { void *q = base + off; off += align_up(bytes, 256); return q; };
    for (int sd = 0; sd < 2; ++sd) {
        d.cache[sd] = (int32_t *)take((size_t)n * 4);
        d.pay[sd] = p.passes > 1 ? (int4 *)take((size_t)n * 16) : nullptr;
        d.batch[sd] = p.passes > 1 ? (int32_t *)take((size_t)n * 4) : nullptr;
    }
    d.bytes = off;
    return d;
}

// ---- the masters ------------------------------------------------------------------------------------------------------------
// Both orders are stable LSD sorts of the 64-bit key (major id << 32 | minor id) from stream order — minor digits first, so
// the result is sorted by (major, minor, stream index) — in the same launches (grid.y = order).  An id outside its table counts
// as id 0, what the reference's vocabulary lookup gives an unknown token (reference src/models/estimator.py:26-28), before
// anything is sorted.
struct MasterSide {
    const int32_t *major, *minor;               // the raw ids by stream index
    uint32_t major_below, minor_below;
    const uint64_t *k_in; uint64_t *k_out;      // ping-pong between passes
    const int32_t *i_in; int32_t *i_out;
    int32_t *o_id, *o_partner, *o_perm;         // last pass: the order itself and the stream index of every position
    int shift;
    uint32_t mask;
};

template <bool FIRST, bool LAST>
struct MasterJob {
    MasterSide s[2];
    struct Item { uint64_t key; int32_t idx; };
    __device__ uint64_t key_of(int side, int64_t i) const
    {
        const MasterSide &sd = s[side];
        if (!FIRST) return sd.k_in[i];
        uint32_t a = (uint32_t)sd.major[i], b = (uint32_t)sd.minor[i];
        if (a >= sd.major_below) a = 0;
        if (b >= sd.minor_below) b = 0;
        return (uint64_t)a << 32 | b;
    }
    static constexpr bool kSeats = false;
    __device__ uint32_t position(int, int64_t) const { return 0u; }
    __device__ uint32_t seat_once(uint32_t x) const { return x; }
    __device__ bool seated(uint32_t) const { return true; }
    __device__ int digit_of_seat(int, int64_t, uint32_t) const { return 0; }
    __device__ int digit(int side, int64_t i) const { return (int)((uint32_t)(key_of(side, i) >> s[side].shift) & s[side].mask); }
    __device__ int load(int side, int64_t i, Item &it) const
    {
        it.key = key_of(side, i);
        it.idx = FIRST ? (int32_t)i : s[side].i_in[i];
        return (int)((uint32_t)(it.key >> s[side].shift) & s[side].mask);
    }
    __device__ void store(int side, int64_t dest, const Item &it) const
    {
        const MasterSide &sd = s[side];
        if (LAST) {
            sd.o_id[dest] = (int32_t)(it.key >> 32);
            sd.o_partner[dest] = (int32_t)(uint32_t)it.key;
            sd.o_perm[dest] = it.idx;
        } else {
            sd.k_out[dest] = it.key;
            sd.i_out[dest] = it.idx;
        }
    }
};

// w / y pulled through both permutations, the inverse of the row-major one, and the ids that were mapped to 0
__global__ __launch_bounds__(kBlock) void masters_gather_kernel(const float *__restrict__ w, const float *__restrict__ y,
                                                                const int32_t *__restrict__ row, const int32_t *__restrict__ col,
                                                                uint32_t row_below, uint32_t col_below, int64_t n,
                                                                const int32_t *__restrict__ perm_r, const int32_t *__restrict__ perm_c,
                                                                float *__restrict__ w_r, float *__restrict__ y_r,
                                                                float *__restrict__ w_c, float *__restrict__ y_c,
                                                                int32_t *__restrict__ inv_r, int32_t *__restrict__ mapped)
{
    int bad = 0;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (int64_t)gridDim.x * kBlock) {
        const int32_t pr = perm_r[k], pc = perm_c[k];
        w_r[k] = w[pr]; y_r[k] = y[pr];
        w_c[k] = w[pc]; y_c[k] = y[pc];
        inv_r[pr] = (int32_t)k;
        bad += ((uint32_t)row[k] >= row_below) + ((uint32_t)col[k] >= col_below);
    }
    bad = wave_sum_int(bad);
    if (mapped && (threadIdx.x & 63) == 0 && bad) atomicAdd(mapped, bad);
}

__global__ __launch_bounds__(kBlock) void masters_link_kernel(const int32_t *__restrict__ perm_c, const int32_t *__restrict__ inv_r,
                                                              int64_t n, int32_t *__restrict__ link)
{
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += (int64_t)gridDim.x * kBlock) link[q] = inv_r[perm_c[q]];
}

struct MastersWs { CsWs cs; uint64_t *key[2][2]; int32_t *idx[2][2]; int32_t *perm[2]; int32_t *inv; size_t bytes; };
static MastersWs carve_masters(void *ws, int64_t n)
{
    MastersWs m;
    m.cs = carve_cs(ws, n, kCsMaxDigits);
    size_t off = m.cs.bytes;
    char *base = (char *)ws;
    auto take = [&](size_t bytes) { void *q = base + off; off += align_up(bytes, 256); return q; };
    const size_t nn = (size_t)(n > 0 ? n : 1);
    for (int sd = 0; sd < 2; ++sd)
        for (int i = 0; i < 2; ++i) {
            m.key[sd][i] = (uint64_t *)take(nn * 8);
            m.idx[sd][i] = (int32_t *)take(nn * 4);
        }
    for (int sd = 0; sd < 2; ++sd) m.perm[sd] = (int32_t *)take(nn * 4);
    m.inv = (int32_t *)take(nn * 4);
    m.bytes = off;
    return m;
}

static bool pairs_ok(const glove_pairs *p) { return p && p->id && p->partner && p->w && p->y; }

}  // namespace glove

using namespace glove;

extern "C" {

size_t glove_masters_workspace_bytes(int64_t n)
{
    if (n < 0) return 0;
    return carve_masters(nullptr, n).bytes;
}

int glove_masters_build(const int32_t *row, const int32_t *col, const float *w, const float *y, int64_t n, int32_t V,
                        int32_t V_row, const glove_pairs *row_major, const glove_pairs *col_major, int32_t *link,
                        int32_t *mapped_out, void *ws, size_t ws_bytes, void *stream)
{
    if (n < 0 || V <= 0 || V_row < 0 || V_row > V || n >= (1ll << 31)) return GLOVE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (mapped_out)
        if (hipError_t e = zero_words(mapped_out, 1, st)) return (int)e;
    if (n == 0) return 0;
    if (!row || !col || !w || !y || !pairs_ok(row_major) || !pairs_ok(col_major) || !link || !ws) return GLOVE_E_BADARG;
    const MastersWs m = carve_masters(ws, n);
    if (m.bytes > ws_bytes) return GLOVE_E_WORKSPACE;
    const int32_t Vr = V_row > 0 ? V_row : V;
    // bits of the ids of each order: [order][0 minor, 1 major]
    const int bits[2][2] = {{ceil_log2_64(V), ceil_log2_64(Vr)}, {ceil_log2_64(Vr), ceil_log2_64(V)}};
    int most = 1;
    for (int sd = 0; sd < 2; ++sd)
        for (int f = 0; f < 2; ++f) most = bits[sd][f] > most ? bits[sd][f] : most;
    const int P = (most + kCsMaxBits - 1) / kCsMaxBits;           // passes per field, the same for both orders
    MasterSide base[2];
    base[0] = MasterSide{row, col, (uint32_t)Vr, (uint32_t)V, nullptr, nullptr, nullptr, nullptr,
                         row_major->id, row_major->partner, m.perm[0], 0, 0};
    base[1] = MasterSide{col, row, (uint32_t)V, (uint32_t)Vr, nullptr, nullptr, nullptr, nullptr,
                         col_major->id, col_major->partner, m.perm[1], 0, 0};
    const int total = 2 * P;
    for (int p = 0; p < total; ++p) {
        const int field = p / P, q = p % P;                        // minor digits first
        MasterSide s[2] = {base[0], base[1]};
        int db[2];
        for (int sd = 0; sd < 2; ++sd) {
            const int fb = bits[sd][field], per = (fb + P - 1) / P;
            const int lo = q * per < fb ? q * per : fb, hi = lo + per < fb ? lo + per : fb;
            db[sd] = hi - lo > 0 ? hi - lo : 1;                    // (a pass over no bits of this order: digit 0 for everybody, order kept)
            s[sd].shift = 32 * field + lo;
            s[sd].mask = hi - lo > 0 ? (1u << (hi - lo)) - 1u : 0u;
            s[sd].k_in = m.key[sd][(p + 1) & 1]; s[sd].i_in = m.idx[sd][(p + 1) & 1];
            s[sd].k_out = m.key[sd][p & 1]; s[sd].i_out = m.idx[sd][p & 1];
        }
        int rc;
        if (p == 0 && total == 1) rc = run_cs_pass(MasterJob<true, true>{{s[0], s[1]}}, m.cs, n, db, st);
        else if (p == 0) rc = run_cs_pass(MasterJob<true, false>{{s[0], s[1]}}, m.cs, n, db, st);
        else if (p == total - 1) rc = run_cs_pass(MasterJob<false, true>{{s[0], s[1]}}, m.cs, n, db, st);
        else rc = run_cs_pass(MasterJob<false, false>{{s[0], s[1]}}, m.cs, n, db, st);
        if (rc) return rc;
    }
    const int nbk = blocks_for(n, kBlock);
    hipLaunchKernelGGL(masters_gather_kernel, dim3(nbk), dim3(kBlock), 0, st, w, y, row, col, (uint32_t)Vr, (uint32_t)V, n,
                       (const int32_t *)m.perm[0], (const int32_t *)m.perm[1], row_major->w, row_major->y, col_major->w,
                       col_major->y, m.inv, mapped_out);
    hipLaunchKernelGGL(masters_link_kernel, dim3(nbk), dim3(kBlock), 0, st, (const int32_t *)m.perm[1], (const int32_t *)m.inv, n, link);
    return (int)hipGetLastError();
}

size_t glove_epoch_deal_workspace_bytes(int64_t n, int64_t B)
{
    if (n < 0 || B <= 0) return 0;
    return carve_deal(nullptr, n, B).bytes;
}

int glove_epoch_deal(const glove_pairs *row_major, const glove_pairs *col_major, const int32_t *link, int64_t n, int64_t B,
                     uint64_t key_lo, uint64_t key_hi, const glove_pairs *row_side, const glove_pairs *col_side, void *ws,
                     size_t ws_bytes, void *stream)
{
    if (n < 0 || B <= 0 || n >= (1ll << 31)) return GLOVE_E_BADARG;
    if (n == 0) return 0;
    if (!pairs_ok(row_major) || !pairs_ok(col_major) || !link || !pairs_ok(row_side) || !pairs_ok(col_side) || !ws) return GLOVE_E_BADARG;
    if (row_side->id == row_major->id || col_side->id == col_major->id) return GLOVE_E_BADARG;      // out of place
    const DealPlan dp = deal_plan(n, B);
    if (dp.db[0] > kCsMaxBits || dp.db[1] > kCsMaxBits) return GLOVE_E_BADARG;                      // beyond 4 M batches
    const DealWs d = carve_deal(ws, n, B);
    if (d.bytes > ws_bytes) return GLOVE_E_WORKSPACE;
    const int h = feistel_bits(n);
    if (h < 0) return GLOVE_E_BADARG;
    const uint4 key = make_uint4((uint32_t)key_lo, (uint32_t)(key_lo >> 32), (uint32_t)key_hi, (uint32_t)(key_hi >> 32));
    hipStream_t st = (hipStream_t)stream;
    // both orders in and out, 64 B per pair: from 128 MB on the stream is beyond the L2s and most of the Infinity Cache
    // (measured, whole step with the deal beside it: V = 400 k, B = 1 M 683 -> 672 us; text8, 1.2 M pairs: 29.2 -> 29.9, not taken there)
    const int nt = n * 64 >= (128ll << 20) ? 1 : 0;
    DealSide s[2];
    s[0] = DealSide{row_major->id, row_major->partner, row_major->w, row_major->y, nullptr,
                    row_side->id, row_side->partner, row_side->w, row_side->y, d.cache[0], d.pay[0], d.batch[0]};
    s[1] = DealSide{col_major->id, col_major->partner, col_major->w, col_major->y, link,
                    col_side->id, col_side->partner, col_side->w, col_side->y, d.cache[1], d.pay[1], d.batch[1]};
    int bshift = -1;
    if ((B & (B - 1)) == 0)
        for (bshift = 0; (1ll << bshift) < B; ++bshift) {}
    const int cache8 = (n + B - 1) / B <= 256 ? 1 : 0;
    if (dp.passes == 1) {
        const int db[2] = {dp.db[0], dp.db[0]};
        return run_cs_pass(DealJob<true, true>{{s[0], s[1]}, (uint64_t)n, (uint32_t)B, h, key, 0, (1u << dp.db[0]) - 1u, bshift, nt, cache8}, d.cs, n, db, st);
    }
    const int db0[2] = {dp.db[0], dp.db[0]}, db1[2] = {dp.db[1], dp.db[1]};
    if (int rc = run_cs_pass(DealJob<true, false>{{s[0], s[1]}, (uint64_t)n, (uint32_t)B, h, key, 0, (1u << dp.db[0]) - 1u, bshift, nt, cache8}, d.cs, n, db0, st))
        return rc;
    return run_cs_pass(DealJob<false, true>{{s[0], s[1]}, (uint64_t)n, (uint32_t)B, h, key, dp.db[0], (1u << dp.db[1]) - 1u, bshift, nt, cache8}, d.cs, n, db1, st);
}

}  // extern "C"
