// TEST INFRASTRUCTURE ONLY (never linked into libglove_hip.so): range checks of a dedup index ("plan") on the device,
// launched on a stream BETWEEN glove_plan_build and the step that consumes the plan — also inside a captured hipGraph,
// where no host-side check can look.  The step kernels gather `(uint32) id * row_bytes` from whatever r_partner /
// c_partner / the chunk records hold: one slot a builder left unwritten, or stale from an earlier build, is a wild load.
//
//   errors[0]  r_partner entry outside [0, V)            errors[1]  c_partner entry outside [0, V_row)
//   errors[2]  c_perm entry outside [0, B)               errors[3]  r_to_c entry outside [0, B)
//   errors[4]  c_perm not a bijection / r_to_c not its inverse
//   errors[5]  counts out of range (chunks, ids, heavy)   errors[6]  chunk_id / chunk_start / uniq_rec inconsistent
//   errors[7]  chunk record: id, pair count or a partner slot (padding slots included) out of range
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/glove_hip.h"

namespace {

__device__ inline void flag(int32_t *errors, int which) { atomicAdd(errors + which, 1); }

__global__ void check_pairs(glove_plan p, int32_t V, int32_t Vr, int32_t *errors)
{
    const int64_t B = p.B;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < B; k += (int64_t)gridDim.x * blockDim.x) {
        if ((uint32_t)p.r_partner[k] >= (uint32_t)V) flag(errors, 0);
        if ((uint32_t)p.c_partner[k] >= (uint32_t)Vr) flag(errors, 1);
        if (!p.c_perm) continue;                                        // the optional links between the two orders
        const int32_t q = p.c_perm[k], inv = p.r_to_c[k];
        if ((uint32_t)q >= (uint64_t)B) flag(errors, 2);
        else if (p.r_to_c[q] != (int32_t)k) flag(errors, 4);            // r_to_c inverts c_perm => both are bijections
        if ((uint32_t)inv >= (uint64_t)B) flag(errors, 3);
    }
}

__global__ void check_sides(glove_plan p, int32_t V, int32_t Vr, int32_t *errors)
{
    const int side = blockIdx.y;
    const int32_t *counts = p.counts;
    const int nch = counts[2 * side], nu = counts[2 * side + 1];
    const int32_t own_V = side == 0 ? Vr : V, partner_V = side == 0 ? V : Vr;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (nch < 0 || nch > p.cap_chunks || nu < 0 || nu > p.cap_uniq || nu > nch || counts[4] < 0) flag(errors, 5);
    }
    if (nch < 0 || nch > p.cap_chunks || nu < 0 || nu > p.cap_uniq) return;
    const int32_t *chunk_id = side ? p.c_chunk_id : p.r_chunk_id, *chunk_start = side ? p.c_chunk_start : p.r_chunk_start;
    const int32_t *uniq_rec = side ? p.c_uniq_rec : p.r_uniq_rec, *crec = side ? p.c_crec : p.r_crec;
    const int capP = (p.chunk_cap + 7) / 8 * 8, rd = 4 * ((8 + 6 * (capP / 8 - 1) + 7) / 8 * 8);   // whole 128-byte lines per record
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nch; j += gridDim.x * blockDim.x) {
        const int s = chunk_start[j], e = chunk_start[j + 1];
        if ((uint32_t)chunk_id[j] >= (uint32_t)own_V || s < 0 || e <= s || e - s > p.chunk_cap || e > p.B) flag(errors, 6);
        if (j == nch - 1 && e != p.B) flag(errors, 6);
        if (crec) {
            const int32_t *r = crec + (size_t)j * rd;
            bool bad = (uint32_t)r[0] >= (uint32_t)own_V || r[1] < 1 || r[1] > p.chunk_cap || (uint32_t)r[2] >= (uint32_t)nu;
            const int blocks = (r[1] + 7) / 8;                          // what a reader of this chunk may touch
            for (int b = 0; b < blocks && !bad; ++b)
                for (int t = 0; t < 8; ++t)
                    if ((uint32_t)r[(b == 0 ? 4 : 32 + 24 * (b - 1)) + t] >= (uint32_t)partner_V) bad = true;   // block 0 shares line 0 with the header
            if (bad) flag(errors, 7);
        }
    }
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nu; q += gridDim.x * blockDim.x) {
        const int4 u = reinterpret_cast<const int4 *>(uniq_rec)[q];     // {id, first chunk, chunks, pairs}
        if ((uint32_t)u.x >= (uint32_t)own_V || u.y < 0 || u.z < 1 || u.y + u.z > nch || u.w < u.z || u.w > p.B) flag(errors, 6);
        else if (chunk_id[u.y] != u.x || chunk_id[u.y + u.z - 1] != u.x) flag(errors, 6);
    }
}

}  // namespace

extern "C" int glove_test_check_plan(const glove_plan *plan, int32_t V, int32_t *errors8, void *stream)
{
    if (!plan || !errors8 || V <= 0) return GLOVE_E_BADARG;
    if (plan->B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int32_t Vr = plan->V_row > 0 ? plan->V_row : V;
    const int nb = (int)((plan->B + 255) / 256 < 2048 ? (plan->B + 255) / 256 : 2048);
    if (plan->r_partner && plan->c_partner)        // (a plan of a dealt epoch may keep its pair fields in its chunk records only)
        hipLaunchKernelGGL(check_pairs, dim3(nb), dim3(256), 0, st, *plan, V, Vr, errors8);
    hipLaunchKernelGGL(check_sides, dim3(nb, 2), dim3(256), 0, st, *plan, V, Vr, errors8);
    return (int)hipGetLastError();
}
