// rs_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for RustSolver's regret /
// strategy-sum update path.  Everything here is HBM-bound elementwise work over the LANE axis
// (lane = board * n_clusters + cluster): each thread owns kVec = 4 consecutive lanes, so every
// table / utility row is touched with one 16-byte access per thread (1 KiB per wave
// instruction), all rows of a node are issued up front, and nothing is re-read.  No MFMA, no
// LDS: there is no reuse between lanes to stage.
//
// Numerics follow the Rust reference bit for bit in RS_I32 mode (rules listed in DESIGN.md):
// no FMA contraction, sequential f32 sums in action order, RNE i32->f32, saturating
// NaN->0 f32->int casts, i64-add-then-clamp (cfr.rs:445-461) or wrapping i32 add (cfr.rs:616-619).
#include <cstdlib>

#include "rs_internal.hpp"
#include "rs_device.hpp"
#include "rs_eval.hpp"

#pragma clang fp contract(off)

namespace rs {

static_assert(kVec == kVecD && kPruneThreshold == kPruneThresholdD, "device constants");
static_assert(RS_I32 == kDT_I32 && RS_F32 == kDT_F32 && RS_F16 == kDT_F16 && RS_UPD_CLAMP_I64 == kARITH_CLAMP && RS_UPD_WRAP_I32 == kARITH_WRAP, "enum values");

#define RS_JOB_DECL(T) const T job = jobs[blockIdx.y];   // by value, before any store: one-time scalar loads into SGPRs

// utility of one action for 4 lanes (cfr.rs:314-348 for terminals, child buffers otherwise), in two phases so
// that the loads of all children are in flight together: issue_child only issues the row load (no use of the
// data, hence no s_waitcnt in its branch); finish_child turns the raw row into the utility.
__device__ __forceinline__ void issue_child(const ChildSrc &c, uint32_t v, float (&out)[kVec]) {
    if ((c.kind & 0xff) != CH_CONST) load_f32_row(c.buf, v, out);   // wave-uniform branch
}
__device__ __forceinline__ void finish_child(const ChildSrc &c, uint32_t v, float (&out)[kVec]) {
    (void)v;
    const int kind = c.kind & 0xff;
    if (kind == CH_CONST) {
#pragma unroll
        for (int j = 0; j < kVec; j++) out[j] = c.value;
    } else if (kind == CH_SIGN) {
        sign_to_util(out, (c.kind & 0x100) != 0, c.value);
    }
}

// the thread's vector inside a (possibly tiled) node block of A rows: see NodeJob.row_stride
template <int A>
__device__ __forceinline__ uint32_t tiled_vec(uint32_t v, uint32_t tile_shift) {
    return (((v >> tile_shift) * (uint32_t)A) << tile_shift) + (v & ((1u << tile_shift) - 1u));
}

// prune = true is a property of the DEAL in train() (cfr.rs:219: t > PRUNE_THRESHOLD && q > 0.05): deal batches carry one flag byte per lane
__device__ __forceinline__ void lane_prune_flags(const NodeJob &job, bool prune, uint32_t v, bool (&out)[kVec]) {
    uint32_t w = 0x01010101u;
    if (prune && job.prune_lane) w = reinterpret_cast<const uint32_t *>(job.prune_lane)[v];
#pragma unroll
    for (int j = 0; j < kVec; j++) out[j] = prune && ((w >> (8 * j)) & 0xffu) != 0;
}

// table row of a node for the thread's 4 lanes: vector access (lane model) or gather through cluster ids (deal batches)
template <int DT>
__device__ __forceinline__ void load_table_row(const NodeJob &job, const void *base, uint32_t row_off, uint32_t v,
                                               const unsigned (&idx)[kVec], typename Row<DT>::val (&out)[kVec]) {
    if constexpr (DT == RS_I32) {
        if (job.cidx) {
            gather_i32(base, row_off, idx, out);
            return;
        }
    }
    Row<DT>::load(base, row_off, v, out);
}

// =====================================================================================================
// update kernel: one traverser visit of every lane of the node(s) in `jobs` (blockIdx.y = job)
// algorithmic bytes per lane: A*(4 regret + 4 ssum + 4 util) + 4 reach read, A*(4+4) + 4 written = 20A+8
// =====================================================================================================
template <int A, int DT, int ARITH>
__global__ __launch_bounds__(kBlock) void k_update(const NodeJob *__restrict__ jobs, int flags) {
    RS_JOB_DECL(NodeJob)
    const uint32_t n_vec = job.n_vec, pitch = job.row_stride, tsh = job.tile_shift;
    const bool rmplus = (flags & RS_UPD_RMPLUS) != 0, prune = (flags & RS_UPD_PRUNE) != 0;
    using R = Row<DT>;
    using V = typename R::val;
    for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < n_vec; v += gridDim.x * kBlock) {
        V r[A][kVec], s[A][kVec];
        float u[A][kVec], reach[kVec];
        unsigned idx[kVec] = {0, 0, 0, 0};
        if (job.cidx) load_u32_row(job.cidx, v, idx);
#pragma unroll
        for (int a = 0; a < A; a++) load_table_row<DT>(job, job.regrets, a * pitch, tiled_vec<A>(v, tsh), idx, r[a]);
#pragma unroll
        for (int a = 0; a < A; a++) load_table_row<DT>(job, job.ssum, a * pitch, tiled_vec<A>(v, tsh), idx, s[a]);
#pragma unroll
        for (int a = 0; a < A; a++) issue_child(job.child[a], v, u[a]);
        if (job.reach) load_f32_row(job.reach, v, reach);
        else {
#pragma unroll
            for (int j = 0; j < kVec; j++) reach[j] = job.reach_const;
        }
#pragma unroll
        for (int a = 0; a < A; a++) finish_child(job.child[a], v, u[a]);
        if (job.cidx) mask_tail_lanes(reach, v, job.n_lanes);
        bool prune_l[kVec];
        lane_prune_flags(job, prune, v, prune_l);
        V r0[A][kVec], s0[A][kVec];   // values before the visit (deal batches turn the update into a delta)
#pragma unroll
        for (int a = 0; a < A; a++)
#pragma unroll
            for (int j = 0; j < kVec; j++) { r0[a][j] = r[a][j]; s0[a][j] = s[a][j]; }
        float util[kVec];
#pragma unroll
        for (int j = 0; j < kVec; j++) {
            V rl[A], sl[A];
            float ul[A];
#pragma unroll
            for (int a = 0; a < A; a++) { rl[a] = r[a][j]; sl[a] = s[a][j]; ul[a] = u[a][j]; }
            if constexpr (DT == RS_I32) util[j] = visit_i32<A, ARITH>(rl, sl, ul, reach[j], job.scale, rmplus, prune_l[j]);
            else util[j] = visit_f32<A>(rl, sl, ul, reach[j], job.scale, rmplus);
#pragma unroll
            for (int a = 0; a < A; a++) { r[a][j] = rl[a]; s[a][j] = sl[a]; }
        }
        bool scattered = false;
        if constexpr (DT == RS_I32) {
            if (job.cidx) {
#pragma unroll
                for (int a = 0; a < A; a++) scatter_add_i32(job.dreg, a * pitch, idx, r[a], r0[a]);
#pragma unroll
                for (int a = 0; a < A; a++) scatter_add_i32(job.dssm, a * pitch, idx, s[a], s0[a]);
                scattered = true;
            }
        }
        if (!scattered) {
#pragma unroll
            for (int a = 0; a < A; a++) R::store(job.regrets, a * pitch, tiled_vec<A>(v, tsh), r[a]);
#pragma unroll
            for (int a = 0; a < A; a++) R::store(job.ssum, a * pitch, tiled_vec<A>(v, tsh), s[a]);
        }
        if (job.out_util) store_f32_row(job.out_util, v, util);
    }
}

// opponent / read-only visit: util = sum_a u[a]*sigma[a] (cfr.rs:574,:588).  8A+4 bytes per lane.
template <int A, int DT>
__global__ __launch_bounds__(kBlock) void k_node_util(const NodeJob *__restrict__ jobs, const uint64_t *__restrict__ d_seed) {
    const bool sampled = d_seed != nullptr;
    const unsigned long long seed = sampled ? *d_seed : 0ull;
    RS_JOB_DECL(NodeJob)
    const uint32_t n_vec = job.n_vec, pitch = job.row_stride, tsh = job.tile_shift;
    using R = Row<DT>;
    using V = typename R::val;
    for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < n_vec; v += gridDim.x * kBlock) {
        V r[A][kVec];
        float u[A][kVec];
        unsigned idx[kVec] = {0, 0, 0, 0};
        if (job.cidx) load_u32_row(job.cidx, v, idx);
#pragma unroll
        for (int a = 0; a < A; a++) load_table_row<DT>(job, job.regrets, a * pitch, tiled_vec<A>(v, tsh), idx, r[a]);
#pragma unroll
        for (int a = 0; a < A; a++) issue_child(job.child[a], v, u[a]);
#pragma unroll
        for (int a = 0; a < A; a++) finish_child(job.child[a], v, u[a]);
        float util[kVec];
#pragma unroll
        for (int j = 0; j < kVec; j++) {
            V rl[A];
            float sig[A];
#pragma unroll
            for (int a = 0; a < A; a++) rl[a] = r[a][j];
            regret_match<A, V>(rl, sig);
            float acc = 0.0f;
            if (sampled) {   // cfr.rs:471-476: the value of the ONE sampled action
                const int a_s = weighted_index<A>(sig, sample_bits(seed, job.node_index, (unsigned long long)job.lane_base + (unsigned long long)v * kVec + j));
#pragma unroll
                for (int a = 0; a < A; a++) acc = (a == a_s) ? u[a][j] : acc;
            } else {
#pragma unroll
                for (int a = 0; a < A; a++) acc += u[a][j] * sig[a];
            }
            util[j] = acc;
        }
        store_f32_row(job.out_util, v, util);
    }
}

// opponent reach, top-down: out_reach[a] = sigma[a] * reach (cfr.rs:585) for the children that need it
template <int A, int DT>
__global__ __launch_bounds__(kBlock) void k_reach(const NodeJob *__restrict__ jobs, const uint64_t *__restrict__ d_seed) {
    const bool sampled = d_seed != nullptr;
    const unsigned long long seed = sampled ? *d_seed : 0ull;
    RS_JOB_DECL(NodeJob)
    const uint32_t n_vec = job.n_vec, pitch = job.row_stride, tsh = job.tile_shift;
    using R = Row<DT>;
    using V = typename R::val;
    for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < n_vec; v += gridDim.x * kBlock) {
        V r[A][kVec];
        float reach[kVec], out[A][kVec];
        unsigned idx[kVec] = {0, 0, 0, 0};
        if (job.cidx) load_u32_row(job.cidx, v, idx);
#pragma unroll
        for (int a = 0; a < A; a++) load_table_row<DT>(job, job.regrets, a * pitch, tiled_vec<A>(v, tsh), idx, r[a]);
        if (job.reach) load_f32_row(job.reach, v, reach);
        else {
#pragma unroll
            for (int j = 0; j < kVec; j++) reach[j] = job.reach_const;
        }
#pragma unroll
        for (int j = 0; j < kVec; j++) {
            V rl[A];
            float sig[A];
#pragma unroll
            for (int a = 0; a < A; a++) rl[a] = r[a][j];
            regret_match<A, V>(rl, sig);
            if (sampled) {   // only the sampled action's subtree stays active (NaN reach = inactive lane)
                const int a_s = weighted_index<A>(sig, sample_bits(seed, job.node_index, (unsigned long long)job.lane_base + (unsigned long long)v * kVec + j));
#pragma unroll
                for (int a = 0; a < A; a++) out[a][j] = (a == a_s) ? reach[j] * sig[a] : __builtin_nanf("");
            } else {
#pragma unroll
                for (int a = 0; a < A; a++) out[a][j] = sig[a] * reach[j];
            }
        }
#pragma unroll
        for (int a = 0; a < A; a++)
            if (job.out_reach[a]) store_f32_row(job.out_reach[a], v, out[a]);
    }
}

// traverser node in prune mode, top-down: children of unexplored actions get a NaN reach (= lane inactive
// in that subtree, the reference never recurses there: cfr.rs:379-386); explored ones inherit reach.
template <int A>
__global__ __launch_bounds__(kBlock) void k_prune_reach(const NodeJob *__restrict__ jobs) {
    RS_JOB_DECL(NodeJob)
    const uint32_t n_vec = job.n_vec, pitch = job.row_stride, tsh = job.tile_shift;
    for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < n_vec; v += gridDim.x * kBlock) {
        int32_t r[A][kVec];
        float reach[kVec];
        unsigned idx[kVec] = {0, 0, 0, 0};
        if (job.cidx) load_u32_row(job.cidx, v, idx);
#pragma unroll
        for (int a = 0; a < A; a++) load_table_row<RS_I32>(job, job.regrets, a * pitch, tiled_vec<A>(v, tsh), idx, r[a]);
        if (job.reach) load_f32_row(job.reach, v, reach);
        else {
#pragma unroll
            for (int j = 0; j < kVec; j++) reach[j] = job.reach_const;
        }
        bool prune_l[kVec];
        lane_prune_flags(job, true, v, prune_l);
#pragma unroll
        for (int a = 0; a < A; a++) {
            if (!job.out_reach[a]) continue;
            float out[kVec];
#pragma unroll
            for (int j = 0; j < kVec; j++) out[j] = (!prune_l[j] || r[a][j] > kPruneThreshold) ? reach[j] : __builtin_nanf("");
            store_f32_row(job.out_reach[a], v, out);
        }
    }
}

// bulk get_strategy / get_final_strategy over a node: src[A][pitch] -> dst[A][pitch] f32
template <int A, int DT>
__global__ __launch_bounds__(kBlock) void k_strategy(const void *__restrict__ src, float *__restrict__ dst, uint32_t pitch, uint32_t row_stride, uint32_t tsh) {
    const uint32_t n_vec = pitch / kVec;
    using R = Row<DT>;
    using V = typename R::val;
    for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < n_vec; v += gridDim.x * kBlock) {
        V r[A][kVec];
        float out[A][kVec];
#pragma unroll
        for (int a = 0; a < A; a++) R::load(src, a * row_stride, tiled_vec<A>(v, tsh), r[a]);
#pragma unroll
        for (int j = 0; j < kVec; j++) {
            V rl[A];
            float sig[A];
#pragma unroll
            for (int a = 0; a < A; a++) rl[a] = r[a][j];
            regret_match<A, V>(rl, sig);
#pragma unroll
            for (int a = 0; a < A; a++) out[a][j] = sig[a];
        }
#pragma unroll
        for (int a = 0; a < A; a++) store_f32_row(dst + (size_t)a * pitch, v, out[a]);
    }
}


// ---- public chance nodes (cfr.rs:502-522) ----------------------------------------------------------------
// top-down: child_cfr_reach = cfr_reach * (1.0 / len) for each of the `fan` deals of a parent board
// All chance nodes of one tree depth share a launch (blockIdx.y = chance node).  VEC = 4 when n_clusters % 4 == 0:
// 4 consecutive lanes then belong to one board and move as one 16-byte access.
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_chance_expand(const ChanceJob *__restrict__ jobs) {
    const ChanceJob job = jobs[blockIdx.y];
    // lane counts are < 2^31 per node (checked at table creation): 32-bit index math
    const uint32_t C = job.n_clusters, fan = job.fan;
    const uint32_t n_child = job.n_child_lanes / VEC;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_child; i += gridDim.x * kBlock) {
        const uint32_t l = i * VEC;
        const uint32_t bc = l / C, c = l - bc * C;
        const uint32_t bp = (bc + job.board_off) / fan;   // board_off: this rank's first board when the child round is sharded
        if constexpr (VEC == 4) {
            float rp[4];
            if (job.src) load_f32_row(job.src + (size_t)bp * C + c, 0, rp);
            else rp[0] = rp[1] = rp[2] = rp[3] = job.src_const;
            const float out[4] = {rp[0] * job.inv, rp[1] * job.inv, rp[2] * job.inv, rp[3] * job.inv};
            store_f32_row(job.dst + l, 0, out);
        } else {
            const float rp = job.src ? job.src[(size_t)bp * C + c] : job.src_const;
            job.dst[l] = rp * job.inv;
        }
    }
}
// bottom-up: util = 0 + u_0 + u_1 + ... in deal order (util.store(util.load() + u), cfr.rs:519)
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_chance_reduce(const ChanceJob *__restrict__ jobs) {
    // the scalar fields by value; shard_lo[] stays where it is (a uniform address: scalar loads) -- a by-value copy of the whole descriptor put the array, which `row` indexes
    // with a loop variable, in scratch: 100 bytes of scratch stores per thread in front of the 64 bytes it is there to write (PMC write bytes 2.4x the algorithmic ones)
    struct { const float *src; float *dst; uint32_t shard_world, rank_stride; } job = {jobs[blockIdx.y].src, jobs[blockIdx.y].dst, jobs[blockIdx.y].shard_world, jobs[blockIdx.y].rank_stride};
    const uint32_t *__restrict__ shard_lo = jobs[blockIdx.y].shard_lo;
    const uint32_t C = jobs[blockIdx.y].n_clusters, fan = jobs[blockIdx.y].fan;
    const uint32_t n_par = jobs[blockIdx.y].n_parent_lanes / VEC;
    // row of deal d: contiguous [boards][C], or, when the child round is sharded, inside the owning rank's slot
    auto row = [&](uint32_t l, uint32_t d) -> const float * {
        const uint32_t b = l / C, c = l - b * C;
        const uint32_t gb = b * fan + d;
        if (job.shard_world == 0) return job.src + (size_t)gb * C + c;
        uint32_t g = 0;
        while (g + 1 < job.shard_world && gb >= shard_lo[g + 1]) ++g;
        return job.src + (size_t)g * job.rank_stride + (size_t)(gb - shard_lo[g]) * C + c;
    };
    if constexpr (VEC == 4) {
        // A board's row of C floats starts wherever C puts it (5 000 floats: 32 bytes off a cache line), so the 1 KB window a wave reads per load touches nine lines, the first and
        // the last shared with the neighbouring windows.  A thread therefore owns kWin vectors a wave apart -- the wave's kWin windows are one contiguous stretch, the shared lines
        // are re-read by the SAME wave in its next instruction -- and the loads are the plain kind (a non-temporal load lets go of the line and the neighbour fetches it from memory
        // again).  PMC read bytes over algorithmic: 1.08 (one window per thread, streaming loads) -> 1.04 (plain loads) -> see profiles/r04_config3.md.
        constexpr uint32_t kWin = 4, G = 4;   // G deals' rows in flight together; the sum itself stays in deal order (cfr.rs:519)
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        for (uint32_t w0 = (blockIdx.x * (kBlock / 64) + wave) * (64 * kWin); w0 < n_par; w0 += gridDim.x * (kBlock / 64) * (64 * kWin)) {
            float acc[kWin][4];
            uint32_t li[kWin];
            bool ok[kWin];
#pragma unroll
            for (uint32_t k = 0; k < kWin; k++) {
                const uint32_t i = w0 + k * 64 + lane;
                ok[k] = i < n_par;
                li[k] = (ok[k] ? i : n_par - 1) * 4;
                acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.0f;
            }
            uint32_t d = 0;
            for (; d + G <= fan; d += G) {
                f32x4 u[G][kWin];
#pragma unroll
                for (uint32_t g = 0; g < G; g++)
#pragma unroll
                    for (uint32_t k = 0; k < kWin; k++) u[g][k] = *(as_global<f32x4>(row(li[k], d + g)));
#pragma unroll
                for (uint32_t g = 0; g < G; g++)
#pragma unroll
                    for (uint32_t k = 0; k < kWin; k++) {
                        acc[k][0] = acc[k][0] + u[g][k].x;
                        acc[k][1] = acc[k][1] + u[g][k].y;
                        acc[k][2] = acc[k][2] + u[g][k].z;
                        acc[k][3] = acc[k][3] + u[g][k].w;
                    }
            }
            for (; d < fan; d++)
#pragma unroll
                for (uint32_t k = 0; k < kWin; k++) {
                    const f32x4 x = *(as_global<f32x4>(row(li[k], d)));
                    acc[k][0] = acc[k][0] + x.x;
                    acc[k][1] = acc[k][1] + x.y;
                    acc[k][2] = acc[k][2] + x.z;
                    acc[k][3] = acc[k][3] + x.w;
                }
#pragma unroll
            for (uint32_t k = 0; k < kWin; k++)
                if (ok[k]) *(as_global<f32x4>(job.dst + li[k])) = f32x4{acc[k][0], acc[k][1], acc[k][2], acc[k][3]};   // read by the parent round's kernel next: a plain store keeps it in L2
        }
    } else {
        for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_par; i += gridDim.x * kBlock) {
            float acc = 0.0f;
            for (uint32_t d = 0; d < fan; d++) acc = acc + *row(i, d);
            job.dst[i] = acc;
        }
    }
}

// ---- discount sweep (cfr.rs:250-261): 16 bytes per cell (2 arrays, read + write) ------------------------------
template <int DT>
__device__ __forceinline__ void discount_body(void *__restrict__ regrets, void *__restrict__ ssum, size_t n_vec, float d) {
    using R = Row<DT>;
    using V = typename R::val;
    constexpr size_t esize = (DT == RS_F16) ? 2 : 4;
    constexpr int U = 4;   // vectors per array per thread and trip: 8 x 16-byte loads in flight before the first use
    // a workgroup's trip covers ONE contiguous 16 KB of each array (its U vectors are kBlock apart, not a whole grid apart): the same locality
    // argument as the tiled node blocks
    const size_t chunk = (size_t)kBlock * U;
    for (size_t v0 = (size_t)blockIdx.x * chunk + threadIdx.x; v0 < n_vec; v0 += (size_t)gridDim.x * chunk) {
        constexpr size_t stride = kBlock;
        V r[U][kVec], s[U][kVec];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t v = v0 + (size_t)u * stride;
            if (v < n_vec) {   // 64-bit offsets: a table can exceed 2^32 cells
                R::load((char *)regrets + v * kVec * esize, 0, 0, r[u]);
                R::load((char *)ssum + v * kVec * esize, 0, 0, s[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t v = v0 + (size_t)u * stride;
            if (v < n_vec) {
#pragma unroll
                for (int j = 0; j < kVec; j++) {
                    if constexpr (DT == RS_I32) {
                        r[u][j] = f32_as_i32((float)r[u][j] * d);   // cfr.rs:256
                        s[u][j] = f32_as_i32((float)s[u][j] * d);   // cfr.rs:257
                    } else {
                        r[u][j] = r[u][j] * d;
                        s[u][j] = s[u][j] * d;
                    }
                }
                R::store((char *)regrets + v * kVec * esize, 0, 0, r[u]);
                R::store((char *)ssum + v * kVec * esize, 0, 0, s[u]);
            }
        }
    }
}
template <int DT>
__global__ __launch_bounds__(kBlock) void k_discount(void *__restrict__ regrets, void *__restrict__ ssum, size_t n_vec, float d) {
    discount_body<DT>(regrets, ssum, n_vec, d);
}
// the same over a list of stretches (blockIdx.y): the table WITHOUT the nodes whose kept shadow records are the working copy (rs_solver.cpp solver_kept_primary)
template <int DT>
__global__ __launch_bounds__(kBlock) void k_discount_jobs(const DiscountJob *__restrict__ jobs, float d) {
    const DiscountJob j = jobs[blockIdx.y];
    discount_body<DT>(j.regrets, j.ssum, j.n_vec, d);
}

// ---- sparse deal sweeps: the deals whose reach into a subtree is not NaN, as an index list.  A workgroup counts the live lanes of its 256
// with four ballots, ONE atomic reserves the slots (a first version issued one returning atomic per wave on 72 adjacent counters: they share three
// cache lines, the L2 serialised 1.2 M of them, 8.3 ms per launch; counters now sit 256 B apart).  The order of workgroups in the list is
// arbitrary, which is fine because every consumer of the list commutes.
constexpr uint32_t kMaxParts = 64;
constexpr uint32_t kCompactPerThread = 4;   // lanes per thread and iteration: one slot reservation per workgroup covers 1 024 lanes
__global__ __launch_bounds__(kBlock) void k_compact_live(const CompactJob *__restrict__ jobs) {
    __shared__ uint32_t wave_count[kCompactPerThread][kBlock / 64][kMaxParts];   // live lanes of every (sub-round, wave) = 64 consecutive source entries, per cluster range
    __shared__ uint32_t part_base[kMaxParts];                 // list slot reserved for this workgroup, per cluster range
    const CompactJob *job = jobs + blockIdx.y;
    const uint32_t n_parts = job->n_parts, part_size = job->part_size;
    const float *__restrict__ reach = job->reach;
    const uint32_t *__restrict__ key = job->key;
    const uint32_t lane_in_wave = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t kTile = kBlock * kCompactPerThread;
    const uint32_t src_parts = job->src_list ? job->src_parts : 1u;
    for (uint32_t sq = 0; sq < src_parts; ++sq) {   // the parent's live lists, or once over the whole batch
    const uint32_t n = job->src_list ? job->src_count[(size_t)sq * job->src_count_stride] : job->n_lanes;
    const uint32_t *__restrict__ src = job->src_list ? job->src_list + (size_t)sq * job->src_list_stride : nullptr;
    for (uint32_t base = blockIdx.x * kTile; base < n; base += gridDim.x * kTile) {   // whole workgroups iterate together
        bool live[kCompactPerThread];
        float rv[kCompactPerThread];
        uint32_t deal[kCompactPerThread], part[kCompactPerThread], rank[kCompactPerThread];   // rank: among the live lanes of the same part in this wave and sub-round
#pragma unroll
        for (uint32_t i = 0; i < kCompactPerThread; ++i) {   // sub-round i covers entries base + i*256 .. +255: coalesced reads
            const uint32_t e = base + i * kBlock + threadIdx.x;
            deal[i] = e < n ? (src ? src[e] : e) : 0u;
            const size_t ri = (src && job->pos_rows) ? (size_t)sq * job->src_list_stride + e : (size_t)deal[i];
            if (job->mask) {   // liveness from the parent's mask word; the reach of the live ones alone
                live[i] = e < n && ((job->mask[ri] >> job->bit) & 1u);
                rv[i] = (live[i] && reach) ? reach[ri] : 0.0f;
            } else {
                rv[i] = (e < n && reach) ? reach[ri] : 0.0f;
                live[i] = e < n && rv[i] == rv[i];
            }
            part[i] = (live[i] && key) ? min(key[(size_t)deal[i] * job->key_stride] / part_size, n_parts - 1u) : 0u;
            rank[i] = 0;
        }
        for (uint32_t q = 0; q < n_parts; ++q) {   // ballots only: no barrier inside
#pragma unroll
            for (uint32_t i = 0; i < kCompactPerThread; ++i) {
                const bool mine = live[i] && part[i] == q;
                const unsigned long long ballot = __ballot(mine);
                if (mine) rank[i] = (uint32_t)__popcll(ballot & ((1ull << lane_in_wave) - 1ull));
                if (lane_in_wave == 0) wave_count[i][wave][q] = (uint32_t)__popcll(ballot);
            }
        }
        __syncthreads();
        if (threadIdx.x < n_parts) {   // ONE reservation per workgroup and part, issued by different threads; the part's counts become their exclusive prefix in source order
            uint32_t total = 0;
            for (uint32_t i = 0; i < kCompactPerThread; ++i)
                for (int w = 0; w < kBlock / 64; ++w) {
                    const uint32_t c = wave_count[i][w][threadIdx.x];
                    wave_count[i][w][threadIdx.x] = total;
                    total += c;
                }
            part_base[threadIdx.x] = total ? atomicAdd(job->count + (size_t)threadIdx.x * job->count_stride, total) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t i = 0; i < kCompactPerThread; ++i)
            if (live[i]) {   // slots follow the SOURCE order inside the tile (sub-round, wave, lane): an ordered sweep's lists stay in runs of equal last-round cluster
                const uint32_t slot = part_base[part[i]] + wave_count[i][wave][part[i]] + rank[i];
                job->list[(size_t)part[i] * job->list_stride + slot] = deal[i];
                if (job->rlist) job->rlist[(size_t)part[i] * job->list_stride + slot] = rv[i];
                if (job->plist) job->plist[(size_t)part[i] * job->list_stride + slot] = (src && job->pos_rows) ? sq * job->src_list_stride + (base + i * kBlock + threadIdx.x) : deal[i];
            }
        __syncthreads();   // wave_count / part_base are rewritten by the next iteration
    }
    }
}

// ---- deal batches: SoA table rows -> AoS records for the sweep's gathers (rs_device.hpp gather_rec); reads coalesce over clusters,
// every thread writes its record with 16-byte stores (a wave covers 2-4 KB contiguous)
__global__ __launch_bounds__(kBlock) void k_build_shadow(const ShadowJob *__restrict__ jobs, uint64_t *__restrict__ seed_state, uint32_t *__restrict__ zero, uint32_t n_zero) {
    if (seed_state && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {   // k_next_seed folded in: the tree kernels read state[2] after this launch
        seed_state[2] = sweep_seed(seed_state[0], seed_state[1]);
        seed_state[1] += 1;
    }
    if (zero && blockIdx.y == 0)   // the sweep's list counters, which every compaction of this sweep adds to: zeroed here instead of by a memset in front of each compaction
        for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_zero; i += gridDim.x * kBlock) zero[i] = 0u;
    const ShadowJob *job = jobs + blockIdx.y;
    const uint32_t n = job->n_clusters, pitch = job->pitch, A = job->n_actions, half = job->half;
    const int32_t *__restrict__ reg = job->regrets, *__restrict__ ssm = job->ssum;
    int32_t *__restrict__ dst = job->dst;
    const bool both = job->stride == 2 * half;
    for (uint32_t c = blockIdx.x * kBlock + threadIdx.x; c < n; c += gridDim.x * kBlock) {
        int32_t rec[16];
#pragma unroll
        for (uint32_t a = 0; a < 8; ++a) {
            rec[a] = a < A ? reg[(size_t)a * pitch + c] : 0;
            rec[8 + a] = (both && a < A) ? ssm[(size_t)a * pitch + c] : 0;
        }
        if (!both && job->sigma) {   // an opponent node of this sweep: the record holds get_strategy() of the regrets (infoset.rs:83-102), by the very function the walks use
            float sg[8];
            switch (A) {
#define RS_SIGMA_CASE(A_) case A_: { int rr[A_]; float ss[A_]; for (int a = 0; a < A_; ++a) rr[a] = rec[a]; regret_match<A_, int>(rr, ss); for (int a = 0; a < A_; ++a) sg[a] = ss[a]; } break;
                RS_SIGMA_CASE(1) RS_SIGMA_CASE(2) RS_SIGMA_CASE(3) RS_SIGMA_CASE(4) RS_SIGMA_CASE(5) RS_SIGMA_CASE(6) RS_SIGMA_CASE(7) RS_SIGMA_CASE(8)
#undef RS_SIGMA_CASE
            default: break;
            }
            for (uint32_t a = 0; a < A && a < 8; ++a) rec[a] = __float_as_int(sg[a]);
        }
        i32x4 *out = reinterpret_cast<i32x4 *>(dst + (size_t)c * job->row_stride);
        if (half == 2) {   // two actions: 8 bytes of regrets, or one 16-byte record {r0, r1, s0, s1}
            if (both) out[0] = i32x4{rec[0], rec[1], rec[8], rec[9]};
            else *reinterpret_cast<i32x2 *>(dst + (size_t)c * job->row_stride) = i32x2{rec[0], rec[1]};
        } else if (half == 4) {
            out[0] = i32x4{rec[0], rec[1], rec[2], rec[3]};
            if (both) out[1] = i32x4{rec[8], rec[9], rec[10], rec[11]};
        } else {
            out[0] = i32x4{rec[0], rec[1], rec[2], rec[3]};
            out[1] = i32x4{rec[4], rec[5], rec[6], rec[7]};
            if (both) {
                out[2] = i32x4{rec[8], rec[9], rec[10], rec[11]};
                out[3] = i32x4{rec[12], rec[13], rec[14], rec[15]};
            }
        }
    }
}

// ---- deal batches: the per-deal inputs of a round -- both players' cluster ids, the showdown sign (or utility), the prune flag -- packed into ONE 16-byte record per deal,
// so that a sparse subtree kernel fetches them with one gather per live deal instead of four 4-byte gathers that move a 64-byte sector each (rs_device.hpp gather_attr)
__global__ __launch_bounds__(kBlock) void k_pack_attr(const PackJob *__restrict__ jobs) {
    const PackJob job = jobs[blockIdx.y];
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < job.n; i += gridDim.x * kBlock) {
        u32x4 rec;
        rec.x = job.cid0 ? job.cid0[i] : 0u;
        rec.y = job.cid1 ? job.cid1[i] : 0u;
        rec.z = job.leaf ? __float_as_uint(job.leaf[i]) : 0u;
        rec.w = job.prune ? (uint32_t)job.prune[i] : 1u;   // no flag vector = every deal is traversed with prune = true (cfr.rs:219)
        ((u32x4 *)job.out)[i] = rec;
    }
}

static inline uint32_t grid_for(size_t n_threads_needed);
// ---- ordered deal sweeps: the batch in the order of the traverser's last-round cluster (rs_device.hpp load_arec, seg_add) ----------------------------------------
// A counting sort over the cluster id (at most kOrderMaxBins bins) in two launches.  k_order_hist: every workgroup counts the keys of its chunk of the batch in LDS and
// adds its counts to the global totals (coalesced atomics).  k_order_scatter: every workgroup scans the totals into bin starts (LDS), counts its chunk again, reserves its
// share of every bin it holds with one returning atomic on that bin's cursor, and deals its chunk's records out to their slots (LDS cursors), writing the 32-byte record
// of each deal as it goes -- the per-deal input arrays are read coalesced, once.  The order inside a bin is whatever the atomics make it: every consumer commutes
// (i32 deltas), and nothing the oracle sees depends on it.  tot[] and cursor[] (2 x n_bins words, contiguous) are zeroed by a memset in front of the pair.
__global__ __launch_bounds__(kOrderThreads) void k_order_hist(const OrderJob job) {
    extern __shared__ uint32_t lds_bins[];
    for (uint32_t b = threadIdx.x; b < job.n_bins; b += blockDim.x) lds_bins[b] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * job.chunk, hi = min(job.n, lo + job.chunk);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) atomicAdd(&lds_bins[min(job.key[i], job.n_bins - 1u)], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < job.n_bins; b += blockDim.x) {
        const uint32_t c = lds_bins[b];
        if (c) atomicAdd(job.tot + b, c);
    }
}
__global__ __launch_bounds__(kOrderThreads) void k_order_scatter(const OrderJob job) {
    extern __shared__ uint32_t lds_bins[];   // [n_bins] start of every bin, then this workgroup's cursors; [n_bins] its counts
    __shared__ uint32_t part[kOrderThreads];
    uint32_t *cnt = lds_bins + job.n_bins;
    for (uint32_t b = threadIdx.x; b < job.n_bins; b += blockDim.x) cnt[b] = 0;
    // exclusive scan of tot[] over the bins: thread t owns bins [t * per, (t + 1) * per)
    const uint32_t per = (job.n_bins + blockDim.x - 1) / blockDim.x;
    const uint32_t b0 = min(job.n_bins, threadIdx.x * per), b1 = min(job.n_bins, b0 + per);
    uint32_t sum = 0;
    for (uint32_t b = b0; b < b1; ++b) sum += job.tot[b];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < blockDim.x; d <<= 1) {   // inclusive scan of the per-thread sums
        const uint32_t x = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += x;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t b = b0; b < b1; ++b) {
        lds_bins[b] = run;
        run += job.tot[b];
    }
    const uint32_t lo = blockIdx.x * job.chunk, hi = min(job.n, lo + job.chunk);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) atomicAdd(&cnt[min(job.key[i], job.n_bins - 1u)], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < job.n_bins; b += blockDim.x) {   // this workgroup's share of every bin it holds: consecutive threads, consecutive cursors
        const uint32_t c = cnt[b];
        if (c) lds_bins[b] += atomicAdd(job.cursor + b, c);
    }
    __syncthreads();
    u32x4 *__restrict__ out = (u32x4 *)job.arec;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint32_t slot = atomicAdd(&lds_bins[min(job.key[i], job.n_bins - 1u)], 1u);
        u32x4 a, b;
        a.x = job.cid[0] ? job.cid[0][i] : 0u;
        a.y = job.cid[1] ? job.cid[1][i] : 0u;
        a.z = job.cid[2] ? job.cid[2][i] : 0u;
        a.w = job.cid[3] ? job.cid[3][i] : 0u;
        b.x = job.cid[4] ? job.cid[4][i] : 0u;
        b.y = job.cid[5] ? job.cid[5][i] : 0u;
        b.z = job.leaf ? __float_as_uint(job.leaf[i]) : 0u;
        b.w = (i << 1) | (job.prune ? (uint32_t)(job.prune[i] != 0) : 1u);   // no flag vector = every deal is traversed with prune = true (cfr.rs:219)
        out[2 * (size_t)slot] = a;
        out[2 * (size_t)slot + 1] = b;
    }
}
hipError_t launch_order(const OrderJob &job, hipStream_t stream) {
    if (job.n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(job.tot, 0, size_t(2) * job.n_bins * sizeof(uint32_t), stream);   // tot[] and cursor[]
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_order_hist, dim3(job.n_chunks), dim3(kOrderThreads), size_t(job.n_bins) * sizeof(uint32_t), stream, job);
    hipLaunchKernelGGL(k_order_scatter, dim3(job.n_chunks), dim3(kOrderThreads), size_t(2) * job.n_bins * sizeof(uint32_t), stream, job);
    return hipGetLastError();
}
// the root utilities of an ordered sweep are by rank: hand them out by deal id
__global__ __launch_bounds__(kBlock) void k_unpermute_f32(const float *__restrict__ in, const uint32_t *__restrict__ arec, float *__restrict__ out, uint32_t n) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[arec[8 * (size_t)i + 7] >> 1] = in[i];
}
hipError_t launch_unpermute_f32(const float *in, const void *arec, float *out, uint32_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_unpermute_f32, dim3(grid_for(n)), dim3(kBlock), 0, stream, in, (const uint32_t *)arec, out, n);
    return hipGetLastError();
}

// ---- deal batches: table += delta (wrapping), delta = 0; 32 bytes per cell, whole table, end of every sweep ------
__global__ __launch_bounds__(kBlock) void k_apply_delta(int32_t *__restrict__ regrets, int32_t *__restrict__ dregrets,
                                                        int32_t *__restrict__ ssum, int32_t *__restrict__ dssum, size_t n_vec) {
    const i32x4 zero = {0, 0, 0, 0};
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < n_vec; v += (size_t)gridDim.x * kBlock) {
        const i32x4 dr = ((const i32x4 *)dregrets)[v], ds = ((const i32x4 *)dssum)[v];
        if (dr.x | dr.y | dr.z | dr.w) {
            const i32x4 r = ((i32x4 *)regrets)[v];
            const i32x4 n = {(int)((unsigned)r.x + (unsigned)dr.x), (int)((unsigned)r.y + (unsigned)dr.y),
                             (int)((unsigned)r.z + (unsigned)dr.z), (int)((unsigned)r.w + (unsigned)dr.w)};
            ((i32x4 *)regrets)[v] = n;
            ((i32x4 *)dregrets)[v] = zero;
        }
        if (ds.x | ds.y | ds.z | ds.w) {
            const i32x4 q = ((i32x4 *)ssum)[v];
            const i32x4 n = {(int)((unsigned)q.x + (unsigned)ds.x), (int)((unsigned)q.y + (unsigned)ds.y),
                             (int)((unsigned)q.z + (unsigned)ds.z), (int)((unsigned)q.w + (unsigned)ds.w)};
            ((i32x4 *)ssum)[v] = n;
            ((i32x4 *)dssum)[v] = zero;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_apply_delta_jobs(int32_t *__restrict__ regrets, int32_t *__restrict__ dregrets, int32_t *__restrict__ ssum,
                                                             int32_t *__restrict__ dssum, const ApplyJob *__restrict__ jobs) {
    const ApplyJob job = jobs[blockIdx.y];
    const i32x4 zero = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < job.n_vec; i += (size_t)gridDim.x * kBlock) {
        const size_t v = job.first_vec + i;
        const i32x4 dr = ((const i32x4 *)dregrets)[v], ds = ((const i32x4 *)dssum)[v];
        if (dr.x | dr.y | dr.z | dr.w) {
            const i32x4 r = ((i32x4 *)regrets)[v];
            const i32x4 n = {(int)((unsigned)r.x + (unsigned)dr.x), (int)((unsigned)r.y + (unsigned)dr.y),
                             (int)((unsigned)r.z + (unsigned)dr.z), (int)((unsigned)r.w + (unsigned)dr.w)};
            ((i32x4 *)regrets)[v] = n;
            ((i32x4 *)dregrets)[v] = zero;
        }
        if (ds.x | ds.y | ds.z | ds.w) {
            const i32x4 q = ((i32x4 *)ssum)[v];
            const i32x4 n = {(int)((unsigned)q.x + (unsigned)ds.x), (int)((unsigned)q.y + (unsigned)ds.y),
                             (int)((unsigned)q.z + (unsigned)ds.z), (int)((unsigned)q.w + (unsigned)ds.w)};
            ((i32x4 *)ssum)[v] = n;
            ((i32x4 *)dssum)[v] = zero;
        }
    }
}

// ---- showdown evaluation on the device (cfr.rs:38-46, :324-333): the evaluator itself is rs_eval.hpp ------------
// cards[9][pitch] u8: rows 0-4 board, 5-6 player 0 hole cards, 7-8 player 1 hole cards; sign[lane] = sign(score0 - score1)
__global__ __launch_bounds__(kBlock) void k_showdown_sign(const uint8_t *__restrict__ cards, float *__restrict__ sign, uint32_t n,
                                                          uint32_t pitch) {
    for (uint32_t l = blockIdx.x * kBlock + threadIdx.x; l < n; l += gridDim.x * kBlock) {
        uint32_t m0[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 5; i++) add_card(m0, cards[(size_t)i * pitch + l]);   // TrainHand.board[2..7], cfr.rs:42-44: shared by both hands
        uint32_t m1[4] = {m0[0], m0[1], m0[2], m0[3]};
        add_card(m0, cards[(size_t)5 * pitch + l]);
        add_card(m0, cards[(size_t)6 * pitch + l]);
        add_card(m1, cards[(size_t)7 * pitch + l]);
        add_card(m1, cards[(size_t)8 * pitch + l]);
        const uint32_t s0 = evaluate_suits(m0), s1 = evaluate_suits(m1);
        sign[l] = s0 == s1 ? 0.0f : (s0 > s1 ? 1.0f : -1.0f);                   // cfr.rs:326-333
    }
}

// ---- synthetic fills (bench / tests); mirrored in rustsolver_amd/synth.py ------------------------------------------
// state = {base seed, call index, seed of the current sweep}: first launch of every sampled-opponent plan, so that
// a captured hipGraph advances the sweep seed on every replay without any host involvement
__global__ void k_next_seed(uint64_t *state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        state[2] = sweep_seed(state[0], state[1]);
        state[1] += 1;
    }
}
template <int DT>
__global__ __launch_bounds__(kBlock) void k_fill_random(void *__restrict__ dst, size_t n, uint64_t seed, int64_t lo,
                                                        uint64_t span) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const uint64_t h = splitmix64(seed ^ (i * 0x9E3779B97F4A7C15ull));
        const int64_t val = lo + (int64_t)(h % span);
        if constexpr (DT == RS_I32) ((int32_t *)dst)[i] = (int32_t)val;
        else if constexpr (DT == RS_F32) ((float *)dst)[i] = (float)val;
        else ((_Float16 *)dst)[i] = (_Float16)(float)val;
    }
}
__global__ __launch_bounds__(kBlock) void k_fill_uniform(float *__restrict__ dst, size_t n, uint64_t seed, float lo,
                                                         float width, size_t index_offset) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const uint64_t h = splitmix64(seed ^ ((i + index_offset) * 0x9E3779B97F4A7C15ull));
        const float u = (float)(uint32_t)(h >> 40) * 5.9604644775390625e-08f;  // 24 bits * 2^-24 -> [0,1)
        dst[i] = lo + width * u;
    }
}

// bench inputs as SURVEY.md 8(d) prescribes: one cell in `one_in` gets |regret| > 2.1e9, so that the saturating adds of the clamp update are part of what is timed
__global__ __launch_bounds__(kBlock) void k_plant_saturating(int32_t *__restrict__ regrets, size_t n, uint64_t seed, uint32_t one_in) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const uint64_t h = splitmix64((seed ^ 0x5341545552415445ull) ^ (i * 0x9E3779B97F4A7C15ull));
        if (h % one_in != 0) continue;
        const int32_t mag = 2100000000 + (int32_t)((h >> 20) % 47000000u);
        regrets[i] = (h >> 63) ? -mag : mag;
    }
}
// ... and one value in `one_in` of a utility row gets +-magnitude, which sends (scale * reach) * (u - util) beyond 2^31: the exact i64 branch of the clamp update
__global__ __launch_bounds__(kBlock) void k_plant_outliers(float *__restrict__ dst, size_t n, uint64_t seed, uint32_t one_in, float magnitude) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const uint64_t h = splitmix64((seed ^ 0x4F55544C49455253ull) ^ (i * 0x9E3779B97F4A7C15ull));
        if (h % one_in == 0) dst[i] = (h >> 63) ? -magnitude : magnitude;
    }
}
hipError_t launch_plant_saturating(void *regrets, size_t n_cells, uint64_t seed, uint32_t one_in, hipStream_t stream) {
    hipLaunchKernelGGL(k_plant_saturating, dim3(grid_for(n_cells)), dim3(kBlock), 0, stream, (int32_t *)regrets, n_cells, seed, one_in);
    return hipGetLastError();
}
hipError_t launch_plant_outliers(float *dst, size_t n, uint64_t seed, uint32_t one_in, float magnitude, hipStream_t stream) {
    hipLaunchKernelGGL(k_plant_outliers, dim3(grid_for(n)), dim3(kBlock), 0, stream, dst, n, seed, one_in, magnitude);
    return hipGetLastError();
}

// replicated-round deltas for the multi-GPU all-reduce: x -= snap / x += snap (wrapping for i32)
template <int DT, int SIGN>
__global__ __launch_bounds__(kBlock) void k_delta(void *__restrict__ x, const void *__restrict__ snap, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        if constexpr (DT == RS_I32) {
            uint32_t a = ((uint32_t *)x)[i], b = ((const uint32_t *)snap)[i];
            ((uint32_t *)x)[i] = SIGN > 0 ? a + b : a - b;
        } else {
            float a = ((float *)x)[i], b = ((const float *)snap)[i];
            ((float *)x)[i] = SIGN > 0 ? a + b : a - b;
        }
    }
}

// batched get-infoset (rs_get_infosets): out[a][k] = block cell (action a, lane lanes[k]) of a plain or tiled node block; raw elements (2 or 4 bytes)
template <typename E>
__global__ __launch_bounds__(kBlock) void k_gather_lanes(const E *__restrict__ block, const uint32_t *__restrict__ lanes, size_t n, uint32_t A, uint32_t T,
                                                         E *__restrict__ out) {
    for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (size_t)gridDim.x * kBlock) {
        const size_t l = lanes[k], base = (l / T) * A * T + l % T;
        for (uint32_t a = 0; a < A; ++a) out[(size_t)a * n + k] = block[base + (size_t)a * T];
    }
}

// order-independent checksum of the REAL cells of one node block (rs_table_checksum): sum of splitmix64(cell index ^ bits * GOLD); pitch-padding lanes are skipped
// (kernels that walk only real lanes leave them at whatever the fill put there)
template <typename E>
__global__ __launch_bounds__(kBlock) void k_checksum(const E *__restrict__ x, size_t n, size_t cell_off, uint32_t A, uint32_t T, size_t lanes, unsigned long long *__restrict__ out) {
    unsigned long long acc = 0;
    const size_t tile_cells = (size_t)A * T;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const size_t lane = (i / tile_cells) * T + i % T;
        if (lane < lanes) acc += splitmix64((uint64_t)(cell_off + i) ^ ((uint64_t)x[i] * 0x9E3779B97F4A7C15ull));
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

// rs_selftest_division: div_exact_pos against the compiler's division on pairs (positive integer regret, positive sum >= it) drawn from a hash plus the edges
__global__ __launch_bounds__(kBlock) void k_selftest_division(size_t n, uint64_t seed, unsigned long long *__restrict__ mismatches, float *__restrict__ first_bad) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const uint64_t h = splitmix64(seed ^ (i * 0x9E3779B97F4A7C15ull)), g = splitmix64(h);
        // a: an i32 regret as f32 (all magnitudes: the shift spreads the exponents); b: a sum of up to 8 such values, >= a
        const float a = (float)((((uint32_t)h >> 1) >> ((g >> 8) % 31)) | 1u);   // 1 .. 2^31 - 1
        float extra = (float)(uint32_t)(g >> 32) * (float)((g >> 3) & 7u);
        if ((i & 15) == 0) extra = 0.0f;                               // b == a: quotient 1
        const float b = a + extra, want = a / b, got = div_exact_pos(a, b);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            if (atomicAdd(mismatches, 1ull) == 0) {
                first_bad[0] = a;
                first_bad[1] = b;
            }
        }
    }
}

// d = snap - x;  x = snap;  snap = d: every rank restarts from the bit-identical snapshot (x + (snap - x) is NOT snap in f32, and its rounding error depends on the rank's own x)
template <int DT>
__global__ __launch_bounds__(kBlock) void k_delta_swap(void *__restrict__ x, void *__restrict__ snap, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        if constexpr (DT == RS_I32) {
            const uint32_t a = ((uint32_t *)x)[i], b = ((uint32_t *)snap)[i];
            ((uint32_t *)x)[i] = b;
            ((uint32_t *)snap)[i] = b - a;
        } else {
            const float a = ((float *)x)[i], b = ((float *)snap)[i];
            ((float *)x)[i] = b;
            ((float *)snap)[i] = b - a;
        }
    }
}

// =====================================================================================================
// launchers
// =====================================================================================================
static inline uint32_t grid_for(size_t n_threads_needed) {
    // memory-bound streaming: enough workgroups to fill 256 CUs x 8, grid-stride beyond that
    size_t blocks = (n_threads_needed + kBlock - 1) / kBlock;
    const size_t cap = 256 * 16;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return (uint32_t)blocks;
}

#define RS_DISPATCH_A(A_, MACRO)                                   \
    switch (A_) {                                                  \
    case 1: MACRO(1); break;                                       \
    case 2: MACRO(2); break;                                       \
    case 3: MACRO(3); break;                                       \
    case 4: MACRO(4); break;                                       \
    case 5: MACRO(5); break;                                       \
    case 6: MACRO(6); break;                                       \
    case 7: MACRO(7); break;                                       \
    case 8: MACRO(8); break;                                       \
    default: return hipErrorInvalidValue;                          \
    }

hipError_t launch_update(const NodeJob *d_jobs, int n_jobs, uint32_t max_n_vec, int n_actions,
                         KernelCfg cfg, hipStream_t stream) {
    dim3 grid(grid_for(max_n_vec), (uint32_t)n_jobs), block(kBlock);
    const int arith = cfg.mode & RS_UPD_ARITH_MASK, flags = cfg.mode & ~RS_UPD_ARITH_MASK;
#define RS_UPD(A_)                                                                                               \
    if (cfg.dtype == RS_I32 && arith == RS_UPD_CLAMP_I64)                                                        \
        hipLaunchKernelGGL((k_update<A_, RS_I32, RS_UPD_CLAMP_I64>), grid, block, 0, stream, d_jobs, flags);     \
    else if (cfg.dtype == RS_I32)                                                                                \
        hipLaunchKernelGGL((k_update<A_, RS_I32, RS_UPD_WRAP_I32>), grid, block, 0, stream, d_jobs, flags);      \
    else if (cfg.dtype == RS_F32)                                                                                \
        hipLaunchKernelGGL((k_update<A_, RS_F32, 0>), grid, block, 0, stream, d_jobs, flags);                    \
    else                                                                                                         \
        hipLaunchKernelGGL((k_update<A_, RS_F16, 0>), grid, block, 0, stream, d_jobs, flags)
    RS_DISPATCH_A(n_actions, RS_UPD)
#undef RS_UPD
    return hipGetLastError();
}

hipError_t launch_node_util(const NodeJob *d_jobs, int n_jobs, uint32_t max_n_vec, int n_actions,
                            KernelCfg cfg, const uint64_t *d_seed, hipStream_t stream) {
    dim3 grid(grid_for(max_n_vec), (uint32_t)n_jobs), block(kBlock);
#define RS_NU(A_)                                                                                  \
    if (cfg.dtype == RS_I32) hipLaunchKernelGGL((k_node_util<A_, RS_I32>), grid, block, 0, stream, d_jobs, d_seed); \
    else if (cfg.dtype == RS_F32) hipLaunchKernelGGL((k_node_util<A_, RS_F32>), grid, block, 0, stream, d_jobs, d_seed); \
    else hipLaunchKernelGGL((k_node_util<A_, RS_F16>), grid, block, 0, stream, d_jobs, d_seed)
    RS_DISPATCH_A(n_actions, RS_NU)
#undef RS_NU
    return hipGetLastError();
}

hipError_t launch_reach(const NodeJob *d_jobs, int n_jobs, uint32_t max_n_vec, int n_actions,
                        KernelCfg cfg, const uint64_t *d_seed, hipStream_t stream) {
    dim3 grid(grid_for(max_n_vec), (uint32_t)n_jobs), block(kBlock);
#define RS_RE(A_)                                                                              \
    if (cfg.dtype == RS_I32) hipLaunchKernelGGL((k_reach<A_, RS_I32>), grid, block, 0, stream, d_jobs, d_seed); \
    else if (cfg.dtype == RS_F32) hipLaunchKernelGGL((k_reach<A_, RS_F32>), grid, block, 0, stream, d_jobs, d_seed); \
    else hipLaunchKernelGGL((k_reach<A_, RS_F16>), grid, block, 0, stream, d_jobs, d_seed)
    RS_DISPATCH_A(n_actions, RS_RE)
#undef RS_RE
    return hipGetLastError();
}

hipError_t launch_prune_reach(const NodeJob *d_jobs, int n_jobs, uint32_t max_n_vec, int n_actions,
                              KernelCfg cfg, hipStream_t stream) {
    if (cfg.dtype != RS_I32) return hipErrorInvalidValue;
    dim3 grid(grid_for(max_n_vec), (uint32_t)n_jobs), block(kBlock);
#define RS_PR(A_) hipLaunchKernelGGL((k_prune_reach<A_>), grid, block, 0, stream, d_jobs)
    RS_DISPATCH_A(n_actions, RS_PR)
#undef RS_PR
    return hipGetLastError();
}


// The roots below ONE parent subtree scan the same source -- the parent's live list, or the whole batch below the first root -- each for its own reach row.  One job per
// root walks that source once per root (twelve turn roots: the 4 M-deal batch twelve times) and pays the two barriers and the returning atomic of a tile once per root; here
// ONE workgroup takes a tile of the source for all sibling roots: entries read once, a ballot round per sibling, the siblings' slot reservations issued side by side by
// different threads.  Same lists as k_compact_live writes (slots in source order inside a tile), cluster ranges excluded (n_parts == 1).
constexpr uint32_t kMaxSiblings = 16;
static_assert(kMaxSiblings * kCompactPerThread <= 64, "k_compact_siblings keeps one liveness bit per (sibling, entry of the thread) in a 64-bit word");
__global__ __launch_bounds__(kBlock) void k_compact_siblings(const CompactJob *__restrict__ jobs, const CompactGroup *__restrict__ groups) {
    __shared__ uint32_t wave_count[kMaxSiblings][kCompactPerThread][kBlock / 64];
    __shared__ uint32_t slot_base[kMaxSiblings];
    const CompactGroup g = groups[blockIdx.y];
    const CompactJob *__restrict__ J = jobs + g.first;
    const uint32_t nc = g.n;
    const uint32_t lane_in_wave = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t kTile = kBlock * kCompactPerThread;
    const uint32_t *__restrict__ src = J[0].src_list;
    const uint32_t n = src ? J[0].src_count[0] : J[0].n_lanes;
    const bool by_pos = src && J[0].pos_rows;
    for (uint32_t base = blockIdx.x * kTile; base < n; base += gridDim.x * kTile) {
        uint32_t deal[kCompactPerThread], mw[kCompactPerThread];
        unsigned long long live_bits = 0;   // bit q * kCompactPerThread + i: entry i of this thread is live below sibling q
        const uint32_t *__restrict__ mask = J[0].mask;   // (siblings share their parent: one mask row, a bit each)
#pragma unroll
        for (uint32_t i = 0; i < kCompactPerThread; ++i) {
            const uint32_t e = base + i * kBlock + threadIdx.x;
            deal[i] = e < n ? (src ? src[e] : e) : 0u;
            mw[i] = (mask && e < n) ? mask[by_pos ? (size_t)e : (size_t)deal[i]] : 0u;
        }
        for (uint32_t q = 0; q < nc; ++q) {
            const float *__restrict__ reach = J[q].reach;
            const uint32_t bit = J[q].bit;
#pragma unroll
            for (uint32_t i = 0; i < kCompactPerThread; ++i) {
                const uint32_t e = base + i * kBlock + threadIdx.x;
                bool live;
                if (mask) live = (mw[i] >> bit) & 1u;
                else {
                    const float rv = e < n ? reach[by_pos ? (size_t)e : (size_t)deal[i]] : __builtin_nanf("");
                    live = rv == rv;
                }
                const unsigned long long ballot = __ballot(live);
                if (live) live_bits |= 1ull << (q * kCompactPerThread + i);
                if (lane_in_wave == 0) wave_count[q][i][wave] = (uint32_t)__popcll(ballot);
            }
        }
        __syncthreads();
        if (threadIdx.x < nc) {   // the sibling's counts become their exclusive prefix in source order (sub-round, wave): a live entry then finds its slot with one read
            uint32_t total = 0;
            for (uint32_t i = 0; i < kCompactPerThread; ++i)
                for (int w = 0; w < kBlock / 64; ++w) {
                    const uint32_t c = wave_count[threadIdx.x][i][w];
                    wave_count[threadIdx.x][i][w] = total;
                    total += c;
                }
            slot_base[threadIdx.x] = total ? atomicAdd(J[threadIdx.x].count, total) : 0u;
        }
        __syncthreads();
        for (uint32_t q = 0; q < nc; ++q) {
            const CompactJob &job = J[q];
#pragma unroll
            for (uint32_t i = 0; i < kCompactPerThread; ++i) {
                const bool live = (live_bits >> (q * kCompactPerThread + i)) & 1ull;
                const unsigned long long ballot = __ballot(live);
                if (!live) continue;
                const uint32_t slot = slot_base[q] + wave_count[q][i][wave] + (uint32_t)__popcll(ballot & ((1ull << lane_in_wave) - 1ull));
                const uint32_t e = base + i * kBlock + threadIdx.x;
                job.list[slot] = deal[i];
                if (job.rlist) job.rlist[slot] = job.reach[by_pos ? (size_t)e : (size_t)deal[i]];
                if (job.plist) job.plist[slot] = by_pos ? e : deal[i];
            }
        }
        __syncthreads();   // wave_count / slot_base are rewritten by the next iteration
    }
}
hipError_t launch_compact_siblings(const CompactJob *d_jobs, const CompactGroup *d_groups, int n_groups, uint32_t max_lanes, hipStream_t stream) {
    if (n_groups <= 0) return hipSuccess;
    const size_t tile = size_t(kBlock) * kCompactPerThread;
    dim3 grid((unsigned)std::max<size_t>(1, std::min<size_t>((size_t(max_lanes) + tile - 1) / tile, 2048)), (unsigned)n_groups), block(kBlock);
    hipLaunchKernelGGL(k_compact_siblings, grid, block, 0, stream, d_jobs, d_groups);
    return hipGetLastError();
}
hipError_t launch_compact_live(const CompactJob *d_jobs, int n_jobs, uint32_t max_lanes, hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    const size_t tile = size_t(kBlock) * kCompactPerThread;
    dim3 grid((unsigned)std::max<size_t>(1, std::min<size_t>((size_t(max_lanes) + tile - 1) / tile, 2048)), (unsigned)n_jobs), block(kBlock);
    hipLaunchKernelGGL(k_compact_live, grid, block, 0, stream, d_jobs);
    return hipGetLastError();
}
// one workgroup: trips per job from the live-list counts, exclusive prefix (jobs in chunks of kBlock with a running carry)
__global__ __launch_bounds__(kBlock) void k_worklist(const WorklistBatch batch) {   // by value: the descriptors sit in the kernarg segment (scalar loads)
    const WorklistDesc &D = batch.d[blockIdx.x];
    const unsigned char *__restrict__ blob = D.blob;
    const uint32_t stride = D.stride, off_count = D.off_count, n_jobs = D.n_jobs, deals_per_trip = D.deals_per_trip;
    uint32_t *__restrict__ wl = D.wl;
    __shared__ uint32_t scan[kBlock];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_jobs; base += kBlock) {
        const uint32_t j = base + threadIdx.x;
        uint32_t need = 0;
        if (j < n_jobs) {
            const uint32_t *cp = *reinterpret_cast<const uint32_t *const *>(blob + (size_t)j * stride + off_count);
            need = (*cp + deals_per_trip - 1) / deals_per_trip;
        }
        scan[threadIdx.x] = need;
        __syncthreads();
        for (uint32_t d = 1; d < kBlock; d <<= 1) {   // inclusive scan
            const uint32_t x = threadIdx.x >= d ? scan[threadIdx.x - d] : 0u;
            __syncthreads();
            scan[threadIdx.x] += x;
            __syncthreads();
        }
        if (j < n_jobs) wl[2 + j] = carry + scan[threadIdx.x] - need;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry += scan[kBlock - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        wl[2 + n_jobs] = carry;
        wl[1] = n_jobs;
        wl[0] = 0;
    }
}
hipError_t launch_worklist(const WorklistBatch &batch, int n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_worklist, dim3((unsigned)n), dim3(kBlock), 0, stream, batch);
    return hipGetLastError();
}
hipError_t launch_build_shadow(const ShadowJob *d_jobs, int n_jobs, uint32_t max_clusters, hipStream_t stream, uint64_t *d_seed_state, uint32_t *d_zero, uint32_t n_zero) {
    if (n_jobs <= 0) {
        if (d_zero && n_zero) {
            const hipError_t e = hipMemsetAsync(d_zero, 0, size_t(n_zero) * sizeof(uint32_t), stream);
            if (e != hipSuccess) return e;
        }
        return d_seed_state ? launch_next_seed(d_seed_state, stream) : hipSuccess;
    }
    dim3 grid((unsigned)std::max<size_t>(1, std::min<size_t>((size_t(max_clusters) + kBlock - 1) / kBlock, 1024)), (unsigned)n_jobs), block(kBlock);
    hipLaunchKernelGGL(k_build_shadow, grid, block, 0, stream, d_jobs, d_seed_state, d_zero, n_zero);
    return hipGetLastError();
}
// the way back for KEPT wide records ({regrets, strategy sums}, ShadowJob.stride == 2 * half): the table's rows from the records that were the working copy of a training loop
__global__ __launch_bounds__(kBlock) void k_unbuild_shadow(const ShadowJob *__restrict__ jobs) {
    const ShadowJob *job = jobs + blockIdx.y;
    const uint32_t n = job->n_clusters, pitch = job->pitch, A = job->n_actions, half = job->half;
    int32_t *__restrict__ reg = const_cast<int32_t *>(job->regrets), *__restrict__ ssm = const_cast<int32_t *>(job->ssum);
    const int32_t *__restrict__ src = job->dst;
    for (uint32_t c = blockIdx.x * kBlock + threadIdx.x; c < n; c += gridDim.x * kBlock) {
        const int32_t *rec = src + (size_t)c * job->row_stride;
        for (uint32_t a = 0; a < A; ++a) {
            reg[(size_t)a * pitch + c] = rec[a];
            ssm[(size_t)a * pitch + c] = rec[half + a];
        }
    }
}
hipError_t launch_unbuild_shadow(const ShadowJob *d_jobs, int n_jobs, uint32_t max_clusters, hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    dim3 grid((unsigned)std::max<size_t>(1, std::min<size_t>((size_t(max_clusters) + kBlock - 1) / kBlock, 1024)), (unsigned)n_jobs), block(kBlock);
    hipLaunchKernelGGL(k_unbuild_shadow, grid, block, 0, stream, d_jobs);
    return hipGetLastError();
}
hipError_t launch_pack_attr(const PackJob *d_jobs, int n_jobs, uint32_t max_n, hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    dim3 grid(grid_for(max_n), (unsigned)n_jobs), block(kBlock);
    hipLaunchKernelGGL(k_pack_attr, grid, block, 0, stream, d_jobs);
    return hipGetLastError();
}
hipError_t launch_apply_delta(void *regrets, void *dregrets, void *ssum, void *dssum, size_t n_cells, hipStream_t stream) {
    const size_t n_vec = n_cells / kVec;
    dim3 grid(grid_for(n_vec)), block(kBlock);
    hipLaunchKernelGGL(k_apply_delta, grid, block, 0, stream, (int32_t *)regrets, (int32_t *)dregrets, (int32_t *)ssum,
                       (int32_t *)dssum, n_vec);
    return hipGetLastError();
}
// deal sweeps on f32 tables: one thread per (cluster, row of the node's [2A] delta rows) adds the deltas of the cluster's deals one after the other, in deal order
// E = the table's element (float or _Float16): the sum is f32 whatever the storage, the cell is rounded ONCE, on this write; rmplus: a regret that does not end above 0 ends at 0
template <typename E>
__global__ __launch_bounds__(kBlock) void k_apply_f32_rows(const ApplyF32Job *__restrict__ jobs, uint32_t pitch, int rmplus) {
    const ApplyF32Job job = jobs[blockIdx.y];
    const uint32_t rows = 2 * job.n_actions, n = job.n_clusters * rows;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint32_t x = i / job.n_clusters, c = i - x * job.n_clusters;   // consecutive threads: consecutive clusters of one row
        const float *__restrict__ row = job.rows + (size_t)x * pitch;
        float acc = 0.0f;
        for (uint32_t m = job.start[c]; m < job.start[c + 1]; ++m) acc += row[job.members[m]];
        const bool regret = x < job.n_actions;
        E *cell = (regret ? (E *)job.reg + (size_t)x * job.tpitch : (E *)job.ssm + (size_t)(x - job.n_actions) * job.tpitch) + c;
        float v = (float)*cell + acc;
        if (rmplus && regret && !(v > 0.0f)) v = 0.0f;
        *cell = (E)v;
    }
}
hipError_t launch_apply_f32_rows(const ApplyF32Job *d_jobs, int n_jobs, uint32_t max_clusters, uint32_t pitch, int dtype, bool rmplus, hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    dim3 grid((unsigned)std::max<size_t>(1, std::min<size_t>((size_t(max_clusters) * 2 * RS_MAX_ACTIONS + kBlock - 1) / kBlock, 1024)), (unsigned)n_jobs), block(kBlock);
    if (dtype == RS_F16) hipLaunchKernelGGL((k_apply_f32_rows<_Float16>), grid, block, 0, stream, d_jobs, pitch, rmplus ? 1 : 0);
    else hipLaunchKernelGGL((k_apply_f32_rows<float>), grid, block, 0, stream, d_jobs, pitch, rmplus ? 1 : 0);
    return hipGetLastError();
}
// i32 deal sweeps with delta rows (rs_kernel_forms.delta_rows): workgroup (x, y) sums positions [y * chunk, (y + 1) * chunk) of job x's rows per cluster (the JOB is the fast
// grid axis: workgroups go to the 8 XCDs round-robin in launch order, most lists end after a few chunks, and with 16 chunks on the fast axis the working ones all landed on
// three XCDs -- 3.0 against 0.97 ms per launch).  Streaming: every
// position is read once (key + n_rows deltas, 16-byte loads), added to the LDS tile with ds_add where non-zero, and the tile's non-zero cells go to the delta table
// with one global atomic each (consecutive threads: consecutive cells).  No order dependence anywhere: wrapping integer adds.
constexpr int kRowSumBlock = 512;
constexpr int kRowSumMaxRows = 8;   // RS_MAX_ACTIONS
__global__ __launch_bounds__(kRowSumBlock) void k_row_sums(const RowSumJob *__restrict__ jobs, uint32_t chunk) {
    const RowSumJob J = jobs[blockIdx.x];
    const uint32_t n = J.count ? *J.count : J.n_const;
    const uint32_t lo = blockIdx.y * chunk;   // chunk % 4 == 0
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + chunk);
    extern __shared__ int row_tile[];
    const uint32_t cells = J.n_rows * J.n_clusters;
    for (uint32_t i = threadIdx.x; i < cells; i += kRowSumBlock) row_tile[i] = 0;
    __syncthreads();
    // every workgroup starts at its own place inside its chunk and wraps around: thousands of them stream rows laid out alike, and in step they would all sit on the
    // same memory channels.  The loads of slot it + 1 are issued before slot it is added up (the adds are LDS atomics: nothing else would overlap the next round trip).
    constexpr uint32_t kStep = 4 * kRowSumBlock;
    const uint32_t n_it = (hi - lo + kStep - 1) / kStep, rot = (blockIdx.x * 7u + blockIdx.y * 3u) % n_it;
    const __attribute__((address_space(1))) uint32_t *key = (const __attribute__((address_space(1))) uint32_t *)J.key;
    const __attribute__((address_space(1))) int *rows = (const __attribute__((address_space(1))) int *)J.rows;
    const bool key_vec = (reinterpret_cast<uintptr_t>(J.key) & 15) == 0;   // the key row of a dense job is the caller's cluster-id vector
    auto place = [&](uint32_t it) {
        uint32_t slot = it + rot;
        if (slot >= n_it) slot -= n_it;
        return lo + slot * kStep + 4 * threadIdx.x;
    };
    uint32_t kc[4], kn[4];
    i32x4 dc[kRowSumMaxRows], dn[kRowSumMaxRows];
    auto fetch = [&](uint32_t i, uint32_t (&k)[4], i32x4 (&d)[kRowSumMaxRows]) {   // a full vector of positions i .. i + 3 (< hi)
        if (key_vec) {
            const u32x4 kv = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(key + i);
            k[0] = kv.x; k[1] = kv.y; k[2] = kv.z; k[3] = kv.w;
        } else {
            for (int j = 0; j < 4; j++) k[j] = key[i + j];
        }
#pragma unroll
        for (int r = 0; r < kRowSumMaxRows; ++r)
            if (r < (int)J.n_rows) d[r] = *reinterpret_cast<const __attribute__((address_space(1))) i32x4 *>(rows + (size_t)r * J.pitch + i);
    };
    uint32_t i_cur = place(0);
    bool full_cur = i_cur + 4 <= hi;
    if (full_cur) fetch(i_cur, kc, dc);
    for (uint32_t it = 0; it < n_it; ++it) {
        const uint32_t i_next = it + 1 < n_it ? place(it + 1) : hi;
        const bool full_next = i_next + 4 <= hi;
        if (full_next) fetch(i_next, kn, dn);
        if (full_cur) {
#pragma unroll
            for (int r = 0; r < kRowSumMaxRows; ++r)
                if (r < (int)J.n_rows) {
                    int *t = row_tile + r * J.n_clusters;
                    if (kc[0] == kc[3] && kc[0] == kc[1] && kc[0] == kc[2]) {   // an ordered sweep's lists come in runs of equal cluster: one add for the thread's four positions
                        const int sum4 = (int)((unsigned)dc[r].x + (unsigned)dc[r].y + (unsigned)dc[r].z + (unsigned)dc[r].w);
                        if (sum4) __hip_atomic_fetch_add(t + kc[0], sum4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        continue;
                    }
                    if (dc[r].x) __hip_atomic_fetch_add(t + kc[0], dc[r].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (dc[r].y) __hip_atomic_fetch_add(t + kc[1], dc[r].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (dc[r].z) __hip_atomic_fetch_add(t + kc[2], dc[r].z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (dc[r].w) __hip_atomic_fetch_add(t + kc[3], dc[r].w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
        } else {
            for (uint32_t q = i_cur; q < hi; ++q) {   // the ragged end of the list
                const uint32_t k = key[q];
                for (uint32_t r = 0; r < J.n_rows; ++r) {
                    const int d = rows[(size_t)r * J.pitch + q];
                    if (d) __hip_atomic_fetch_add(row_tile + r * J.n_clusters + k, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        i_cur = i_next;
        full_cur = full_next;
#pragma unroll
        for (int j = 0; j < 4; j++) kc[j] = kn[j];
#pragma unroll
        for (int r = 0; r < kRowSumMaxRows; ++r) dc[r] = dn[r];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cells; i += kRowSumBlock) {
        const int x = row_tile[i];
        if (x == 0) continue;
        const uint32_t r = i / J.n_clusters, c = i - r * J.n_clusters;
        __hip_atomic_fetch_add(J.dst + (size_t)r * J.tpitch + c, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// delta rows of nodes with more clusters than an LDS tile holds (a lossless river abstraction: 180 234): every non-zero delta goes straight into the TABLE's cell of its
// position's cluster.  Runs once the round's walks -- the only readers of these nodes in the sweep -- are done; integer adds commute, so the table ends up as with delta tables
// and an apply pass, without either.
__global__ __launch_bounds__(kBlock) void k_row_apply(const RowSumJob *__restrict__ jobs) {
    const RowSumJob J = jobs[blockIdx.y];
    const uint32_t n = J.count ? *J.count : J.n_const;
    const __attribute__((address_space(1))) uint32_t *key = (const __attribute__((address_space(1))) uint32_t *)J.key;
    const __attribute__((address_space(1))) int *rows = (const __attribute__((address_space(1))) int *)J.rows;
    const bool only_records = J.mirror && J.primary && *J.primary != 0;   // inside a training loop the kept records are the working copy: the table's rows follow at its end
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint32_t k = key[i];
        for (uint32_t r = 0; r < J.n_rows; ++r) {
            const int d = rows[(size_t)r * J.pitch + i];
            if (d) {
                if (!only_records) __hip_atomic_fetch_add(J.dst + (size_t)r * J.tpitch + k, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (J.mirror) __hip_atomic_fetch_add(J.mirror + (size_t)(k * J.mstride) + r, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the node's kept shadow record
            }
        }
    }
}
// Data-parallel deal batches (rs_solver.cpp solver_exchange_deltas): what k_row_apply would add, written out instead as (job, row, cluster, delta) items -- 12 bytes each -- for
// the ranks to exchange: the rows of a direct round hold a few dozen non-zero deltas per walked deal, the table behind them millions of cells per node.
// word 0 = (job index in the plan's row jobs << 3) | row, word 1 = the traverser's cluster, word 2 = the delta.  One cursor reservation per wave; items beyond `cap` are
// counted, not written (the host grows the buffer and asks again).
__global__ __launch_bounds__(kBlock) void k_rows_to_items(const RowSumJob *__restrict__ jobs, uint32_t first_job, uint32_t *__restrict__ items, uint32_t *__restrict__ cursor, uint32_t cap) {
    const RowSumJob J = jobs[first_job + blockIdx.y];
    const uint32_t n = J.count ? *J.count : J.n_const;
    const __attribute__((address_space(1))) uint32_t *key = (const __attribute__((address_space(1))) uint32_t *)J.key;
    const __attribute__((address_space(1))) int *rows = (const __attribute__((address_space(1))) int *)J.rows;
    const uint32_t lane = threadIdx.x & 63u, trips = (n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
    for (uint32_t trip = 0; trip < trips; ++trip) {   // every lane of a wave makes every trip: the ballots below are the whole wave's
        const uint32_t i = (trip * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
        const bool in = i < n;
        const uint32_t k = in ? key[i] : 0u;
        for (uint32_t r = 0; r < J.n_rows; ++r) {
            const int d = in ? rows[(size_t)r * J.pitch + i] : 0;
            const unsigned long long ballot = __ballot(d != 0);
            if (!ballot) continue;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(cursor, (uint32_t)__popcll(ballot));
            base = (uint32_t)__shfl((int)base, 0, 64);
            if (d) {
                const uint32_t at = base + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
                if (at < cap) {
                    items[3 * (size_t)at] = ((first_job + blockIdx.y) << 3) | r;
                    items[3 * (size_t)at + 1] = k;
                    items[3 * (size_t)at + 2] = (uint32_t)d;
                }
            }
        }
    }
}
// ... and every rank's items added to this rank's table (and kept records) exactly as k_row_apply adds its own rows: integer adds, any order, same bits
__global__ __launch_bounds__(kBlock) void k_apply_items(const RowSumJob *__restrict__ jobs, const uint32_t *__restrict__ items, uint32_t n) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint32_t w = items[3 * (size_t)i], k = items[3 * (size_t)i + 1];
        const int d = (int)items[3 * (size_t)i + 2];
        const RowSumJob *J = jobs + (w >> 3);
        const uint32_t r = w & 7u;
        int32_t *mirror = J->mirror;
        const bool only_records = mirror && J->primary && *J->primary != 0;
        if (!only_records) __hip_atomic_fetch_add(J->dst + (size_t)r * J->tpitch + k, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mirror) __hip_atomic_fetch_add(mirror + (size_t)(k * J->mstride) + r, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
hipError_t launch_rows_to_items(const RowSumJob *d_jobs, int first_job, int n_jobs, uint32_t max_entries, uint32_t *d_items, uint32_t *d_cursor, uint32_t cap, hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>((size_t(max_entries) + kBlock - 1) / kBlock, 1), 1024);
    hipLaunchKernelGGL(k_rows_to_items, dim3(blocks, (unsigned)n_jobs), dim3(kBlock), 0, stream, d_jobs, (uint32_t)first_job, d_items, d_cursor, cap);
    return hipGetLastError();
}
hipError_t launch_apply_items(const RowSumJob *d_jobs, const uint32_t *d_items, uint32_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_apply_items, dim3(grid_for(n)), dim3(kBlock), 0, stream, d_jobs, d_items, n);
    return hipGetLastError();
}
// the delta cells of the traverser's own nodes (the ranges table += delta runs over) packed into one contiguous buffer [2][n_vec] for the ranks' sum, and back
__global__ __launch_bounds__(kBlock) void k_pack_cells(int32_t *__restrict__ dregrets, int32_t *__restrict__ dssum, const ApplyJob *__restrict__ jobs, const size_t *__restrict__ pack_off,
                                                       int32_t *__restrict__ packed, size_t total_vec, int unpack) {
    const ApplyJob job = jobs[blockIdx.y];
    const size_t off = pack_off[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < job.n_vec; i += (size_t)gridDim.x * kBlock) {
        const size_t v = job.first_vec + i;
        if (unpack) {
            ((i32x4 *)dregrets)[v] = ((const i32x4 *)packed)[off + i];
            ((i32x4 *)dssum)[v] = ((const i32x4 *)packed)[total_vec + off + i];
        } else {
            ((i32x4 *)packed)[off + i] = ((const i32x4 *)dregrets)[v];
            ((i32x4 *)packed)[total_vec + off + i] = ((const i32x4 *)dssum)[v];
        }
    }
}
hipError_t launch_pack_cells(void *dregrets, void *dssum, const ApplyJob *d_jobs, const size_t *d_pack_off, int n_jobs, size_t max_vec, void *packed, size_t total_vec, bool unpack, hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    dim3 grid((unsigned)std::max<size_t>(1, std::min<size_t>((max_vec + kBlock - 1) / kBlock, 1024)), (unsigned)n_jobs), block(kBlock);
    hipLaunchKernelGGL(k_pack_cells, grid, block, 0, stream, (int32_t *)dregrets, (int32_t *)dssum, d_jobs, d_pack_off, (int32_t *)packed, total_vec, unpack ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_row_apply(const RowSumJob *d_jobs, int n_jobs, uint32_t max_entries, hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    // max_entries = the batch; a job's list holds a fraction of it (about eight jobs -- a subtree's traverser nodes x two arrays -- share a list, and a round's lists hold at most
    // about two walks per deal between them).  A grid sized for the whole batch per job was 147 K workgroups for the 574 jobs of a 64 K-deal batch, nearly all of which found
    // nothing (146 -> 128 us per launch; the rest is the atomics themselves: ~5 M scattered read-modify-writes into 2-5 GB, which two arrays' deltas issued back to back by
    // one thread -- one visit per 32-byte record instead of two -- did not speed up: they execute at the memory side).  Four times the average list; the loop takes the rest.
    const size_t lists = std::max<size_t>(1, size_t(n_jobs) / 8), est = std::min<size_t>(max_entries, size_t(max_entries) * 8 / lists + kBlock);
    const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>((est + kBlock - 1) / kBlock, 1), 1024);
    hipLaunchKernelGGL(k_row_apply, dim3(blocks, (unsigned)n_jobs), dim3(kBlock), 0, stream, d_jobs);
    return hipGetLastError();
}

hipError_t launch_row_sums(const RowSumJob *d_jobs, int n_jobs, uint32_t max_entries, uint32_t chunk, uint32_t max_cells, hipStream_t stream) {
    if (n_jobs <= 0 || max_entries == 0) return hipSuccess;
    chunk = std::max<uint32_t>(std::max<uint32_t>(4, chunk / 4 * 4), (max_entries / 65535 + 4) / 4 * 4);   // grid.y <= 65535
    dim3 grid((unsigned)n_jobs, (max_entries + chunk - 1) / chunk), block(kRowSumBlock);
    hipLaunchKernelGGL(k_row_sums, grid, block, size_t(max_cells) * sizeof(int), stream, d_jobs, chunk);
    return hipGetLastError();
}
hipError_t launch_apply_delta_jobs(void *regrets, void *dregrets, void *ssum, void *dssum, const ApplyJob *d_jobs, int n_jobs, size_t max_vec, hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    dim3 grid((unsigned)std::max<size_t>(1, std::min<size_t>((max_vec + kBlock - 1) / kBlock, 1024)), (unsigned)n_jobs), block(kBlock);
    hipLaunchKernelGGL(k_apply_delta_jobs, grid, block, 0, stream, (int32_t *)regrets, (int32_t *)dregrets, (int32_t *)ssum, (int32_t *)dssum, d_jobs);
    return hipGetLastError();
}
hipError_t launch_showdown_sign(const uint8_t *cards, float *sign, uint32_t n, uint32_t pitch, hipStream_t stream) {
    dim3 grid(grid_for(n)), block(kBlock);
    hipLaunchKernelGGL(k_showdown_sign, grid, block, 0, stream, cards, sign, n, pitch);
    return hipGetLastError();
}
// ---- what does a plain streaming copy reach on THIS card?  (bench.py prices the tree kernel against it beside the 8 TB/s spec) --------------
__global__ __launch_bounds__(kBlock) void k_probe_copy(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, size_t n) {
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += (size_t)gridDim.x * kBlock)
        __builtin_nontemporal_store(__builtin_nontemporal_load(in + v), out + v);
}
hipError_t launch_probe_copy(const void *in, void *out, size_t bytes, unsigned blocks, hipStream_t stream) {
    hipLaunchKernelGGL(k_probe_copy, dim3(blocks), dim3(kBlock), 0, stream, (const f32x4 *)in, (f32x4 *)out, bytes / 16);
    return hipGetLastError();
}

hipError_t launch_next_seed(uint64_t *d_state, hipStream_t stream) {
    hipLaunchKernelGGL(k_next_seed, dim3(1), dim3(64), 0, stream, d_state);
    return hipGetLastError();
}

hipError_t launch_strategy(const void *src, float *dst, uint32_t pitch, uint32_t row_stride, uint32_t tile_shift, int n_actions, int dtype, hipStream_t stream) {
    dim3 grid(grid_for(pitch / kVec)), block(kBlock);
#define RS_ST(A_)                                                                                     \
    if (dtype == RS_I32) hipLaunchKernelGGL((k_strategy<A_, RS_I32>), grid, block, 0, stream, src, dst, pitch, row_stride, tile_shift); \
    else if (dtype == RS_F32) hipLaunchKernelGGL((k_strategy<A_, RS_F32>), grid, block, 0, stream, src, dst, pitch, row_stride, tile_shift); \
    else hipLaunchKernelGGL((k_strategy<A_, RS_F16>), grid, block, 0, stream, src, dst, pitch, row_stride, tile_shift)
    RS_DISPATCH_A(n_actions, RS_ST)
#undef RS_ST
    return hipGetLastError();
}

hipError_t launch_chance_expand(const ChanceJob *d_jobs, int n_jobs, size_t max_child_lanes, bool vec4, hipStream_t stream) {
    dim3 grid(grid_for(max_child_lanes / (vec4 ? 4 : 1)), (uint32_t)n_jobs), block(kBlock);
    if (vec4) hipLaunchKernelGGL((k_chance_expand<4>), grid, block, 0, stream, d_jobs);
    else hipLaunchKernelGGL((k_chance_expand<1>), grid, block, 0, stream, d_jobs);
    return hipGetLastError();
}
hipError_t launch_chance_reduce(const ChanceJob *d_jobs, int n_jobs, size_t max_parent_lanes, bool vec4, hipStream_t stream) {
    dim3 grid(grid_for(vec4 ? (max_parent_lanes / 4 + 3) / 4 : max_parent_lanes), (uint32_t)n_jobs), block(kBlock);   // vec4: a thread owns four vectors, a wave apart
    if (vec4) hipLaunchKernelGGL((k_chance_reduce<4>), grid, block, 0, stream, d_jobs);
    else hipLaunchKernelGGL((k_chance_reduce<1>), grid, block, 0, stream, d_jobs);
    return hipGetLastError();
}

hipError_t launch_discount(void *regrets, void *ssum, size_t n_cells, float d, int dtype, hipStream_t stream) {
    const size_t n_vec = n_cells / kVec;
    dim3 grid(grid_for((n_vec + 3) / 4)), block(kBlock);
    if (dtype == RS_I32) hipLaunchKernelGGL((k_discount<RS_I32>), grid, block, 0, stream, regrets, ssum, n_vec, d);
    else if (dtype == RS_F32) hipLaunchKernelGGL((k_discount<RS_F32>), grid, block, 0, stream, regrets, ssum, n_vec, d);
    else hipLaunchKernelGGL((k_discount<RS_F16>), grid, block, 0, stream, regrets, ssum, n_vec, d);
    return hipGetLastError();
}

hipError_t launch_discount_jobs(const DiscountJob *d_jobs, int n_jobs, size_t max_vec, float d, int dtype, hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    dim3 grid(grid_for((max_vec + 3) / 4), (unsigned)n_jobs), block(kBlock);
    if (dtype == RS_I32) hipLaunchKernelGGL((k_discount_jobs<RS_I32>), grid, block, 0, stream, d_jobs, d);
    else if (dtype == RS_F32) hipLaunchKernelGGL((k_discount_jobs<RS_F32>), grid, block, 0, stream, d_jobs, d);
    else hipLaunchKernelGGL((k_discount_jobs<RS_F16>), grid, block, 0, stream, d_jobs, d);
    return hipGetLastError();
}

hipError_t launch_fill_random(void *dst, size_t n_cells, uint64_t seed, int64_t lo, int64_t hi, int dtype,
                              hipStream_t stream) {
    const uint64_t span = (uint64_t)(hi - lo) + 1;
    dim3 grid(grid_for(n_cells)), block(kBlock);
    if (dtype == RS_I32) hipLaunchKernelGGL((k_fill_random<RS_I32>), grid, block, 0, stream, dst, n_cells, seed, lo, span);
    else if (dtype == RS_F32) hipLaunchKernelGGL((k_fill_random<RS_F32>), grid, block, 0, stream, dst, n_cells, seed, lo, span);
    else hipLaunchKernelGGL((k_fill_random<RS_F16>), grid, block, 0, stream, dst, n_cells, seed, lo, span);
    return hipGetLastError();
}
hipError_t launch_fill_uniform(float *dst, size_t n, uint64_t seed, float lo, float hi, hipStream_t stream, size_t index_offset) {
    dim3 grid(grid_for(n)), block(kBlock);
    hipLaunchKernelGGL(k_fill_uniform, grid, block, 0, stream, dst, n, seed, lo, hi - lo, index_offset);
    return hipGetLastError();
}
// fill and checksum keyed by the LOGICAL cell (node, action, global lane) instead of the element index: a rank that holds a slice of a round's boards (lane_off = first global
// lane of its slice) fills and sums exactly what the unsharded table holds for those cells, whatever the pitches and tilings are
template <typename E>
__global__ __launch_bounds__(kBlock) void k_logical(E *__restrict__ x, size_t n, uint32_t node, uint32_t A, uint32_t T, size_t lanes, size_t lane_off, int op, uint64_t seed,
                                                    int64_t lo, uint64_t span, unsigned long long *__restrict__ out) {
    unsigned long long acc = 0;
    const size_t tile_cells = (size_t)A * T;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const size_t lane = (i / tile_cells) * T + i % T, a = (i % tile_cells) / T;
        if (lane >= lanes) continue;
        const uint64_t key = ((uint64_t)node << 44) ^ ((uint64_t)a << 40) ^ (uint64_t)(lane_off + lane);
        if (op == 0) {
            const int64_t val = lo + (int64_t)(splitmix64(seed ^ (key * 0x9E3779B97F4A7C15ull)) % span);
            if constexpr (sizeof(E) == 4) x[i] = (E)(int32_t)val;
            else x[i] = (E)__builtin_bit_cast(uint16_t, (_Float16)(float)val);
        } else acc += splitmix64(key ^ ((uint64_t)x[i] * 0x9E3779B97F4A7C15ull));
    }
    if (op != 0) {
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
    }
}
hipError_t launch_logical(void *x, size_t n, uint32_t node, uint32_t A, size_t tile, size_t lanes, size_t lane_off, size_t es, int op, uint64_t seed, int64_t lo, int64_t hi,
                          unsigned long long *d_out, hipStream_t stream) {
    dim3 grid(grid_for(n)), block_(kBlock);
    const uint64_t span = (uint64_t)(hi - lo) + 1;
    if (es == 4) hipLaunchKernelGGL((k_logical<uint32_t>), grid, block_, 0, stream, (uint32_t *)x, n, node, A, (uint32_t)tile, lanes, lane_off, op, seed, lo, span, d_out);
    else hipLaunchKernelGGL((k_logical<uint16_t>), grid, block_, 0, stream, (uint16_t *)x, n, node, A, (uint32_t)tile, lanes, lane_off, op, seed, lo, span, d_out);
    return hipGetLastError();
}
hipError_t launch_delta_sub(void *x, const void *snap, size_t n, int dtype, hipStream_t stream) {
    dim3 grid(grid_for(n)), block(kBlock);
    if (dtype == RS_I32) hipLaunchKernelGGL((k_delta<RS_I32, -1>), grid, block, 0, stream, x, snap, n);
    else if (dtype == RS_F32) hipLaunchKernelGGL((k_delta<RS_F32, -1>), grid, block, 0, stream, x, snap, n);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t launch_gather_lanes(const void *block, const uint32_t *d_lanes, size_t n, uint32_t A, size_t tile, size_t es, void *d_out, hipStream_t stream) {
    dim3 grid(grid_for(n)), block_(kBlock);
    if (es == 4) hipLaunchKernelGGL((k_gather_lanes<uint32_t>), grid, block_, 0, stream, (const uint32_t *)block, d_lanes, n, A, (uint32_t)tile, (uint32_t *)d_out);
    else hipLaunchKernelGGL((k_gather_lanes<uint16_t>), grid, block_, 0, stream, (const uint16_t *)block, d_lanes, n, A, (uint32_t)tile, (uint16_t *)d_out);
    return hipGetLastError();
}
hipError_t launch_checksum(const void *x, size_t n, size_t cell_off, uint32_t A, size_t tile, size_t lanes, size_t es, unsigned long long *d_out, hipStream_t stream) {
    dim3 grid(grid_for(n)), block_(kBlock);
    if (es == 4) hipLaunchKernelGGL((k_checksum<uint32_t>), grid, block_, 0, stream, (const uint32_t *)x, n, cell_off, A, (uint32_t)tile, lanes, d_out);
    else hipLaunchKernelGGL((k_checksum<uint16_t>), grid, block_, 0, stream, (const uint16_t *)x, n, cell_off, A, (uint32_t)tile, lanes, d_out);
    return hipGetLastError();
}
hipError_t launch_selftest_division(size_t n, uint64_t seed, unsigned long long *d_mismatches, float *d_first_bad, hipStream_t stream) {
    hipLaunchKernelGGL(k_selftest_division, dim3(grid_for(n)), dim3(kBlock), 0, stream, n, seed, d_mismatches, d_first_bad);
    return hipGetLastError();
}
hipError_t launch_delta_swap(void *x, void *snap, size_t n, int dtype, hipStream_t stream) {
    dim3 grid(grid_for(n)), block(kBlock);
    if (dtype == RS_I32) hipLaunchKernelGGL((k_delta_swap<RS_I32>), grid, block, 0, stream, x, snap, n);
    else if (dtype == RS_F32) hipLaunchKernelGGL((k_delta_swap<RS_F32>), grid, block, 0, stream, x, snap, n);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t launch_delta_add(void *x, const void *snap, size_t n, int dtype, hipStream_t stream) {
    dim3 grid(grid_for(n)), block(kBlock);
    if (dtype == RS_I32) hipLaunchKernelGGL((k_delta<RS_I32, +1>), grid, block, 0, stream, x, snap, n);
    else if (dtype == RS_F32) hipLaunchKernelGGL((k_delta<RS_F32, +1>), grid, block, 0, stream, x, snap, n);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace rs
