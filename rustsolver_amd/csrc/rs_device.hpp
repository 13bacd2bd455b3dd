// rs_device.hpp -- the hand-written per-lane device code shared by the static kernels (rs_kernels.hip)
// and the tree-specialised kernels compiled at run time with hipRTC (rs_jit.cpp).  Self-contained on
// purpose: no includes, only builtin types, so the same text compiles under hipcc and hipRTC.
//
// Numerics (bit-exact contract with the Rust reference in RS_I32 mode): no FMA
// contraction, sequential f32 sums in action order from +0.0, RNE i32->f32, correctly rounded
// division, saturating NaN->0 float->int casts, i64-add-then-clamp (cfr.rs:445-461) or wrapping
// i32 add (cfr.rs:616-619).
#pragma once
#pragma clang fp contract(off)

namespace rs {

#ifndef RS_DEVICE_CONSTS
#define RS_DEVICE_CONSTS
#ifndef RS_LANES
#define RS_LANES 4                             // lanes per thread; generated kernels for small deal batches define 1 (rs_jit.cpp)
#endif
constexpr int kVecD = RS_LANES;
constexpr int kPruneThresholdD = -10000000;    // cfr.rs:352
constexpr int kDT_I32 = 0, kDT_F32 = 1, kDT_F16 = 2;      // == RS_I32 / RS_F32 / RS_F16
constexpr int kARITH_CLAMP = 0, kARITH_WRAP = 1;           // == RS_UPD_CLAMP_I64 / RS_UPD_WRAP_I32
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Every table / utility pointer reaches the kernels through a job descriptor in memory, so the compiler cannot
// know its address space and would emit flat_load / flat_store (which also tick lgkmcnt and return out of
// order).  All of them are hipMalloc'ed global memory: say so, and get global_load_dwordx4 / global_store_dwordx4.
#define RS_GLOBAL __attribute__((address_space(1)))
// streaming accesses: every table row is read once and written once per sweep, so keep it out of the caches' way
#define RS_LOADG(p) __builtin_nontemporal_load(p)
#define RS_STOREG(v, p) __builtin_nontemporal_store(v, p)
template <typename T> __device__ __forceinline__ const RS_GLOBAL T *as_global(const void *p) {
    return (const RS_GLOBAL T *)(unsigned long long)p;
}
template <typename T> __device__ __forceinline__ RS_GLOBAL T *as_global(void *p) { return (RS_GLOBAL T *)(unsigned long long)p; }

// ---- Rust casts ---------------------------------------------------------------------------------
// `f32 as i64` then `+ i64::from(r)` then clamp to i32 (cfr.rs:445-451).  2^32 <= |x| < 2^63 saturates the sum whatever r is, so x is first
// limited to +-2^32 (exact in f32) and the rest is exact i64 work.  |x| >= 2^63 (an infinite utility) is where the cast itself saturates, to
// i64::MAX / i64::MIN, and adding an r of the same sign overflows the i64: a release build wraps (a debug build panics), so the clamp then
// lands on the OPPOSITE limit -- reproduced, like the oracle does.
__device__ __forceinline__ int add_clamp_i64(int r, float x) {
    if (x >= 9223372036854775808.0f) return r > 0 ? -2147483647 - 1 : 2147483647;
    if (x <= -9223372036854775808.0f) return r < 0 ? 2147483647 : -2147483647 - 1;
    if (x != x) x = 0.0f;                                  // NaN -> 0
    x = fminf(fmaxf(x, -4294967296.0f), 4294967296.0f);
    long long sum = (long long)r + (long long)x;            // trunc toward zero
    sum = sum > 2147483647LL ? 2147483647LL : sum;
    sum = sum < -2147483648LL ? -2147483648LL : sum;
    return (int)sum;
}

// `f32 as i32`: truncate, saturate, NaN -> 0 (cfr.rs:256-257, :617, :619)
__device__ __forceinline__ int f32_as_i32(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return -2147483647 - 1;
    return (int)x;
}

__device__ __forceinline__ int add_wrap_i32(int r, float x) {
    return (int)((unsigned)r + (unsigned)f32_as_i32(x));
}

// ---- row access: 4 consecutive lanes of one [pitch] row, as the compute type ---------------------
template <int DT> struct Row;
template <> struct Row<kDT_I32> {
    using val = int;
    static __device__ __forceinline__ void load(const void *base, unsigned row_off, unsigned v, val (&out)[kVecD]) {
#if RS_LANES == 4
        i32x4 x = RS_LOADG(as_global<i32x4>((const int *)base + row_off) + v);
        out[0] = x.x; out[1] = x.y; out[2] = x.z; out[3] = x.w;
#else
        for (int j = 0; j < kVecD; j++) out[j] = RS_LOADG(as_global<int>((const int *)base + row_off) + v * kVecD + j);
#endif
    }
    static __device__ __forceinline__ void store(void *base, unsigned row_off, unsigned v, const val (&in)[kVecD]) {
#if RS_LANES == 4
        i32x4 x = {in[0], in[1], in[2], in[3]};
        RS_STOREG(x, as_global<i32x4>((int *)base + row_off) + v);
#else
        for (int j = 0; j < kVecD; j++) RS_STOREG(in[j], as_global<int>((int *)base + row_off) + v * kVecD + j);
#endif
    }
};
template <> struct Row<kDT_F32> {
    using val = float;
    static __device__ __forceinline__ void load(const void *base, unsigned row_off, unsigned v, val (&out)[kVecD]) {
#if RS_LANES == 4
        f32x4 x = RS_LOADG(as_global<f32x4>((const float *)base + row_off) + v);
        out[0] = x.x; out[1] = x.y; out[2] = x.z; out[3] = x.w;
#else
        for (int j = 0; j < kVecD; j++) out[j] = RS_LOADG(as_global<float>((const float *)base + row_off) + v * kVecD + j);
#endif
    }
    static __device__ __forceinline__ void store(void *base, unsigned row_off, unsigned v, const val (&in)[kVecD]) {
#if RS_LANES == 4
        f32x4 x = {in[0], in[1], in[2], in[3]};
        RS_STOREG(x, as_global<f32x4>((float *)base + row_off) + v);
#else
        for (int j = 0; j < kVecD; j++) RS_STOREG(in[j], as_global<float>((float *)base + row_off) + v * kVecD + j);
#endif
    }
};
template <> struct Row<kDT_F16> {
    using val = float;  // binary16 in HBM, f32 in registers
    static __device__ __forceinline__ void load(const void *base, unsigned row_off, unsigned v, val (&out)[kVecD]) {
#if RS_LANES == 4
        f16x4 x = RS_LOADG(as_global<f16x4>((const _Float16 *)base + row_off) + v);
        out[0] = (float)x.x; out[1] = (float)x.y; out[2] = (float)x.z; out[3] = (float)x.w;
#else
        for (int j = 0; j < kVecD; j++) out[j] = (float)RS_LOADG(as_global<_Float16>((const _Float16 *)base + row_off) + v * kVecD + j);
#endif
    }
    static __device__ __forceinline__ void store(void *base, unsigned row_off, unsigned v, const val (&in)[kVecD]) {
#if RS_LANES == 4
        f16x4 x = {(_Float16)in[0], (_Float16)in[1], (_Float16)in[2], (_Float16)in[3]};  // RNE
        RS_STOREG(x, as_global<f16x4>((_Float16 *)base + row_off) + v);
#else
        for (int j = 0; j < kVecD; j++) RS_STOREG((_Float16)in[j], as_global<_Float16>((_Float16 *)base + row_off) + v * kVecD + j);
#endif
    }
};

__device__ __forceinline__ void load_f32_row(const float *base, unsigned v, float (&out)[kVecD]) {
#if RS_LANES == 4
    f32x4 x = RS_LOADG(as_global<f32x4>(base) + v);
    out[0] = x.x; out[1] = x.y; out[2] = x.z; out[3] = x.w;
#else
    for (int j = 0; j < kVecD; j++) out[j] = RS_LOADG(as_global<float>(base) + v * kVecD + j);
#endif
}
__device__ __forceinline__ void store_f32_row(float *base, unsigned v, const float (&in)[kVecD]) {
#if RS_LANES == 4
    f32x4 x = {in[0], in[1], in[2], in[3]};
    RS_STOREG(x, as_global<f32x4>(base) + v);
#else
    for (int j = 0; j < kVecD; j++) RS_STOREG(in[j], as_global<float>(base) + v * kVecD + j);
#endif
}

// showdown / all-in leaf from a sign row: compare as evaluate() scores (cfr.rs:323-334); p1 flips the view
__device__ __forceinline__ void sign_to_util(float (&x)[kVecD], bool p1, float pot) {
#pragma unroll
    for (int j = 0; j < kVecD; j++) {
        const float s = x[j];
        const bool wins = p1 ? (s < 0.0f) : (s > 0.0f);
        x[j] = (s == 0.0f) ? 0.0f : (wins ? pot : -pot);
    }
}

// ---- deal batches: get-infoset addressing through per-lane cluster ids (cfr.rs:361-375) -------------------------
// A lane is a deal; idx[j] is the dense cluster id get_cluster() returned for the acting player on this round.
// Reads gather from the table (i32 only), writes become atomic adds of (new - old) into a delta table that is
// applied after the sweep: several deals may hit one info set, integer adds commute, so the result is deterministic.
__device__ __forceinline__ void load_u32_row(const unsigned *base, unsigned v, unsigned (&out)[kVecD]) {
#if RS_LANES == 4
    const i32x4 x = *(as_global<i32x4>(base) + v);
    out[0] = (unsigned)x.x; out[1] = (unsigned)x.y; out[2] = (unsigned)x.z; out[3] = (unsigned)x.w;
#else
    for (int j = 0; j < kVecD; j++) out[j] = *(as_global<unsigned>(base) + v * kVecD + j);
#endif
}
template <typename V>
__device__ __forceinline__ void gather_i32(const void *base, unsigned row_off, const unsigned (&idx)[kVecD], V (&out)[kVecD]) {   // V = int or float: a 4-byte table element
    const RS_GLOBAL V *p = as_global<V>((const V *)base + row_off);
#pragma unroll
    for (int j = 0; j < kVecD; j++) out[j] = p[idx[j]];
}
// float tables under deal sweeps (no shadow): a node's regrets from the table's own [A][pitch] rows, binary32 or binary16 in memory, f32 in registers
template <int A, int DT>
__device__ __forceinline__ void gather_tbl(const void *reg, unsigned tpitch, const unsigned (&idx)[kVecD], float (&r)[A][kVecD]) {
    static_assert(DT == kDT_F32 || DT == kDT_F16, "float tables");
#pragma unroll
    for (int a = 0; a < A; a++)
#pragma unroll
        for (int j = 0; j < kVecD; j++) {
            if constexpr (DT == kDT_F32) r[a][j] = as_global<float>((const float *)reg + (size_t)a * tpitch)[idx[j]];
            else r[a][j] = (float)as_global<_Float16>((const _Float16 *)reg + (size_t)a * tpitch)[idx[j]];
        }
}
// AoS shadow of one action node for deal sweeps: record c = [regrets 0..H) [strategy_sum 0..H)], H = 2 ints (A <= 2), 4 (A <= 4) or 8 (A <= 8), built from
// the SoA table at the start of every sweep (k_build_shadow).  A deal then reads a node with ONE load where it can -- 8 bytes of regrets or the whole 16-byte record of a
// two-action node -- instead of one 4-byte load per (action, array): the 64 lanes of a gather touch 64 different cache lines, every one of them a cycle of the CU's L1,
// and those cycles are what the deal kernels run out of (round 3: 34 L1 accesses per walked deal and subtree, half a lookup per CU and cycle).
template <int A>
constexpr int shadow_half() { return A <= 2 ? 2 : (A <= 4 ? 4 : 8); }
template <int A>
__device__ __forceinline__ void gather_rec_half(const RS_GLOBAL int *rec, int (&out)[A][kVecD], int j) {
    if constexpr (A <= 2) {
        const i32x2 lo = *reinterpret_cast<const RS_GLOBAL i32x2 *>(rec);
        out[0][j] = lo.x;
        if (A > 1) out[1 < A ? 1 : 0][j] = lo.y;
    } else {
        const i32x4 lo = *reinterpret_cast<const RS_GLOBAL i32x4 *>(rec);
        out[0][j] = lo.x;
        if (A > 1) out[1 < A ? 1 : 0][j] = lo.y;
        if (A > 2) out[2 < A ? 2 : 0][j] = lo.z;
        if (A > 3) out[3 < A ? 3 : 0][j] = lo.w;
        if (A > 4) {
            const i32x4 hi = *reinterpret_cast<const RS_GLOBAL i32x4 *>(rec + 4);
            out[4 < A ? 4 : 0][j] = hi.x;
            if (A > 5) out[5 < A ? 5 : 0][j] = hi.y;
            if (A > 6) out[6 < A ? 6 : 0][j] = hi.z;
            if (A > 7) out[7 < A ? 7 : 0][j] = hi.w;
        }
    }
}
// A record holds 2H ints -- regrets and strategy_sum -- at the sweep's traverser nodes and H at the opponent's (the shadow is rebuilt per sweep, so it holds what THIS
// traverser's sweep reads).  `stride` (ints between two clusters' records of one node) is the ROW of the node's round subtree and role: a deal addresses every traverser node
// of a round subtree with ONE cluster id and every opponent node with another, so the records of all those nodes sit side by side, one row per cluster
// (rs_solver.cpp setup_table_shadow) -- the 7 gathers a river walk makes at its own nodes land in two cache lines instead of seven
// (idx * stride in 32 bits: setup_table_shadow keeps no shadow of 2^32 ints)
template <int A>
__device__ __forceinline__ void gather_rec(const void *shadow, unsigned stride, const unsigned (&idx)[kVecD], int (&r)[A][kVecD]) {   // regrets only
    const RS_GLOBAL int *p = as_global<int>((const int *)shadow);
#pragma unroll
    for (int j = 0; j < kVecD; j++) gather_rec_half<A>(p + (size_t)(idx[j] * stride), r, j);
}
template <int A>
__device__ __forceinline__ void gather_rec2(const void *shadow, unsigned stride, const unsigned (&idx)[kVecD], int (&r)[A][kVecD], int (&s)[A][kVecD]) {
    constexpr int H = shadow_half<A>();
    const RS_GLOBAL int *p = as_global<int>((const int *)shadow);
#pragma unroll
    for (int j = 0; j < kVecD; j++) {
        if constexpr (A <= 2) {   // one 16-byte record {r0, r1, s0, s1}
            const i32x4 w = *reinterpret_cast<const RS_GLOBAL i32x4 *>(p + (size_t)(idx[j] * stride));
            r[0][j] = w.x;
            s[0][j] = w.z;
            if (A > 1) { r[1 < A ? 1 : 0][j] = w.y; s[1 < A ? 1 : 0][j] = w.w; }
        } else {
            gather_rec_half<A>(p + (size_t)(idx[j] * stride), r, j);
            gather_rec_half<A>(p + (size_t)(idx[j] * stride) + H, s, j);
        }
    }
}
// an opponent node whose shadow record holds its strategy (ShadowJob.sigma): sampled from as it comes
template <int A>
__device__ __forceinline__ void gather_sigma(const void *shadow, unsigned stride, const unsigned (&idx)[kVecD], float (&g)[A][kVecD]) {
    int r[A][kVecD];
    gather_rec<A>(shadow, stride, idx, r);
#pragma unroll
    for (int a = 0; a < A; a++)
#pragma unroll
        for (int j = 0; j < kVecD; j++) g[a][j] = __int_as_float(r[a][j]);
}
// the strategy of the drawn action (hand-off from a reach-down kernel to the walk of the same subtree)
template <int A>
__device__ __forceinline__ void hand_pick(const float (&g)[A][kVecD], const int (&a)[kVecD], float (&out)[kVecD]) {
#pragma unroll
    for (int j = 0; j < kVecD; j++) {
        // the values go through an opaque move first: the compiler otherwise turns the chain of selects over g[k][j] into ONE load at a computed address, which pins g in
        // private memory -- 40-64 bytes of scratch per lane in every reach-down kernel, a store and a dependent load through memory at each handed-over node
        float v[A];
#pragma unroll
        for (int k = 0; k < A; k++) {
            v[k] = g[k][j];
            asm volatile("" : "+v"(v[k]));
        }
        float x = v[0];
#pragma unroll
        for (int k = 1; k < A; k++) x = (a[j] == k) ? v[k] : x;
        out[j] = x;
    }
}
// A node without a shadow (its table is so much larger than the batch that transposing it every sweep costs more than the extra gathers: rs_solver.cpp) is read
// from the table's own [A][pitch] rows, one 4-byte gather per (action, array).  `shadow` is a kernel argument: the branch is uniform.
template <int A, typename V>
__device__ __forceinline__ void gather_node(const void *shadow, unsigned stride, const void *reg, unsigned tpitch, const unsigned (&idx)[kVecD], V (&r)[A][kVecD]) {
    if constexpr (sizeof(V) == sizeof(int) && (V)0.5 == (V)0) {
        if (shadow) {
            gather_rec<A>(shadow, stride, idx, r);
            return;
        }
    }
    {   // f32 tables have no shadow (the solver builds none)
#pragma unroll
        for (int a = 0; a < A; a++) gather_i32(reg, a * tpitch, idx, r[a]);
    }
}
template <int A, typename V>
__device__ __forceinline__ void gather_node2(const void *shadow, unsigned stride, const void *reg, const void *ssm, unsigned tpitch, const unsigned (&idx)[kVecD],
                                             V (&r)[A][kVecD], V (&s)[A][kVecD]) {
    if constexpr (sizeof(V) == sizeof(int) && (V)0.5 == (V)0) {
        if (shadow) {
            gather_rec2<A>(shadow, stride, idx, r, s);
            return;
        }
    }
    {
#pragma unroll
        for (int a = 0; a < A; a++) {
            gather_i32(reg, a * tpitch, idx, r[a]);
            gather_i32(ssm, a * tpitch, idx, s[a]);
        }
    }
}
// ---- staged rows (deal sweeps, list walkers with one deal per lane) -----------------------------------------------------------------------------------------
// A gather of 64 lanes at 64 clusters costs the CU's L1 one lookup per lane whatever the lanes fetch (profiles/r04_deals.md: with every deal reading cluster 0 a river walk
// takes a third of its time), and a walk makes one such gather per node.  Instead the WAVE copies the rows of its 64 deals into LDS with loads whose consecutive lanes read
// consecutive 16-byte chunks of one row: load i of lane l fetches chunk (64 i + l) % CH of the deal of lane (64 i + l) / CH -- an instruction touches 64 / CH rows instead of 64,
// and every line the wave needs crosses the L1 once.  The lanes then read their own row back (ds_read_b128 at lane * CHP chunks; CHP odd: conflict-free).
// CH = 16-byte chunks per row in memory, CHP = per row in LDS.
__device__ __forceinline__ void wave_lds_sync() {   // LDS operations of one wave execute in order: only the compiler has to be told
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int CH, int CHP>
__device__ __forceinline__ void stage_rows(const int *rows, unsigned idx, int *W) {
    static_assert(kVecD == 1, "staged rows take one deal per lane");
    static_assert(CH >= 1 && CH <= 16 && CHP == CH, "rows of at most 16 chunks, back to back in LDS");
    // G lanes fetch one row (G = the power of two at or above CH; lanes k >= CH of a group sit out): load i covers deals 64 / G * i .. -- the lane's share of every address
    // (its group's deal, its chunk) is the same for all i, so the whole copy costs three address registers and immediate offsets
    constexpr int G = CH <= 2 ? 2 : (CH <= 4 ? 4 : (CH <= 8 ? 8 : 16)), PER = 64 / G;
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const unsigned k = lane & (unsigned)(G - 1), dl = lane / (unsigned)G;
    const RS_GLOBAL i32x4 *base = as_global<i32x4>(rows) + k;
    int *dst = W + (dl * (unsigned)CH + k) * 4u;
    i32x4 buf[G];
    __builtin_amdgcn_sched_barrier(0);   // the loads below start AFTER what came before (the read-back of the previous rows): their registers are the kernel's peak otherwise
    unsigned c[G];
#pragma unroll
    for (int i = 0; i < G; i++) c[i] = (unsigned)__builtin_amdgcn_ds_bpermute((int)((dl + (unsigned)(i * PER)) << 2), (int)idx);   // the cluster of the deal of lane PER i + dl;
                                                                                                                                   // by EVERY lane: a lane that sits out reads as 0
    if (k < (unsigned)CH) {
#pragma unroll
        for (int i = 0; i < G; i++) buf[i] = base[(size_t)c[i] * CH];
    }
    wave_lds_sync();   // whoever read the region before is done with it
    if (k < (unsigned)CH) {
#pragma unroll
        for (int i = 0; i < G; i++) *reinterpret_cast<i32x4 *>(dst + i * (PER * CH * 4)) = buf[i];
    }
    wave_lds_sync();
    __builtin_amdgcn_sched_barrier(0);
}
// The same copy for a row that goes through the region in two PARTS -- chunks [0, CA) and [CA, CH) of every row; the traverser's records all go to registers in front of the
// walk, so its row never has to be in LDS as a whole.  Both parts' loads are issued before the first wait; a part is moved into the region (rows of CA / CH - CA chunks back
// to back) once the part before it has been read: `read_a` / `read_b` are the lane's reads of its records of that part.  The region is what limits the waves a CU keeps (one
// deal per lane, 12 chunks: 12 KB per wave = 3 waves per SIMD), and the walks are latency-bound: with the region doubled a 4 M-deal batch takes 7.6 instead of 6.0 ms.
// (One function on purpose: with the loads and the stores in two functions that hand the registers over by reference the compiler kept 207 registers instead of 135.)
template <int CH, int CA, class FA, class FB>
__device__ __forceinline__ void stage_rows_two(const int *rows, unsigned idx, int *W, FA &&read_a, FB &&read_b) {
    static_assert(kVecD == 1, "staged rows take one deal per lane");
    static_assert(CA >= 1 && CA < CH && CH <= 16, "two parts of a row of at most 16 chunks");
    constexpr int CB = CH - CA;
    constexpr int GA = CA <= 2 ? 2 : (CA <= 4 ? 4 : (CA <= 8 ? 8 : 16)), PA = 64 / GA;
    constexpr int GB = CB <= 2 ? 2 : (CB <= 4 ? 4 : (CB <= 8 ? 8 : 16)), PB = 64 / GB;
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const unsigned ka = lane & (unsigned)(GA - 1), da = lane / (unsigned)GA;
    const unsigned kb = lane & (unsigned)(GB - 1), db = lane / (unsigned)GB;
    const RS_GLOBAL i32x4 *base_a = as_global<i32x4>(rows) + ka;
    const RS_GLOBAL i32x4 *base_b = as_global<i32x4>(rows) + CA + kb;
    int *dst_a = W + (da * (unsigned)CA + ka) * 4u;
    int *dst_b = W + (db * (unsigned)CB + kb) * 4u;
    i32x4 buf_a[GA], buf_b[GB];
    __builtin_amdgcn_sched_barrier(0);   // see stage_rows
    unsigned ca[GA], cb[GB];
#pragma unroll
    for (int i = 0; i < GA; i++) ca[i] = (unsigned)__builtin_amdgcn_ds_bpermute((int)((da + (unsigned)(i * PA)) << 2), (int)idx);   // by EVERY lane (see stage_rows)
#pragma unroll
    for (int i = 0; i < GB; i++) cb[i] = (unsigned)__builtin_amdgcn_ds_bpermute((int)((db + (unsigned)(i * PB)) << 2), (int)idx);
    if (ka < (unsigned)CA) {
#pragma unroll
        for (int i = 0; i < GA; i++) buf_a[i] = base_a[(size_t)ca[i] * CH];
    }
    if (kb < (unsigned)CB) {
#pragma unroll
        for (int i = 0; i < GB; i++) buf_b[i] = base_b[(size_t)cb[i] * CH];
    }
    wave_lds_sync();   // whoever read the region before is done with it
    if (ka < (unsigned)CA) {
#pragma unroll
        for (int i = 0; i < GA; i++) *reinterpret_cast<i32x4 *>(dst_a + i * (PA * CA * 4)) = buf_a[i];
    }
    wave_lds_sync();
    read_a();
    wave_lds_sync();
    if (kb < (unsigned)CB) {
#pragma unroll
        for (int i = 0; i < GB; i++) *reinterpret_cast<i32x4 *>(dst_b + i * (PB * CB * 4)) = buf_b[i];
    }
    wave_lds_sync();
    read_b();
    wave_lds_sync();
    __builtin_amdgcn_sched_barrier(0);
}
// a node's record inside the lane's staged row (`rec` = W + lane * CHP * 4 + the node's offset): regrets only / strategy, or regrets and strategy sums
template <int A>
__device__ __forceinline__ void staged_half(const int *rec, int (&out)[A][kVecD]) {
    if constexpr (A <= 2) {
        const i32x2 lo = *reinterpret_cast<const i32x2 *>(rec);
        out[0][0] = lo.x;
        if (A > 1) out[1 < A ? 1 : 0][0] = lo.y;
    } else {
        const i32x4 lo = *reinterpret_cast<const i32x4 *>(rec);
        out[0][0] = lo.x;
        if (A > 1) out[1 < A ? 1 : 0][0] = lo.y;
        if (A > 2) out[2 < A ? 2 : 0][0] = lo.z;
        if (A > 3) out[3 < A ? 3 : 0][0] = lo.w;
        if (A > 4) {
            const i32x4 hi = *reinterpret_cast<const i32x4 *>(rec + 4);
            out[4 < A ? 4 : 0][0] = hi.x;
            if (A > 5) out[5 < A ? 5 : 0][0] = hi.y;
            if (A > 6) out[6 < A ? 6 : 0][0] = hi.z;
            if (A > 7) out[7 < A ? 7 : 0][0] = hi.w;
        }
    }
}
template <int A>
__device__ __forceinline__ void staged_rec2(const int *rec, int (&r)[A][kVecD], int (&s)[A][kVecD]) {
    if constexpr (A <= 2) {   // {r0, r1, s0, s1}
        const i32x4 w = *reinterpret_cast<const i32x4 *>(rec);
        r[0][0] = w.x;
        s[0][0] = w.z;
        if (A > 1) { r[1 < A ? 1 : 0][0] = w.y; s[1 < A ? 1 : 0][0] = w.w; }
    } else {
        staged_half<A>(rec, r);
        staged_half<A>(rec + shadow_half<A>(), s);
    }
}
template <int A>
__device__ __forceinline__ void staged_sigma(const int *rec, float (&g)[A][kVecD]) {
    int r[A][kVecD];
    staged_half<A>(rec, r);
#pragma unroll
    for (int a = 0; a < A; a++) g[a][0] = __int_as_float(r[a][0]);
}
__device__ __forceinline__ void scatter_add_i32(void *base, unsigned row_off, const unsigned (&idx)[kVecD], const int (&now)[kVecD],
                                                const int (&before)[kVecD]) {
    RS_GLOBAL int *p = as_global<int>((int *)base + row_off);
#pragma unroll
    for (int j = 0; j < kVecD; j++) {
        const int delta = (int)((unsigned)now[j] - (unsigned)before[j]);
        if (delta != 0) __hip_atomic_fetch_add(p + idx[j], delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// Delta rows (rs_kernel_forms.delta_rows): no sum inside the walk at all.  Row a of `rows` ([2A][pitch] ints) takes the regret delta of action a at the deal's list
// position, row A + a its strategy-sum delta; k_row_sums adds every row up per cluster afterwards.  Positions the walk did not own (ok[j] false) are never read.
template <int A>
__device__ __forceinline__ void keep_deltas(const int (&r)[A][kVecD], const int (&q)[A][kVecD], const int (&s)[A][kVecD], const int (&t)[A][kVecD], int (&d)[2 * A][kVecD]) {
#pragma unroll
    for (int a = 0; a < A; a++)
#pragma unroll
        for (int j = 0; j < kVecD; j++) {
            d[a][j] = (int)((unsigned)r[a][j] - (unsigned)q[a][j]);
            d[A + a][j] = (int)((unsigned)s[a][j] - (unsigned)t[a][j]);
        }
}
template <int A>
__device__ __forceinline__ void store_delta_rows(int *rows, unsigned pitch, unsigned v, const bool (&ok)[kVecD], const int (&d)[2 * A][kVecD]) {
    RS_GLOBAL int *p = as_global<int>(rows) + (size_t)v * kVecD;
#pragma unroll
    for (int x = 0; x < 2 * A; x++) {
#if RS_LANES == 4
        if (ok[0] && ok[1] && ok[2] && ok[3]) {
            const i32x4 w = {d[x][0], d[x][1], d[x][2], d[x][3]};
            *reinterpret_cast<RS_GLOBAL i32x4 *>(p + (size_t)x * pitch) = w;
            continue;
        }
#endif
#pragma unroll
        for (int j = 0; j < kVecD; j++)
            if (ok[j]) p[(size_t)x * pitch + j] = d[x][j];
    }
}
// LDS-privatised form for deal batches: a workgroup first sums its deltas for one traverser node in LDS
// (ds_add, conflict = same cell only), then flushes the non-zero cells with COALESCED global atomics
// (consecutive threads -> consecutive cells), instead of 64 lanes adding into 64 different rows.
// Layout of `lds`: [2][A][tpitch] ints (regret deltas, then strategy_sum deltas), same indexing as the delta tables.
__device__ __forceinline__ void lds_zero(int *lds, unsigned n) {
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) lds[i] = 0;
    __syncthreads();
}
__device__ __forceinline__ void lds_add_i32(int *lds, unsigned row_off, const unsigned (&idx)[kVecD], const int (&now)[kVecD],
                                            const int (&before)[kVecD]) {
#pragma unroll
    for (int j = 0; j < kVecD; j++) {
        const int delta = (int)((unsigned)now[j] - (unsigned)before[j]);
        if (delta != 0) __hip_atomic_fetch_add(lds + row_off + idx[j], delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
__device__ __forceinline__ void lds_flush(int *lds, unsigned n_cells, void *dreg, void *dssm) {
    __syncthreads();
    RS_GLOBAL int *pr = as_global<int>(dreg), *ps = as_global<int>(dssm);
    for (unsigned i = threadIdx.x; i < n_cells; i += blockDim.x) {
        const int a = lds[i], b = lds[n_cells + i];
        if (a != 0) __hip_atomic_fetch_add(pr + i, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b != 0) __hip_atomic_fetch_add(ps + i, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
}

// Tiles may cover only the cluster range [c0, c0 + rcount) of a node (cluster-partitioned workgroups: every deal of the job has its traverser
// cluster in that range): tile row a starts at a * rp, cell (a, c - c0) belongs to table cell a * tpitch + c.
__device__ __forceinline__ void lds_flush_part(int *tile, unsigned n_actions, unsigned rp, unsigned rcount, unsigned c0, unsigned tpitch, void *dreg,
                                               void *dssm) {
    __syncthreads();
    RS_GLOBAL int *pr = as_global<int>(dreg), *ps = as_global<int>(dssm);
    const unsigned half = n_actions * rp;
    for (unsigned i = threadIdx.x; i < half; i += blockDim.x) {
        const unsigned a = i / rp, c = i - a * rp;
        if (c >= rcount) continue;
        const int x = tile[i], y = tile[half + i];
        const unsigned g = a * tpitch + c0 + c;
        if (x != 0) __hip_atomic_fetch_add(pr + g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (y != 0) __hip_atomic_fetch_add(ps + g, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
}

// lanes past the end of the batch (pitch padding) must not touch the table: mark them inactive
__device__ __forceinline__ void mask_tail_lanes(float (&reach)[kVecD], unsigned v, unsigned n_lanes) {
#pragma unroll
    for (int j = 0; j < kVecD; j++)
        if (v * kVecD + j >= n_lanes) reach[j] = __builtin_nanf("");
}

// a / b, correctly rounded, for POSITIVE NORMAL a and b whose quotient is far from the f32 range limits (2^-96 < a / b < 2^96): the compiler's own f32 division
// (v_div_scale x2, v_rcp, five fma, a mul, v_div_fmas, v_div_fixup) minus the three instructions that only matter outside that domain -- the scaling of operands
// with extreme exponents and the fix-up of zeros / infinities / NaNs.  Same reciprocal, same Newton steps in the same order, hence the same bits as `a / b`
// (rs_selftest_division compares the two on the device).  Regret matching on i32 tables divides a positive regret by the sum of the positive regrets: both are
// integers in [1, 2^34] as floats, the quotient lies in [2^-34, 1].  Explicit fma calls are not contractions: -ffp-contract=off does not touch them.
__device__ __forceinline__ float div_exact_pos(float a, float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float q0 = a * r1;
    const float e1 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(e2, r1, q1);
}

// ---- regret matching: Infoset::get_strategy (infoset.rs:83-102) -----------------------------------
template <int A, typename V>
__device__ __forceinline__ void regret_match(const V (&r)[A], float (&sig)[A]) {
    float norm = 0.0f;
#pragma unroll
    for (int a = 0; a < A; a++)
        if (r[a] > (V)0) norm += (float)r[a];
    const float uni = 1.0f / (float)A;
    if constexpr (sizeof(V) == sizeof(int) && (V)0.5 == (V)0) {   // i32 regrets: positive integers over their positive sum (float tables keep the general division)
#pragma unroll
        for (int a = 0; a < A; a++) sig[a] = (norm > 0.0f) ? ((r[a] > (V)0) ? div_exact_pos((float)r[a], norm) : 0.0f) : uni;
    } else {
#pragma unroll
        for (int a = 0; a < A; a++) sig[a] = (norm > 0.0f) ? ((r[a] > (V)0) ? (float)r[a] / norm : 0.0f) : uni;
    }
}

// ---- the traverser visit for one lane --------------------------------------------------------------
// I32: cfr.rs:413-464 (ARITH = kARITH_CLAMP) or cfr.rs:612-621 (kARITH_WRAP)
template <int A, int ARITH>
__device__ __forceinline__ float visit_i32(int (&r)[A], int (&s)[A], const float (&u)[A], float reach, float scale,
                                           bool rmplus, bool prune) {
    float sig[A];
    regret_match<A, int>(r, sig);
    bool ex[A];
    float util = 0.0f;
#pragma unroll
    for (int a = 0; a < A; a++) {
        ex[a] = !prune || (r[a] > kPruneThresholdD);  // cfr.rs:380
        if (ex[a]) util += u[a] * sig[a];            // cfr.rs:384 / :588
    }
    const bool active = !(reach != reach);           // NaN reach marks a lane whose subtree was pruned above
    const float k = scale * reach;                   // (100.0 * cfr_reach) first
    if (ARITH == kARITH_CLAMP) {
        // `f32 as i64`, 64-bit add, clamp (add_clamp_i64) is about 22 VALU instructions, and a sweep executes 2A of them per traverser node and lane:
        // nearly half of a tree kernel's arithmetic.  Whenever |x| < 2^31 the same result is one conversion (v_cvt_i32_f32 truncates toward zero
        // like the cast) and one SATURATING i32 add (the exact i64 sum clamped to i32 is what saturation means).  The test is made once per visit
        // and for the whole wave, so both paths run unmasked; lanes whose adds are not executed anyway do not vote.
        float dr[A], ds[A];
        bool small = true;
#pragma unroll
        for (int a = 0; a < A; a++) {
            dr[a] = k * (u[a] - util);
            ds[a] = k * sig[a];
            small = small && (!ex[a] || (__builtin_fabsf(dr[a]) < 2147483648.0f && __builtin_fabsf(ds[a]) < 2147483648.0f));   // false for NaN
        }
        small = small || !active;
        if (__builtin_amdgcn_ballot_w64(small) == __builtin_amdgcn_ballot_w64(true)) {
#pragma unroll
            for (int a = 0; a < A; a++) {
                int nr = __builtin_elementwise_add_sat(r[a], (int)dr[a]);
                if (rmplus && nr < 0) nr = 0;
                const int ns = __builtin_elementwise_add_sat(s[a], (int)ds[a]);
                if (ex[a] && active) {
                    r[a] = nr;
                    s[a] = ns;
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < A; a++) {
                if (ex[a] && active) {
                    int nr = add_clamp_i64(r[a], dr[a]);
                    if (rmplus && nr < 0) nr = 0;
                    r[a] = nr;
                    s[a] = add_clamp_i64(s[a], ds[a]);
                }
            }
        }
    } else {
#pragma unroll
        for (int a = 0; a < A; a++) {
            if (ex[a] && active) {
                r[a] = add_wrap_i32(r[a], k * (u[a] - util));
                s[a] = add_wrap_i32(s[a], k * sig[a]);
            }
        }
    }
    return util;
}

// float tables (extension): r += (scale*reach)*(u-util); s += (scale*reach)*sigma
template <int A>
__device__ __forceinline__ float visit_f32(float (&r)[A], float (&s)[A], const float (&u)[A], float reach, float scale,
                                           bool rmplus) {
    float sig[A];
    regret_match<A, float>(r, sig);
    float util = 0.0f;
#pragma unroll
    for (int a = 0; a < A; a++) util += u[a] * sig[a];
    const bool active = !(reach != reach);   // NaN reach: lane not reached in sampled-opponent mode
    const float k = scale * reach;
#pragma unroll
    for (int a = 0; a < A; a++) {
        if (active) {
            float nr = r[a] + k * (u[a] - util);
            if (rmplus && !(nr > 0.0f)) nr = 0.0f;
            r[a] = nr;
            s[a] = s[a] + k * sig[a];
        }
    }
    return util;
}

// f32 tables under DEAL sweeps: several deals of a batch may address one info set, and f32 sums do not commute, so a visit does not add into anything: it hands out its
// two delta vectors, dr = (scale*reach)*(u-util) and ds = (scale*reach)*sigma (0 for a lane that is not on its deal's path), which the sweep stores per deal; after the
// sweep every cell's deltas are summed IN DEAL ORDER from 0.0 and added to the table (k_apply_f32_rows) -- the order the oracle's sequential loop over the deals has.
template <int A>
__device__ __forceinline__ float visit_f32_delta(const float (&r)[A], const float (&u)[A], float reach, float scale, float (&dr)[A], float (&ds)[A]) {
    float sig[A];
    regret_match<A, float>(r, sig);
    float util = 0.0f;
#pragma unroll
    for (int a = 0; a < A; a++) util += u[a] * sig[a];
    const bool active = !(reach != reach);
    const float k = scale * reach;
#pragma unroll
    for (int a = 0; a < A; a++) {
        dr[a] = active ? k * (u[a] - util) : 0.0f;
        ds[a] = active ? k * sig[a] : 0.0f;
    }
    return util;
}
template <int A>
__device__ __forceinline__ void lanes_visit_delta(const float (&r)[A][kVecD], const float (&u)[A][kVecD], const float (&reach)[kVecD], float scale, float (&dr)[A][kVecD],
                                                  float (&ds)[A][kVecD], float (&dest)[kVecD]) {
#pragma unroll
    for (int j = 0; j < kVecD; j++) {
        float rl[A], ul[A], d1[A], d2[A];
#pragma unroll
        for (int a = 0; a < A; a++) { rl[a] = r[a][j]; ul[a] = u[a][j]; }
        dest[j] = visit_f32_delta<A>(rl, ul, reach[j], scale, d1, d2);
#pragma unroll
        for (int a = 0; a < A; a++) { dr[a][j] = d1[a]; ds[a][j] = d2[a]; }
    }
}

// ---- opponent sampling: mccfr's `WeightedIndex::new(&strategy)` + `dist.sample(rng)` (cfr.rs:471-472) ------------
// rand 0.7 restated over supplied raw bits (the reference's SmallRng is seeded from thread_rng and is not
// reproducible): cumulative[i] = w0+..+wi (f32, sequential), total = sum; u01 = (bits >> 9) * 2^-23;
// chosen = u01 * total + 0; index = number of cumulative[i] <= chosen.  The bits are a counter-based hash of
// (sweep seed, ActionNode.index, lane), identical on the GPU and in the CPU oracle.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ unsigned sample_bits(unsigned long long seed, unsigned node_index, unsigned long long lane) {
    // A 32-bit counter hash (round 2; round 1 ran splitmix64 here: two 64-bit multiplies, about 40 vector instructions per opponent node and lane, a fifth of a deal
    // kernel's VALU work).  The node folds into a wave-uniform word (scalar ALU), the lane into one add and one 32-bit multiply that the compiler shares between all
    // opponent nodes of a kernel; the finisher is the two-multiply "lowbias32" mixer.  23 of the 32 bits reach the sampler (u01 = (bits >> 9) * 2^-23).
    // The upper seed half is ADDED to the lane before the multiply (round 3): with the seed only xor-ed in, two (node, lane) pairs whose 32-bit keys collide drew the same
    // bits in EVERY sweep whatever the seed (at 4 M deals x 700 nodes most keys have such a partner); (lane + s) * M is not xor-linear in s, so which pairs collide now
    // changes with every sweep seed.
    const unsigned s_lo = (unsigned)seed, s_hi = (unsigned)(seed >> 32);
    const unsigned n_mix = (node_index + 1u) * 0xC2B2AE35u;
    const unsigned l_mix = (((unsigned)lane + s_hi) * 0x9E3779B9u) ^ ((unsigned)(lane >> 32) * 0x27D4EB2Fu);
    unsigned x = s_lo ^ n_mix ^ l_mix;
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned long long sweep_seed(unsigned long long base_seed, unsigned long long call_index) {
    return splitmix64(base_seed + call_index * 0x632BE59BD9B4E019ull);
}
template <int A>
__device__ __forceinline__ int weighted_index(const float (&w)[A], unsigned bits) {
    float cumulative[A];
    float total = w[0];
#pragma unroll
    for (int i = 1; i < A; i++) {
        cumulative[i - 1] = total;
        total += w[i];
    }
    const float u01 = (float)(bits >> 9) * 1.1920928955078125e-07f;
    const float chosen = u01 * total + 0.0f;
    int idx = 0;
#pragma unroll
    for (int i = 0; i < A - 1; i++)
        if (cumulative[i] <= chosen) idx = i + 1;
    return idx;
}
// sampled action of 4 lanes of one opponent node
template <int A>
__device__ __forceinline__ void lanes_sample(const float (&sig)[A][kVecD], unsigned long long seed, unsigned node_index,
                                             unsigned v, int (&a_s)[kVecD], unsigned lane_base = 0) {
    // lane_base: data-parallel deal batches -- the global index of this rank's first deal, so that a deal draws the same bits
    // whichever rank owns it
#pragma unroll
    for (int j = 0; j < kVecD; j++) {
        float w[A];
#pragma unroll
        for (int a = 0; a < A; a++) w[a] = sig[a][j];
        a_s[j] = weighted_index<A>(w, sample_bits(seed, node_index, (unsigned long long)lane_base + (unsigned long long)v * kVecD + j));
    }
}

// sparse deal sweeps: the 4 lanes of a thread are entries of a compacted list of live deals; the hash must see the deal itself
template <int A>
__device__ __forceinline__ void lanes_sample_ids(const float (&sig)[A][kVecD], unsigned long long seed, unsigned node_index,
                                                 const unsigned (&ids)[kVecD], int (&a_s)[kVecD], unsigned lane_base) {
#pragma unroll
    for (int j = 0; j < kVecD; j++) {
        float w[A];
#pragma unroll
        for (int a = 0; a < A; a++) w[a] = sig[a][j];
        a_s[j] = weighted_index<A>(w, sample_bits(seed, node_index, (unsigned long long)lane_base + ids[j]));
    }
}
// per-deal rows read / written through the list (ids[j] valid only where real[j])
__device__ __forceinline__ void gather_f32_ids(const float *base, const unsigned (&ids)[kVecD], const bool (&real)[kVecD], float (&out)[kVecD], float pad) {
    const RS_GLOBAL float *p = as_global<float>(base);
#pragma unroll
    for (int j = 0; j < kVecD; j++) out[j] = real[j] ? p[ids[j]] : pad;
}
__device__ __forceinline__ void gather_u32_ids(const unsigned *base, const unsigned (&ids)[kVecD], const bool (&real)[kVecD], unsigned (&out)[kVecD]) {
    const RS_GLOBAL unsigned *p = as_global<unsigned>(base);
#pragma unroll
    for (int j = 0; j < kVecD; j++) out[j] = real[j] ? p[ids[j]] : 0u;
}
// the packed per-deal inputs of a round (k_pack_attr): {cluster id of player 0, of player 1, leaf value bits, prune flag} in ONE 16-byte gather per live deal
__device__ __forceinline__ void gather_attr(const void *base, const unsigned (&ids)[kVecD], const bool (&real)[kVecD], unsigned (&c0)[kVecD], unsigned (&c1)[kVecD],
                                            float (&leaf)[kVecD], unsigned (&flag)[kVecD]) {
    const RS_GLOBAL u32x4 *p = as_global<u32x4>((const unsigned *)base);
#pragma unroll
    for (int j = 0; j < kVecD; j++) {
        const u32x4 r = real[j] ? p[ids[j]] : u32x4{0u, 0u, 0u, 0u};
        c0[j] = r.x;
        c1[j] = r.y;
        leaf[j] = __uint_as_float(r.z);
        flag[j] = r.w;
    }
}
// ---- ordered deal sweeps (rs_kernel_forms.deal_order) ---------------------------------------------------------------------------------------
// The sweep of traverser p walks the batch in the order of p's cluster id on the LAST betting round (k_order_* in rs_kernels.hip sort the deals once per sweep),
// and a "deal" inside the sweep is its RANK in that order.  The per-deal inputs of all rounds travel as ONE 32-byte record per rank,
//   word 0..3 = cluster ids of rounds 0 and 1 ([2 * round + player]),  word 4, 5 = cluster ids of round 2,  word 6 = leaf value bits,
//   word 7 = (original deal id << 1) | prune flag
// so the last round's kernels -- most of a sweep's walks -- fetch the upper 16 bytes only.  The original id is what the opponent-sampling hash sees: the draw of a
// deal does not depend on where the sort put it (and the oracle never sorts).
template <int ROUND>
__device__ __forceinline__ void load_arec(const void *base, const unsigned (&idx)[kVecD], const bool (&ok)[kVecD], unsigned (&c0)[kVecD], unsigned (&c1)[kVecD],
                                          float (&leaf)[kVecD], unsigned (&flag)[kVecD], unsigned (&oid)[kVecD]) {
    const RS_GLOBAL u32x4 *p = as_global<u32x4>((const unsigned *)base);
#pragma unroll
    for (int j = 0; j < kVecD; j++) {
        const u32x4 hi = ok[j] ? p[2 * (size_t)idx[j] + 1] : u32x4{0u, 0u, 0u, 0u};
        if (ROUND == 2) {
            c0[j] = hi.x;
            c1[j] = hi.y;
        } else {
            const u32x4 lo = ok[j] ? p[2 * (size_t)idx[j]] : u32x4{0u, 0u, 0u, 0u};
            c0[j] = ROUND == 0 ? lo.x : lo.z;
            c1[j] = ROUND == 0 ? lo.y : lo.w;
        }
        leaf[j] = __uint_as_float(hi.z);
        flag[j] = hi.w & 1u;
        oid[j] = hi.w >> 1;
    }
}

// In an ordered sweep the deals a wave walks in the last round come in RUNS of equal traverser cluster (the batch is sorted by it; a list keeps the order of its source inside a
// tile of the compaction): the lanes of a run add into the SAME table cells.  Instead of an LDS tile per traverser node, or a row of deltas per node and deal summed by a later
// pass, the wave sums every delta along the runs inside each row of 16 lanes -- a segmented inclusive scan by DPP row shifts, four steps, the same four run masks for every
// value of every node of the walk -- and the last lane of every (row, run) piece issues ONE global atomic per node, action and array.  Pieces of one run in different rows, waves
// or workgroups meet in the atomics.  Integer adds commute, so the result equals the tile form's, the delta rows', and the oracle's, bit for bit.
struct Seg {
    bool m1, m2, m4, m8;   // no run starts among this lane and the 0 / 1 / 3 / 7 lanes before it (inside its row of 16): the lane 1 / 2 / 4 / 8 back belongs to the same run
    bool tail;             // the last lane of its piece: it holds the piece's sums after the scan
};
__device__ __forceinline__ Seg seg_make(unsigned key) {
    Seg sg;
    const unsigned prev = (unsigned)__builtin_amdgcn_update_dpp((int)~key, (int)key, 0x111, 0xf, 0xf, false);   // row_shr:1; the first lane of a row keeps ~key: a run starts there
    const unsigned next = (unsigned)__builtin_amdgcn_update_dpp((int)~key, (int)key, 0x101, 0xf, 0xf, false);   // row_shl:1; the last lane of a row keeps ~key
    const int f1 = prev == key ? 1 : 0;
    const int f2 = f1 & __builtin_amdgcn_update_dpp(0, f1, 0x111, 0xf, 0xf, true);
    const int f4 = f2 & __builtin_amdgcn_update_dpp(0, f2, 0x112, 0xf, 0xf, true);
    const int f8 = f4 & __builtin_amdgcn_update_dpp(0, f4, 0x114, 0xf, 0xf, true);
    sg.m1 = f1 != 0;
    sg.m2 = f2 != 0;
    sg.m4 = f4 != 0;
    sg.m8 = f8 != 0;
    sg.tail = next != key;
    return sg;
}
__device__ __forceinline__ int seg_scan_i32(int x, const Seg &sg) {   // sum of x over the lanes of this lane's run piece up to and including this lane
    int y = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
    x += sg.m1 ? y : 0;
    y = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
    x += sg.m2 ? y : 0;
    y = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
    x += sg.m4 ? y : 0;
    y = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
    x += sg.m8 ? y : 0;
    return x;
}
template <int A>
__device__ __forceinline__ void seg_add(void *dreg, void *dssm, unsigned tpitch, const Seg &sg, unsigned key, const int (&r)[A][kVecD], const int (&q)[A][kVecD],
                                        const int (&s)[A][kVecD], const int (&t)[A][kVecD]) {
    static_assert(kVecD == 1, "segmented sums take one deal per lane");
    int d[2 * A];
    bool any = false;
#pragma unroll
    for (int a = 0; a < A; a++) {
        d[a] = (int)((unsigned)r[a][0] - (unsigned)q[a][0]);
        d[A + a] = (int)((unsigned)s[a][0] - (unsigned)t[a][0]);
        any = any || d[a] != 0 || d[A + a] != 0;
    }
    if (__builtin_amdgcn_ballot_w64(any) == 0ull) return;   // nobody's deal visited the node
    RS_GLOBAL int *pr = as_global<int>(dreg), *ps = as_global<int>(dssm);
#pragma unroll
    for (int a = 0; a < A; a++) {
        const int xr = seg_scan_i32(d[a], sg), xs = seg_scan_i32(d[A + a], sg);
        if (sg.tail && xr != 0) __hip_atomic_fetch_add(pr + (size_t)a * tpitch + key, xr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sg.tail && xs != 0) __hip_atomic_fetch_add(ps + (size_t)a * tpitch + key, xs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ void scatter_f32_ids(float *base, const unsigned (&ids)[kVecD], const bool (&real)[kVecD], const float (&in)[kVecD]) {
    RS_GLOBAL float *p = as_global<float>(base);
#pragma unroll
    for (int j = 0; j < kVecD; j++)
        if (real[j]) p[ids[j]] = in[j];
}

// ---- 4-lane wrappers used by the tree-specialised (hipRTC) kernels ------------------------------------------
template <int A, int DT>
__device__ __forceinline__ void lanes_regret_match(const typename Row<DT>::val (&r)[A][kVecD], float (&sig)[A][kVecD]) {
    using V = typename Row<DT>::val;
#pragma unroll
    for (int j = 0; j < kVecD; j++) {
        V rl[A];
        float sg[A];
#pragma unroll
        for (int a = 0; a < A; a++) rl[a] = r[a][j];
        regret_match<A, V>(rl, sg);
#pragma unroll
        for (int a = 0; a < A; a++) sig[a][j] = sg[a];
    }
}

// one traverser visit of 4 lanes: updates r / s in place, writes the node utility to dest
template <int A, int DT, int ARITH>
__device__ __forceinline__ void lanes_visit(typename Row<DT>::val (&r)[A][kVecD], typename Row<DT>::val (&s)[A][kVecD],
                                            const float (&u)[A][kVecD], const float (&reach)[kVecD], float scale, bool rmplus,
                                            float (&dest)[kVecD]) {
    using V = typename Row<DT>::val;
#pragma unroll
    for (int j = 0; j < kVecD; j++) {
        V rl[A], sl[A];
        float ul[A];
#pragma unroll
        for (int a = 0; a < A; a++) { rl[a] = r[a][j]; sl[a] = s[a][j]; ul[a] = u[a][j]; }
        if constexpr (DT == kDT_I32) dest[j] = visit_i32<A, ARITH>(rl, sl, ul, reach[j], scale, rmplus, false);
        else dest[j] = visit_f32<A>(rl, sl, ul, reach[j], scale, rmplus);
#pragma unroll
        for (int a = 0; a < A; a++) { r[a][j] = rl[a]; s[a][j] = sl[a]; }
    }
}

// the pruned visit (cfr.rs:379-386, :419-441) for the lanes whose deal is traversed with prune = true
template <int A, int DT, int ARITH>
__device__ __forceinline__ void lanes_visit_prune(typename Row<DT>::val (&r)[A][kVecD], typename Row<DT>::val (&s)[A][kVecD],
                                                  const float (&u)[A][kVecD], const float (&reach)[kVecD], float scale, bool rmplus,
                                                  const bool (&prune)[kVecD], float (&dest)[kVecD]) {
    static_assert(DT == kDT_I32, "pruning compares i32 regrets with the threshold (cfr.rs:352)");
#pragma unroll
    for (int j = 0; j < kVecD; j++) {
        int rl[A], sl[A];
        float ul[A];
#pragma unroll
        for (int a = 0; a < A; a++) { rl[a] = r[a][j]; sl[a] = s[a][j]; ul[a] = u[a][j]; }
        dest[j] = visit_i32<A, ARITH>(rl, sl, ul, reach[j], scale, rmplus, prune[j]);
#pragma unroll
        for (int a = 0; a < A; a++) { r[a][j] = rl[a]; s[a][j] = sl[a]; }
    }
}
// bit a*kVecD+j: lane j explores action a (always when its deal is not pruned)
template <int A>
__device__ __forceinline__ unsigned lanes_explored(const int (&r)[A][kVecD], const bool (&prune)[kVecD]) {
    unsigned m = 0;
#pragma unroll
    for (int a = 0; a < A; a++)
#pragma unroll
        for (int j = 0; j < kVecD; j++) m |= (!prune[j] || r[a][j] > kPruneThresholdD) ? 1u << (a * kVecD + j) : 0u;
    return m;
}

}  // namespace rs
