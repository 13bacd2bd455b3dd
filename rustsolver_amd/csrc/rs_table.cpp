// rs_table.cpp -- the info-set table in HBM and the per-node entry points of the C ABI.
// Replaces `InfosetTable = Vec<Vec<Infoset>>` (infoset.rs:6,63-67) with two SoA device arrays
// per action node, regrets[A][pitch] and strategy_sum[A][pitch], lane-major.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "rs_internal.hpp"

namespace rs {

static thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}
int hip_fail(hipError_t e, const char *what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? RS_ERR_OOM : RS_ERR_HIP;
}

#define RS_HIP(call, what)                                  \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return rs::hip_fail(e_, what); \
    } while (0)

// ---- profiling ------------------------------------------------------------------------------------
static hipEvent_t prof_event(rs_table *t) {
    if (!t->prof.pool.empty()) {
        hipEvent_t e = t->prof.pool.back();
        t->prof.pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void prof_begin(rs_table *t, int kind, double bytes) {
    if (!t->prof.on) return;
    Profile::Pending p{prof_event(t), prof_event(t), kind, bytes};
    (void)hipEventRecord(p.a, t->stream);
    t->prof.pending.push_back(p);
}
void prof_end(rs_table *t) {
    if (!t->prof.on) return;
    (void)hipEventRecord(t->prof.pending.back().b, t->stream);
}
static void prof_collect(rs_table *t) {
    for (Profile::Pending &p : t->prof.pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            t->prof.acc.launches[p.kind] += 1;
            t->prof.acc.ms[p.kind] += ms;
            t->prof.acc.algo_bytes[p.kind] += p.bytes;
        }
        t->prof.pool.push_back(p.a);
        t->prof.pool.push_back(p.b);
    }
    t->prof.pending.clear();
}

// algorithmic bytes of one traverser visit of a node: table rows are read and written,
// utilities / reach / node util count only when they are real buffers
double algo_bytes_update(const rs_table *t, int node, int n_buf_children, bool has_reach, bool has_out) {
    const double lanes = double(t->nodes[node].n_boards) * t->nodes[node].n_clusters;
    const double es = double(elem_size(t->dtype));
    const double A = t->nodes[node].n_actions;
    return lanes * (A * 4.0 * es + 4.0 * n_buf_children + (has_reach ? 4.0 : 0.0) + (has_out ? 4.0 : 0.0));
}

// A training loop may have made a solver's kept shadow records the working copy (rs_solver.cpp solver_kept_primary): the table's rows of those nodes are stale until they are
// written back.  Every entry point that reads or writes table contents settles that first -- rows written back, the records no longer the working copy (the loop goes on with table
// and records both updated, as outside a loop) -- so that nothing ever reads stale rows or loses what a loop has trained.
int table_settle(rs_table *t) {
    if (!t) return RS_OK;
    for (rs_solver *s : t->solvers)
        if (solver_is_primary(s))
            if (int rc = solver_kept_primary(s, false)) return rc;
    return RS_OK;
}

static int check_node(const rs_table *t, int node, const char *fn) {
    if (!t) return fail(RS_ERR_INVALID, std::string(fn) + ": table is NULL");
    if (node < 0 || node >= int(t->nodes.size()))
        return fail(RS_ERR_OOB, std::string(fn) + ": node index " + std::to_string(node) + " out of bounds (len " +
                                    std::to_string(t->nodes.size()) + ")");
    return RS_OK;
}

// host f32 <-> binary16 (RNE), used only by the upload / download paths of RS_F16 tables
static uint16_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
static float f16_bits_to_f32(uint16_t b) {
    _Float16 h;
    std::memcpy(&h, &b, 2);
    return (float)h;
}

// Big lane tables keep the rows of a node INTERLEAVED in tiles: block = [pitch / T][A][T] instead of [A][pitch] (T = 16 384 lanes).  A sweep streams
// every row of every node at once; with the rows tens of MB apart that is ~95 streams, each in its own DRAM pages and TLB entries, and the same bytes
// move 22 - 35 % slower than when the rows of a node sit next to each other tile by tile (tools/stream_probe.cpp: 4.43 TB/s rows apart, 5.4 - 5.96 TB/s
// grouped per node on the same card).  Nodes below kTileMinLanes lanes keep the plain block (T = pitch): everything a deal sweep gathers from.
// RS_TABLE_TILE_LANES (a power of two >= 64; 0 = never tile) and RS_TABLE_TILE_MIN_LANES override, read at every table creation (tests tile small tables).
static constexpr size_t kTileLanes = size_t(1) << 14, kTileMinLanes = size_t(1) << 20;

// copy between host [rows][row_len] (host element) and lanes [col0, col0 + row_len) of the first `rows` rows of `node`'s block at d_base.  dir: 0 = upload,
// 1 = download.  One 2-D copy per tile the range touches (rows are T elements apart inside a tile).
// raw: the host side holds the table's own element type (binary16 bits for RS_F16 tables: checkpoints), no conversion.
static int copy_rows(rs_table *t, void *d_base, int node, size_t col0, void *host, size_t rows, size_t row_len, int dir, bool raw = false) {
    const size_t es = elem_size(t->dtype), T = t->tile[size_t(node)], A = t->nodes[size_t(node)].n_actions;
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    const bool f16 = t->dtype == RS_F16 && !raw;
    std::vector<uint16_t> tmp;   // F16: convert on the host through a temporary
    char *h = (char *)host;
    const size_t hes = f16 ? 2 : es;
    if (f16) {
        tmp.resize(rows * row_len);
        h = (char *)tmp.data();
        if (dir == 0) {
            const float *src = (const float *)host;
            for (size_t i = 0; i < rows * row_len; ++i) tmp[i] = f32_to_f16_bits(src[i]);
        }
    }
    for (size_t lo = col0; lo < col0 + row_len;) {
        const size_t tile = lo / T, w = lo % T, len = std::min(T - w, col0 + row_len - lo);
        char *d = (char *)d_base + ((tile * A) * T + w) * es;
        char *hp = h + (lo - col0) * hes;
        if (dir == 0) RS_HIP(hipMemcpy2DAsync(d, T * es, hp, row_len * hes, len * es, rows, hipMemcpyHostToDevice, t->stream), "hipMemcpy2DAsync(upload)");
        else RS_HIP(hipMemcpy2DAsync(hp, row_len * hes, d, T * es, len * es, rows, hipMemcpyDeviceToHost, t->stream), "hipMemcpy2DAsync(download)");
        lo += len;
    }
    RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
    if (f16 && dir == 1) {
        float *dst = (float *)host;
        for (size_t i = 0; i < rows * row_len; ++i) dst[i] = f16_bits_to_f32(tmp[i]);
    }
    return RS_OK;
}

// checkpoints (rs_abstraction.cpp): one array of one node as unpadded [A][lanes] rows of the TABLE's element type, whatever the in-memory tiling
int table_copy_node_raw(rs_table *t, int node, int which, void *host, int dir) {
    const rs_node_desc &nd = t->nodes[size_t(node)];
    if (nd.n_actions == 0) return RS_OK;
    return copy_rows(t, which == 0 ? t->regrets_ptr(node) : t->ssum_ptr(node), node, 0, host, nd.n_actions, size_t(nd.n_boards) * nd.n_clusters, dir, true);
}

}  // namespace rs

using namespace rs;

extern "C" {

const char *rs_last_error(void) { return g_last_error.c_str(); }
int rs_abi_version(void) { return RS_ABI_VERSION; }

int rs_device_count(int *out) {
    if (!out) return fail(RS_ERR_INVALID, "rs_device_count: out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out = 0;
        return hip_fail(e, "hipGetDeviceCount");
    }
    *out = n;
    return RS_OK;
}

// ---- create / destroy --------------------------------------------------------------------------------
int rs_table_create(const rs_node_desc *nodes, int n_nodes, int dtype, int device, rs_table **out) {
    return rs_table_create_with(nodes, n_nodes, dtype, device, nullptr, out);
}

int rs_table_create_with(const rs_node_desc *nodes, int n_nodes, int dtype, int device, const rs_table_params *params, rs_table **out) {
    if (!nodes || !out || n_nodes <= 0) return fail(RS_ERR_INVALID, "rs_table_create: bad argument");
    if (dtype != RS_I32 && dtype != RS_F32 && dtype != RS_F16) return fail(RS_ERR_INVALID, "rs_table_create: bad dtype");
    for (int i = 0; i < n_nodes; ++i) {
        const rs_node_desc &nd = nodes[i];
        // n_actions == 0 is legal: state.rs:125-157 can return no valid action (a short all-in raise that stays below the
        // bet); such a node owns no cells, is worth 0 (cfr.rs:571-589 with empty loops) and is never launched
        if (nd.n_actions > RS_MAX_ACTIONS || nd.n_clusters < 1 || nd.n_boards < 1 || nd.player > 1 || nd.round_idx >= RS_MAX_ROUNDS)
            return fail(RS_ERR_INVALID, "rs_table_create: bad descriptor for node " + std::to_string(i));
        if (size_t(nd.n_boards) * nd.n_clusters * nd.n_actions > (size_t(1) << 31))
            return fail(RS_ERR_UNSUPPORTED, "rs_table_create: a node exceeds 2^31 cells; shard the board axis");
    }
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0)
        return fail(RS_ERR_HIP, std::string("rs_table_create: no usable HIP device (") +
                                    (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                                    "); this library has no CPU fallback");
    if (device < 0 || device >= n_dev) return fail(RS_ERR_INVALID, "rs_table_create: bad device ordinal");
    RS_HIP(hipSetDevice(device), "hipSetDevice");

    rs_table *t = new (std::nothrow) rs_table();
    if (!t) return fail(RS_ERR_OOM, "rs_table_create: out of host memory");
    t->device = device;
    t->dtype = dtype;
    t->nodes.assign(nodes, nodes + n_nodes);
    t->pitch.resize(n_nodes);
    t->cell_off.resize(n_nodes);
    size_t tile_lanes = kTileLanes, tile_min = kTileMinLanes;
    if (params && params->tile_lanes) tile_lanes = (params->tile_lanes >= uint32_t(kLanePad) && (params->tile_lanes & (params->tile_lanes - 1)) == 0) ? size_t(params->tile_lanes) : 0;
    if (params && params->tile_min_lanes) tile_min = params->tile_min_lanes;
    const Knobs knobs = knobs_resolve(nullptr);   // tests force small tiles on small tables
    if (knobs.tile_lanes != kUnset) {   // RS_TABLE_TILE_LANES = T: every node wider than T lanes is tiled in T-lane tiles
        const long v = knobs.tile_lanes;
        tile_lanes = (v >= long(kLanePad) && (v & (v - 1)) == 0) ? size_t(v) : 0;   // anything else: never tile
        tile_min = tile_lanes + 1;
    }
    t->tile.resize(size_t(n_nodes));
    size_t off = 0;
    for (int i = 0; i < n_nodes; ++i) {
        const size_t lanes = size_t(nodes[i].n_boards) * nodes[i].n_clusters;
        const bool tiled = tile_lanes != 0 && lanes >= tile_min && lanes > tile_lanes;
        t->pitch[i] = round_up(lanes, tiled ? tile_lanes : size_t(kLanePad));
        t->tile[i] = tiled ? tile_lanes : t->pitch[i];
        t->cell_off[i] = off;
        off += t->pitch[i] * nodes[i].n_actions;
    }
    t->n_cells = off;
    const size_t bytes = off * elem_size(dtype);
    hipError_t er;
    if ((er = hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking)) != hipSuccess ||
        (er = hipMalloc(&t->d_regrets, bytes)) != hipSuccess || (er = hipMalloc(&t->d_ssum, bytes)) != hipSuccess ||
        (er = hipMemsetAsync(t->d_regrets, 0, bytes, t->stream)) != hipSuccess ||   // Infoset::init zero fill
        (er = hipMemsetAsync(t->d_ssum, 0, bytes, t->stream)) != hipSuccess ||
        (er = hipStreamSynchronize(t->stream)) != hipSuccess) {
        int rc = hip_fail(er, "rs_table_create: device allocation");
        rs_table_destroy(t);
        return rc;
    }
    *out = t;
    return RS_OK;
}

int rs_create_infosets(const rs_tree *tree, const uint32_t n_clusters[RS_MAX_ROUNDS][RS_MAX_PLAYERS],
                       const uint32_t n_boards[RS_MAX_ROUNDS], int dtype, int device, rs_table **out) {
    if (!tree || !n_clusters || !n_boards || !out) return fail(RS_ERR_INVALID, "rs_create_infosets: NULL argument");
    std::vector<rs_node_desc> descs(tree->n_action_nodes);
    // create_infosets_rec (infoset.rs:20-49): one row per Action node, addressed by an.index
    for (const rs_tree_node &nd : tree->nodes) {
        if (nd.kind != RS_NODE_ACTION) continue;
        rs_node_desc d{};
        d.n_actions = uint32_t(nd.n_children);                  // node.children.len(), infoset.rs:33
        d.n_clusters = n_clusters[nd.round_idx][nd.player];     // card_abs[round_idx].get_size(player), infoset.rs:28-32
        d.n_boards = n_boards[nd.round_idx];
        d.player = nd.player;
        d.round_idx = nd.round_idx;
        descs[nd.index] = d;
    }
    return rs_table_create(descs.data(), int(descs.size()), dtype, device, out);
}

void rs_table_destroy(rs_table *t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    while (!t->solvers.empty()) solver_release_device(t->solvers.back());   // they stay valid handles, but inert
    for (Profile::Pending &p : t->prof.pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (hipEvent_t e : t->prof.pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : t->prof.marks) (void)hipEventDestroy(e);
    if (t->d_regrets) (void)hipFree(t->d_regrets);
    if (t->d_ssum) (void)hipFree(t->d_ssum);
    if (t->d_snap_regrets) (void)hipFree(t->d_snap_regrets);
    if (t->d_snap_ssum) (void)hipFree(t->d_snap_ssum);
    if (t->d_job) (void)hipFree(t->d_job);
    if (t->d_query) (void)hipFree(t->d_query);
    if (t->d_err_sink) (void)hipFree(t->d_err_sink);
    if (t->d_km_scratch) (void)hipFree(t->d_km_scratch);
    if (t->d_dregrets) (void)hipFree(t->d_dregrets);
    if (t->d_dssum) (void)hipFree(t->d_dssum);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
}

// ---- introspection ---------------------------------------------------------------------------------------
int rs_table_n_nodes(const rs_table *t) { return t ? int(t->nodes.size()) : RS_ERR_INVALID; }
int rs_table_dtype(const rs_table *t) { return t ? t->dtype : RS_ERR_INVALID; }
int rs_table_device(const rs_table *t) { return t ? t->device : RS_ERR_INVALID; }
int rs_table_node_desc(const rs_table *t, int node, rs_node_desc *out) {
    if (int rc = check_node(t, node, "rs_table_node_desc")) return rc;
    if (!out) return fail(RS_ERR_INVALID, "rs_table_node_desc: out is NULL");
    *out = t->nodes[node];
    return RS_OK;
}
size_t rs_table_lane_pitch(const rs_table *t, int node) {
    if (check_node(t, node, "rs_table_lane_pitch")) return 0;
    return t->pitch[node];
}
size_t rs_table_cells(const rs_table *t) { return t ? t->n_cells : 0; }
size_t rs_table_cell_offset(const rs_table *t, int node) {
    if (check_node(t, node, "rs_table_cell_offset")) return 0;
    return t->cell_off[node];
}
size_t rs_table_bytes(const rs_table *t) { return t ? 2 * t->n_cells * elem_size(t->dtype) : 0; }
void *rs_stream(rs_table *t) { return t ? (void *)t->stream : nullptr; }

// ---- host <-> device ------------------------------------------------------------------------------------------
static int board_copy(rs_table *t, int node, int board, void *regrets, void *ssum, int dir, const char *fn) {
    if (int rc = check_node(t, node, fn)) return rc;
    if (int rc = table_settle(t)) return rc;
    const rs_node_desc &nd = t->nodes[node];
    if (nd.n_actions == 0) return RS_OK;
    if (board < 0 || uint32_t(board) >= nd.n_boards) return fail(RS_ERR_OOB, std::string(fn) + ": board out of bounds");
    const size_t col0 = size_t(board) * nd.n_clusters;
    if (regrets)
        if (int rc = copy_rows(t, t->regrets_ptr(node), node, col0, regrets, nd.n_actions, nd.n_clusters, dir)) return rc;
    if (ssum)
        if (int rc = copy_rows(t, t->ssum_ptr(node), node, col0, ssum, nd.n_actions, nd.n_clusters, dir)) return rc;
    return RS_OK;
}
int rs_table_upload(rs_table *t, int node, int board, const void *regrets, const void *ssum) {
    if (t) ++t->epoch;
    return board_copy(t, node, board, (void *)regrets, (void *)ssum, 0, "rs_table_upload");
}
int rs_table_download(rs_table *t, int node, int board, void *regrets, void *ssum) {
    return board_copy(t, node, board, regrets, ssum, 1, "rs_table_download");
}
static int node_copy(rs_table *t, int node, void *regrets, void *ssum, int dir, const char *fn) {
    if (int rc = check_node(t, node, fn)) return rc;
    if (int rc = table_settle(t)) return rc;
    const rs_node_desc &nd = t->nodes[node];
    if (nd.n_actions == 0) return RS_OK;   // nothing to copy
    const size_t lanes = size_t(nd.n_boards) * nd.n_clusters;
    if (regrets)
        if (int rc = copy_rows(t, t->regrets_ptr(node), node, 0, regrets, nd.n_actions, lanes, dir)) return rc;
    if (ssum)
        if (int rc = copy_rows(t, t->ssum_ptr(node), node, 0, ssum, nd.n_actions, lanes, dir)) return rc;
    return RS_OK;
}
int rs_table_upload_node(rs_table *t, int node, const void *regrets, const void *ssum) {
    if (t) ++t->epoch;
    return node_copy(t, node, (void *)regrets, (void *)ssum, 0, "rs_table_upload_node");
}
int rs_table_download_node(rs_table *t, int node, void *regrets, void *ssum) {
    return node_copy(t, node, regrets, ssum, 1, "rs_table_download_node");
}

// get-infoset: &self.infosets[an.index][cluster_idx] (cfr.rs:375)
static int infoset_copy(rs_table *t, int node, int board, int cluster, void *regrets, void *ssum, int dir, const char *fn) {
    if (int rc = check_node(t, node, fn)) return rc;
    if (int rc = table_settle(t)) return rc;
    const rs_node_desc &nd = t->nodes[node];
    if (nd.n_actions == 0) return RS_OK;
    if (board < 0 || uint32_t(board) >= nd.n_boards || cluster < 0 || uint32_t(cluster) >= nd.n_clusters)
        return fail(RS_ERR_OOB, std::string(fn) + ": index out of bounds: the len is " + std::to_string(nd.n_clusters) +
                                    " but the index is " + std::to_string(cluster));
    const size_t col0 = size_t(board) * nd.n_clusters + size_t(cluster);
    if (regrets)
        if (int rc = copy_rows(t, t->regrets_ptr(node), node, col0, regrets, nd.n_actions, 1, dir)) return rc;
    if (ssum)
        if (int rc = copy_rows(t, t->ssum_ptr(node), node, col0, ssum, nd.n_actions, 1, dir)) return rc;
    return RS_OK;
}
int rs_get_infoset(rs_table *t, int node, int board, int cluster, void *regrets, void *ssum) {
    return infoset_copy(t, node, board, cluster, regrets, ssum, 1, "rs_get_infoset");
}
int rs_set_infoset(rs_table *t, int node, int board, int cluster, const void *regrets, const void *ssum) {
    if (t) ++t->epoch;
    return infoset_copy(t, node, board, cluster, (void *)regrets, (void *)ssum, 0, "rs_set_infoset");
}

// get-infoset for a batch of lanes of one node: one gather kernel per array instead of n strided copies
int rs_get_infosets(rs_table *t, int node, const uint32_t *lanes, size_t n, void *regrets, void *ssum) {
    if (int rc = check_node(t, node, "rs_get_infosets")) return rc;
    if (int rc = table_settle(t)) return rc;
    if (!lanes && n) return fail(RS_ERR_INVALID, "rs_get_infosets: lanes is NULL");
    const rs_node_desc &nd = t->nodes[node];
    if (nd.n_actions == 0 || n == 0) return RS_OK;
    const size_t n_lanes = size_t(nd.n_boards) * nd.n_clusters;
    for (size_t k = 0; k < n; ++k)
        if (lanes[k] >= n_lanes)
            return fail(RS_ERR_OOB, "rs_get_infosets: index out of bounds: the len is " + std::to_string(n_lanes) + " but the index is " + std::to_string(lanes[k]));
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    const size_t es = elem_size(t->dtype), A = nd.n_actions;
    uint32_t *d_lanes = nullptr;
    void *d_out = nullptr;
    hipError_t e = hipMalloc((void **)&d_lanes, n * 4);
    if (e == hipSuccess) e = hipMalloc(&d_out, A * n * es);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lanes, lanes, n * 4, hipMemcpyHostToDevice, t->stream);
    std::vector<uint16_t> tmp(t->dtype == RS_F16 ? A * n : 0);
    for (int which = 0; which < 2 && e == hipSuccess; ++which) {
        void *host = which == 0 ? regrets : ssum;
        if (!host) continue;
        e = launch_gather_lanes(which == 0 ? t->regrets_ptr(node) : t->ssum_ptr(node), d_lanes, n, uint32_t(A), t->tile[size_t(node)], es, d_out, t->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(t->dtype == RS_F16 ? (void *)tmp.data() : host, d_out, A * n * es, hipMemcpyDeviceToHost, t->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
        if (e == hipSuccess && t->dtype == RS_F16)
            for (size_t i = 0; i < A * n; ++i) ((float *)host)[i] = f16_bits_to_f32(tmp[i]);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (d_lanes) (void)hipFree(d_lanes);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return hip_fail(e, "rs_get_infosets");
    return RS_OK;
}

// div_exact_pos (rs_device.hpp: the division of regret matching on i32 tables without the scale / fix-up instructions) against the compiler's `a / b` on n hashed
// pairs from its domain: *mismatches must come back 0; first_bad (may be NULL) = the first differing (a, b)
int rs_selftest_division(rs_table *t, size_t n, uint64_t seed, uint64_t *mismatches, float *first_bad) {
    if (!t || !mismatches) return fail(RS_ERR_INVALID, "rs_selftest_division: NULL argument");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    char *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, 16);
    if (e == hipSuccess) e = hipMemsetAsync(d, 0, 16, t->stream);
    if (e == hipSuccess) e = launch_selftest_division(n, seed, (unsigned long long *)d, (float *)(d + 8), t->stream);
    char host[16] = {0};
    if (e == hipSuccess) e = hipMemcpyAsync(host, d, 16, hipMemcpyDeviceToHost, t->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (d) (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(e, "rs_selftest_division");
    std::memcpy(mismatches, host, 8);
    if (first_bad) std::memcpy(first_bad, host + 8, 8);
    return RS_OK;
}

int rs_table_checksum(rs_table *t, uint64_t *out) {
    if (!t || !out) return fail(RS_ERR_INVALID, "rs_table_checksum: NULL argument");
    if (int rc = table_settle(t)) return rc;
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    unsigned long long *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, 16);
    if (e == hipSuccess) e = hipMemsetAsync(d, 0, 16, t->stream);
    for (int n = 0; n < int(t->nodes.size()) && e == hipSuccess; ++n) {
        const rs_node_desc &nd = t->nodes[size_t(n)];
        if (nd.n_actions == 0) continue;
        const size_t cells = t->pitch[size_t(n)] * nd.n_actions, lanes = size_t(nd.n_boards) * nd.n_clusters;
        e = launch_checksum(t->regrets_ptr(n), cells, t->cell_off[size_t(n)], nd.n_actions, t->tile[size_t(n)], lanes, elem_size(t->dtype), d, t->stream);
        if (e == hipSuccess) e = launch_checksum(t->ssum_ptr(n), cells, t->cell_off[size_t(n)], nd.n_actions, t->tile[size_t(n)], lanes, elem_size(t->dtype), d + 1, t->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, 16, hipMemcpyDeviceToHost, t->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (d) (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(e, "rs_table_checksum");
    return RS_OK;
}

// Infoset::get_strategy / get_final_strategy for ONE info set: the device kernel computes the whole
// 64-lane group that contains the lane (so the numbers come from the same code as the bulk path).
static int single_strategy(rs_table *t, int node, int board, int cluster, float *out, bool final_, const char *fn) {
    if (int rc = check_node(t, node, fn)) return rc;
    if (int rc = table_settle(t)) return rc;
    if (!out) return fail(RS_ERR_INVALID, std::string(fn) + ": out is NULL");
    const rs_node_desc &nd = t->nodes[node];
    if (nd.n_actions == 0) return RS_OK;   // vec![0.0; 0]
    if (board < 0 || uint32_t(board) >= nd.n_boards || cluster < 0 || uint32_t(cluster) >= nd.n_clusters)
        return fail(RS_ERR_OOB, std::string(fn) + ": index out of bounds");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    const size_t lane = size_t(board) * nd.n_clusters + size_t(cluster);
    const size_t g0 = lane / kLanePad * kLanePad;  // 64-lane aligned group
    const size_t es = elem_size(t->dtype);
    // gather the group's A rows into a compact [A][64] block (cached per-table scratch), run the kernel with pitch 64
    if (!t->d_query) RS_HIP(hipMalloc(&t->d_query, RS_MAX_ACTIONS * kLanePad * (4 + sizeof(float))), "hipMalloc(query scratch)");
    void *d_in = t->d_query;
    float *d_out = reinterpret_cast<float *>((char *)t->d_query + RS_MAX_ACTIONS * kLanePad * 4);
    // the 64-lane group never straddles a tile (tiles are multiples of 64 lanes): rows are tile[node] elements apart from its first element on
    const char *src = (const char *)(final_ ? t->ssum_ptr(node) : t->regrets_ptr(node)) + t->elem_index(node, 0, g0) * es;
    hipError_t e = hipMemcpy2DAsync(d_in, kLanePad * es, src, t->tile[size_t(node)] * es, kLanePad * es, nd.n_actions,
                                    hipMemcpyDeviceToDevice, t->stream);
    if (e == hipSuccess) e = launch_strategy(d_in, d_out, kLanePad, kLanePad, 31, int(nd.n_actions), t->dtype, t->stream);
    std::vector<float> host(nd.n_actions * kLanePad);
    if (e == hipSuccess)
        e = hipMemcpyAsync(host.data(), d_out, host.size() * sizeof(float), hipMemcpyDeviceToHost, t->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (e != hipSuccess) return hip_fail(e, fn);
    for (uint32_t a = 0; a < nd.n_actions; ++a) out[a] = host[a * kLanePad + (lane - g0)];
    return RS_OK;
}
size_t rs_table_tile_lanes(const rs_table *t, int node) {
    if (!t || node < 0 || size_t(node) >= t->nodes.size()) return 0;
    return t->tile[size_t(node)];
}
int rs_get_strategy(rs_table *t, int node, int board, int cluster, float *out) {
    return single_strategy(t, node, board, cluster, out, false, "rs_get_strategy");
}
int rs_get_final_strategy(rs_table *t, int node, int board, int cluster, float *out) {
    return single_strategy(t, node, board, cluster, out, true, "rs_get_final_strategy");
}

// ---- synthetic fills ---------------------------------------------------------------------------------------------
int rs_table_fill_random(rs_table *t, uint64_t seed, int64_t rlo, int64_t rhi, int64_t slo, int64_t shi) {
    if (!t) return fail(RS_ERR_INVALID, "rs_table_fill_random: table is NULL");
    if (rhi < rlo || shi < slo) return fail(RS_ERR_INVALID, "rs_table_fill_random: empty range");
    if (int rc = table_settle(t)) return rc;
    ++t->epoch;
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(launch_fill_random(t->d_regrets, t->n_cells, seed, rlo, rhi, t->dtype, t->stream), "fill_random(regrets)");
    RS_HIP(launch_fill_random(t->d_ssum, t->n_cells, seed ^ 0x5353554Dull /* "SSUM" */, slo, shi, t->dtype, t->stream),
           "fill_random(ssum)");
    return RS_OK;
}
int rs_table_plant_saturating(rs_table *t, uint64_t seed, uint32_t one_in) {
    if (!t || one_in == 0) return fail(RS_ERR_INVALID, "rs_table_plant_saturating: bad argument");
    if (t->dtype != RS_I32) return fail(RS_ERR_UNSUPPORTED, "rs_table_plant_saturating: i32 tables (the saturating range is the i32 range)");
    if (int rc = table_settle(t)) return rc;
    ++t->epoch;
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(launch_plant_saturating(t->d_regrets, t->n_cells, seed, one_in, t->stream), "plant_saturating");
    return RS_OK;
}
int rs_plant_outliers_f32(rs_table *t, float *d_dst, size_t n, uint64_t seed, uint32_t one_in, float magnitude) {
    if (!t || !d_dst || one_in == 0) return fail(RS_ERR_INVALID, "rs_plant_outliers_f32: bad argument");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(launch_plant_outliers(d_dst, n, seed, one_in, magnitude, t->stream), "plant_outliers");
    return RS_OK;
}
int rs_fill_uniform_f32_at(rs_table *t, float *d_dst, size_t n, uint64_t seed, float lo, float hi, uint64_t index_offset) {
    if (!t || !d_dst) return fail(RS_ERR_INVALID, "rs_fill_uniform_f32_at: NULL argument");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(launch_fill_uniform(d_dst, n, seed, lo, hi, t->stream, size_t(index_offset)), "fill_uniform");
    return RS_OK;
}
static int logical_sweep(rs_table *t, const uint64_t *lane_off, int op, uint64_t seed, int64_t rlo, int64_t rhi, int64_t slo, int64_t shi, uint64_t *out, const char *fn) {
    if (!t || !lane_off) return fail(RS_ERR_INVALID, std::string(fn) + ": NULL argument");
    if (op == 0 && t->dtype == RS_F32) return fail(RS_ERR_UNSUPPORTED, std::string(fn) + ": i32 and binary16 tables");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    unsigned long long *d = nullptr;
    hipError_t e = hipSuccess;
    if (op == 1) {
        e = hipMalloc((void **)&d, RS_MAX_ROUNDS * 2 * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemsetAsync(d, 0, RS_MAX_ROUNDS * 2 * sizeof(unsigned long long), t->stream);
    }
    for (int n = 0; n < int(t->nodes.size()) && e == hipSuccess; ++n) {
        const rs_node_desc &nd = t->nodes[size_t(n)];
        if (nd.n_actions == 0) continue;
        const size_t cells = t->pitch[size_t(n)] * nd.n_actions, lanes = size_t(nd.n_boards) * nd.n_clusters;
        const size_t off = size_t(lane_off[nd.round_idx]);
        e = launch_logical(t->regrets_ptr(n), cells, uint32_t(n), nd.n_actions, t->tile[size_t(n)], lanes, off, elem_size(t->dtype), op, seed, rlo, rhi,
                           d ? d + 2 * nd.round_idx : nullptr, t->stream);
        if (e == hipSuccess)
            e = launch_logical(t->ssum_ptr(n), cells, uint32_t(n), nd.n_actions, t->tile[size_t(n)], lanes, off, elem_size(t->dtype), op, seed ^ 0x5353554Dull, slo, shi,
                               d ? d + 2 * nd.round_idx + 1 : nullptr, t->stream);
    }
    if (op == 1 && e == hipSuccess) e = hipMemcpyAsync(out, d, RS_MAX_ROUNDS * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, t->stream);
    if (op == 1 && e == hipSuccess) e = hipStreamSynchronize(t->stream);
    if (d) (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(e, fn);
    return RS_OK;
}
int rs_table_fill_random_logical(rs_table *t, uint64_t seed, int64_t rlo, int64_t rhi, int64_t slo, int64_t shi, const uint64_t *lane_off) {
    if (rhi < rlo || shi < slo) return fail(RS_ERR_INVALID, "rs_table_fill_random_logical: empty range");
    if (int rc = table_settle(t)) return rc;
    if (t) ++t->epoch;
    return logical_sweep(t, lane_off, 0, seed, rlo, rhi, slo, shi, nullptr, "rs_table_fill_random_logical");
}
int rs_table_checksum_logical(rs_table *t, const uint64_t *lane_off, uint64_t *out) {
    if (!out) return fail(RS_ERR_INVALID, "rs_table_checksum_logical: NULL argument");
    if (int rc = table_settle(t)) return rc;
    return logical_sweep(t, lane_off, 1, 0, 0, 0, 0, 0, out, "rs_table_checksum_logical");
}
int rs_fill_uniform_f32(rs_table *t, float *d_dst, size_t n, uint64_t seed, float lo, float hi) {
    if (!t || !d_dst) return fail(RS_ERR_INVALID, "rs_fill_uniform_f32: NULL argument");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(launch_fill_uniform(d_dst, n, seed, lo, hi, t->stream), "fill_uniform");
    return RS_OK;
}

// ---- device memory helpers ------------------------------------------------------------------------------------------
int rs_dmalloc(rs_table *t, size_t bytes, void **d_out) {
    if (!t || !d_out) return fail(RS_ERR_INVALID, "rs_dmalloc: NULL argument");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(hipMalloc(d_out, bytes ? bytes : 1), "hipMalloc");
    return RS_OK;
}
int rs_dfree(rs_table *t, void *d_ptr) {
    if (!t) return fail(RS_ERR_INVALID, "rs_dfree: table is NULL");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
    RS_HIP(hipFree(d_ptr), "hipFree");
    return RS_OK;
}
int rs_table_deltas(rs_table *t, int32_t **d_dregrets, int32_t **d_dssum) {
    if (!t || !d_dregrets || !d_dssum) return fail(RS_ERR_INVALID, "rs_table_deltas: NULL argument");
    if (!t->d_dregrets || !t->d_dssum) return fail(RS_ERR_INVALID, "rs_table_deltas: the table has no delta tables (rs_solver_create_deals makes them)");
    *d_dregrets = static_cast<int32_t *>(t->d_dregrets);
    *d_dssum = static_cast<int32_t *>(t->d_dssum);
    return RS_OK;
}
int rs_h2d(rs_table *t, void *d_dst, const void *src, size_t bytes) {
    if (!t || !d_dst || !src) return fail(RS_ERR_INVALID, "rs_h2d: NULL argument");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, t->stream), "hipMemcpyAsync(h2d)");
    RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
    return RS_OK;
}
int rs_d2h(rs_table *t, void *dst, const void *d_src, size_t bytes) {
    if (!t || !dst || !d_src) return fail(RS_ERR_INVALID, "rs_d2h: NULL argument");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, t->stream), "hipMemcpyAsync(d2h)");
    RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
    return RS_OK;
}
int rs_dmemset(rs_table *t, void *d_dst, int byte, size_t bytes) {
    if (!t || !d_dst) return fail(RS_ERR_INVALID, "rs_dmemset: NULL argument");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(hipMemsetAsync(d_dst, byte, bytes, t->stream), "hipMemsetAsync");
    return RS_OK;
}
int rs_sync(rs_table *t) {
    if (!t) return fail(RS_ERR_INVALID, "rs_sync: table is NULL");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
    return RS_OK;
}

// ---- bulk kernels on one node -------------------------------------------------------------------------------------------
static int stage_job(rs_table *t, const NodeJob &job) {
    if (!t->d_job) RS_HIP(hipMalloc((void **)&t->d_job, sizeof(NodeJob)), "hipMalloc(job slot)");
    // stream-ordered: the previous kernel that read the slot has finished before this copy runs
    RS_HIP(hipMemcpyAsync(t->d_job, &job, sizeof(NodeJob), hipMemcpyHostToDevice, t->stream), "hipMemcpyAsync(job)");
    return RS_OK;
}

static void base_job(const rs_table *t, int node, NodeJob &job) {
    std::memset(&job, 0, sizeof(job));
    job.regrets = t->regrets_ptr(node);
    job.ssum = t->ssum_ptr(node);
    job.pitch = uint32_t(t->pitch[node]);
    job.row_stride = uint32_t(t->tile[size_t(node)]);
    job.tile_shift = t->tile_shift(node);
    job.n_vec = uint32_t(t->pitch[node] / kVec);
    job.n_actions = int32_t(t->nodes[node].n_actions);
    job.reach_const = 1.0f;
    job.node_index = uint32_t(node);
}

int rs_regret_match_node(rs_table *t, int node, float *d_strategy) {
    if (int rc = check_node(t, node, "rs_regret_match_node")) return rc;
    if (int rc = table_settle(t)) return rc;
    if (t->nodes[node].n_actions == 0) return RS_OK;   // a node without actions has nothing to compute
    if (!d_strategy) return fail(RS_ERR_INVALID, "rs_regret_match_node: d_strategy is NULL");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    const rs_node_desc &nd = t->nodes[node];
    prof_begin(t, RS_K_STRATEGY, double(nd.n_boards) * nd.n_clusters * nd.n_actions * (elem_size(t->dtype) + 4.0));
    hipError_t e = launch_strategy(t->regrets_ptr(node), d_strategy, uint32_t(t->pitch[node]), uint32_t(t->tile[size_t(node)]), t->tile_shift(node), int(nd.n_actions), t->dtype, t->stream);
    prof_end(t);
    RS_HIP(e, "k_strategy");
    return RS_OK;
}
int rs_final_strategy_node(rs_table *t, int node, float *d_strategy) {
    if (int rc = check_node(t, node, "rs_final_strategy_node")) return rc;
    if (int rc = table_settle(t)) return rc;
    if (t->nodes[node].n_actions == 0) return RS_OK;   // a node without actions has nothing to compute
    if (!d_strategy) return fail(RS_ERR_INVALID, "rs_final_strategy_node: d_strategy is NULL");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    const rs_node_desc &nd = t->nodes[node];
    prof_begin(t, RS_K_STRATEGY, double(nd.n_boards) * nd.n_clusters * nd.n_actions * (elem_size(t->dtype) + 4.0));
    hipError_t e = launch_strategy(t->ssum_ptr(node), d_strategy, uint32_t(t->pitch[node]), uint32_t(t->tile[size_t(node)]), t->tile_shift(node), int(nd.n_actions), t->dtype, t->stream);
    prof_end(t);
    RS_HIP(e, "k_strategy");
    return RS_OK;
}
int rs_final_strategy_all(rs_table *t, float *d_out) {
    if (!t || !d_out) return fail(RS_ERR_INVALID, "rs_final_strategy_all: NULL argument");
    for (int n = 0; n < int(t->nodes.size()); ++n)
        if (int rc = rs_final_strategy_node(t, n, d_out + t->cell_off[n])) return rc;
    return RS_OK;
}

static int check_mode(const rs_table *t, int mode, const char *fn) {
    const int arith = mode & RS_UPD_ARITH_MASK;
    if (arith != RS_UPD_CLAMP_I64 && arith != RS_UPD_WRAP_I32) return fail(RS_ERR_INVALID, std::string(fn) + ": bad update mode");
    if (mode & ~(RS_UPD_ARITH_MASK | RS_UPD_RMPLUS | RS_UPD_PRUNE)) return fail(RS_ERR_INVALID, std::string(fn) + ": unknown mode flag");
    if ((mode & RS_UPD_PRUNE) && t->dtype != RS_I32)
        return fail(RS_ERR_UNSUPPORTED, std::string(fn) + ": RS_UPD_PRUNE needs an RS_I32 table (threshold is an i32 regret, cfr.rs:352)");
    if ((mode & RS_UPD_RMPLUS) && t->dtype == RS_I32 && arith != RS_UPD_CLAMP_I64)
        return fail(RS_ERR_UNSUPPORTED, std::string(fn) + ": RS_UPD_RMPLUS on i32 tables uses the clamp arithmetic");
    return RS_OK;
}

int rs_update_node(rs_table *t, int node, const float *d_action_utils, const float *d_reach, float scale, int mode,
                   float *d_node_util) {
    if (int rc = check_node(t, node, "rs_update_node")) return rc;
    if (int rc = table_settle(t)) return rc;
    if (t->nodes[node].n_actions == 0) return RS_OK;   // a node without actions has nothing to compute
    ++t->epoch;
    if (!d_action_utils) return fail(RS_ERR_INVALID, "rs_update_node: d_action_utils is NULL");
    if (int rc = check_mode(t, mode, "rs_update_node")) return rc;
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    NodeJob job;
    base_job(t, node, job);
    for (int a = 0; a < job.n_actions; ++a) job.child[a] = ChildSrc{d_action_utils + size_t(a) * job.pitch, 0.0f, CH_BUF};
    job.reach = d_reach;
    job.out_util = d_node_util;
    job.scale = scale;
    if (int rc = stage_job(t, job)) return rc;
    prof_begin(t, RS_K_UPDATE, algo_bytes_update(t, node, job.n_actions, d_reach != nullptr, d_node_util != nullptr));
    hipError_t e = launch_update(t->d_job, 1, job.n_vec, job.n_actions, KernelCfg{t->dtype, mode}, t->stream);
    prof_end(t);
    RS_HIP(e, "k_update");
    return RS_OK;
}

int rs_node_util(rs_table *t, int node, const float *d_action_utils, float *d_node_util) {
    if (int rc = check_node(t, node, "rs_node_util")) return rc;
    if (int rc = table_settle(t)) return rc;
    if (t->nodes[node].n_actions == 0) return RS_OK;   // a node without actions has nothing to compute
    if (!d_action_utils || !d_node_util) return fail(RS_ERR_INVALID, "rs_node_util: NULL argument");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    NodeJob job;
    base_job(t, node, job);
    for (int a = 0; a < job.n_actions; ++a) job.child[a] = ChildSrc{d_action_utils + size_t(a) * job.pitch, 0.0f, CH_BUF};
    job.out_util = d_node_util;
    if (int rc = stage_job(t, job)) return rc;
    const rs_node_desc &nd = t->nodes[node];
    prof_begin(t, RS_K_NODE_UTIL, double(nd.n_boards) * nd.n_clusters * (nd.n_actions * (elem_size(t->dtype) + 4.0) + 4.0));
    hipError_t e = launch_node_util(t->d_job, 1, job.n_vec, job.n_actions, KernelCfg{t->dtype, 0}, nullptr, t->stream);
    prof_end(t);
    RS_HIP(e, "k_node_util");
    return RS_OK;
}

int rs_child_reach(rs_table *t, int node, const float *d_reach, float *d_child_reach) {
    if (int rc = check_node(t, node, "rs_child_reach")) return rc;
    if (int rc = table_settle(t)) return rc;
    if (t->nodes[node].n_actions == 0) return RS_OK;   // a node without actions has nothing to compute
    if (!d_child_reach) return fail(RS_ERR_INVALID, "rs_child_reach: d_child_reach is NULL");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    NodeJob job;
    base_job(t, node, job);
    job.reach = d_reach;
    for (int a = 0; a < job.n_actions; ++a) job.out_reach[a] = d_child_reach + size_t(a) * job.pitch;
    if (int rc = stage_job(t, job)) return rc;
    const rs_node_desc &nd = t->nodes[node];
    prof_begin(t, RS_K_REACH, double(nd.n_boards) * nd.n_clusters * (nd.n_actions * (elem_size(t->dtype) + 4.0) + (d_reach ? 4.0 : 0.0)));
    hipError_t e = launch_reach(t->d_job, 1, job.n_vec, job.n_actions, KernelCfg{t->dtype, 0}, nullptr, t->stream);
    prof_end(t);
    RS_HIP(e, "k_reach");
    return RS_OK;
}

float rs_discount_factor(uint64_t tc, uint64_t interval) {  // cfr.rs:248-249
    if (interval == 0) return 0.0f;
    const float p = float(tc / interval);
    return p / (p + 1.0f);
}

int rs_discount(rs_table *t, float d) {
    if (!t) return fail(RS_ERR_INVALID, "rs_discount: table is NULL");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    for (rs_solver *s : t->solvers)
        if (solver_is_primary(s)) {   // inside a training loop whose kept shadow records are the working copy: they and the table without their nodes
            prof_begin(t, RS_K_DISCOUNT, double(t->n_cells) * 4.0 * elem_size(t->dtype));
            const int rc = solver_discount_primary(s, d);
            prof_end(t);
            return rc;
        }
    prof_begin(t, RS_K_DISCOUNT, double(t->n_cells) * 4.0 * elem_size(t->dtype));
    hipError_t e = launch_discount(t->d_regrets, t->d_ssum, t->n_cells, d, t->dtype, t->stream);
    prof_end(t);
    RS_HIP(e, "k_discount");
    const uint64_t before = t->epoch++;
    for (rs_solver *s : t->solvers) solver_table_discounted(s, d, before);   // kept shadow records take the same sweep (every int on its own: (x as f32 * d) as i32)
    return RS_OK;
}

// ---- showdown signs from cards ------------------------------------------------------------------------------
int rs_showdown_sign(rs_table *t, const uint8_t *d_cards, uint32_t n_deals, float *d_sign) {
    if (!t || !d_cards || !d_sign) return fail(RS_ERR_INVALID, "rs_showdown_sign: NULL argument");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    const uint32_t pitch = uint32_t(round_up(n_deals, kLanePad));
    RS_HIP(launch_showdown_sign(d_cards, d_sign, n_deals, pitch, t->stream), "k_showdown_sign");
    return RS_OK;
}

// ---- profiling ----------------------------------------------------------------------------------------------------------------
int rs_profile_enable(rs_table *t, int on) {
    if (!t) return fail(RS_ERR_INVALID, "rs_profile_enable: table is NULL");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
    prof_collect(t);
    t->prof.on = on != 0;
    return RS_OK;
}
int rs_profile_read(rs_table *t, rs_profile *out) {
    if (!t || !out) return fail(RS_ERR_INVALID, "rs_profile_read: NULL argument");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
    prof_collect(t);
    *out = t->prof.acc;
    return RS_OK;
}
// The streaming rate this card reaches on a plain float4 copy (read `bytes`, write `bytes`), best of a few grid sizes: the practical ceiling
// the update kernels are compared with beside the 8 TB/s specification (cards of one pool differ by more than 10 %).
int rs_stream_probe(rs_table *t, size_t bytes, int reps, double *gbps) {
    if (!t || !gbps) return fail(RS_ERR_INVALID, "rs_stream_probe: NULL argument");
    if (bytes < (1u << 20) || reps < 1) return fail(RS_ERR_INVALID, "rs_stream_probe: at least 1 MiB and one repetition");
    bytes &= ~size_t(4095);
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    void *in = nullptr, *out = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t e = hipMalloc(&in, bytes);
    if (e == hipSuccess) e = hipMalloc(&out, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(in, 1, bytes, t->stream);
    if (e == hipSuccess) e = hipMemsetAsync(out, 0, bytes, t->stream);
    if (e == hipSuccess) e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    double best = 0.0;
    for (unsigned blocks : {1024u, 4096u, 16384u}) {
        if (e == hipSuccess) e = launch_probe_copy(in, out, bytes, blocks, t->stream);   // warm-up
        if (e == hipSuccess) e = hipEventRecord(a, t->stream);
        for (int r = 0; e == hipSuccess && r < reps; ++r) e = launch_probe_copy(in, out, bytes, blocks, t->stream);
        if (e == hipSuccess) e = hipEventRecord(b, t->stream);
        if (e == hipSuccess) e = hipEventSynchronize(b);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
        if (e == hipSuccess && ms > 0.0f) best = std::max(best, 2.0 * double(bytes) * reps / (double(ms) * 1e-3) / 1e9);
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (in) (void)hipFree(in);
    if (out) (void)hipFree(out);
    if (e != hipSuccess) return hip_fail(e, "rs_stream_probe");
    *gbps = best;
    return RS_OK;
}
// step boundaries of bench.py's timed region: one event per mark on the table's stream, durations between consecutive marks
int rs_profile_mark(rs_table *t) {
    if (!t) return fail(RS_ERR_INVALID, "rs_profile_mark: table is NULL");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    hipEvent_t e = prof_event(t);
    if (!e) return fail(RS_ERR_HIP, "rs_profile_mark: hipEventCreate failed");
    RS_HIP(hipEventRecord(e, t->stream), "hipEventRecord");
    t->prof.marks.push_back(e);
    return RS_OK;
}
int rs_profile_marks(rs_table *t, float *ms_out, size_t cap, size_t *n_out) {
    if (!t || !n_out || (!ms_out && cap)) return fail(RS_ERR_INVALID, "rs_profile_marks: NULL argument");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
    const size_t n = t->prof.marks.empty() ? 0 : t->prof.marks.size() - 1;
    *n_out = n;
    hipError_t e = hipSuccess;
    for (size_t i = 0; i < n && i < cap && e == hipSuccess; ++i) e = hipEventElapsedTime(&ms_out[i], t->prof.marks[i], t->prof.marks[i + 1]);
    for (hipEvent_t ev : t->prof.marks) t->prof.pool.push_back(ev);
    t->prof.marks.clear();
    if (e != hipSuccess) return hip_fail(e, "rs_profile_marks");
    return RS_OK;
}
int rs_profile_reset(rs_table *t) {
    if (!t) return fail(RS_ERR_INVALID, "rs_profile_reset: table is NULL");
    RS_HIP(hipSetDevice(t->device), "hipSetDevice");
    RS_HIP(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
    prof_collect(t);
    std::memset(&t->prof.acc, 0, sizeof(t->prof.acc));
    return RS_OK;
}

}  // extern "C"
