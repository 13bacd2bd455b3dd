// rs_abstraction.cpp -- host plumbing in front of get-infoset addressing.
//   * bucket files: flat little-endian u32 per canonical hand index, as written by gen_abstraction
//     (gen_abstraction/main.rs:372-380) and read by EMD::init / OCHS::init (card_abstraction.rs:227-229, :269-271);
//   * index_to_cluster (card_abstraction.rs:20-29);
//   * dense ids: generate_maps (card_abstraction.rs:75-184) numbers buckets 0..size in CHANNEL-ARRIVAL order of a
//     rayon par_iter, i.e. nondeterministically.  Here the order is the first appearance in the caller's
//     enumeration, which is deterministic whenever the enumeration is.
// The canonical hand index itself comes from rust_poker's hand_indexer_s (absent from the reference tree, out of
// scope): callers pass indices in.  No GPU code here.
#include <cstdio>
#include <cstring>
#include <new>
#include <unordered_map>

#include "rs_internal.hpp"

struct rs_dense_map {
    std::unordered_map<uint64_t, uint32_t> ids;   // HashMap<u64, usize> cluster_map[player] (card_abstraction.rs:36)
    std::vector<uint64_t> keys;                   // dense id -> bucket
};

using namespace rs;

extern "C" {

int rs_cluster_file_read(const char *path, uint32_t **out, size_t *n_out) {
    if (!path || !out || !n_out) return fail(RS_ERR_INVALID, "rs_cluster_file_read: NULL argument");
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(RS_ERR_INVALID, std::string("rs_cluster_file_read: cannot open ") + path);   // File::open(..).unwrap() panics
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (bytes < 0 || bytes % 4 != 0) {
        std::fclose(f);
        return fail(RS_ERR_INVALID, "rs_cluster_file_read: file size is not a multiple of 4");
    }
    const size_t n = size_t(bytes) / 4;
    uint32_t *buf = static_cast<uint32_t *>(std::malloc(n ? n * 4 : 4));
    if (!buf) {
        std::fclose(f);
        return fail(RS_ERR_OOM, "rs_cluster_file_read: out of memory");
    }
    std::vector<unsigned char> raw(n * 4);
    const bool ok = std::fread(raw.data(), 1, raw.size(), f) == raw.size();
    std::fclose(f);
    if (!ok) {
        std::free(buf);
        return fail(RS_ERR_INVALID, "rs_cluster_file_read: short read");
    }
    for (size_t i = 0; i < n; ++i)   // explicit little-endian decode (LEUnpacker, card_abstraction.rs:228)
        buf[i] = uint32_t(raw[4 * i]) | uint32_t(raw[4 * i + 1]) << 8 | uint32_t(raw[4 * i + 2]) << 16 | uint32_t(raw[4 * i + 3]) << 24;
    *out = buf;
    *n_out = n;
    return RS_OK;
}

int rs_cluster_file_write(const char *path, const uint32_t *clusters, size_t n) {
    if (!path || (!clusters && n)) return fail(RS_ERR_INVALID, "rs_cluster_file_write: NULL argument");
    FILE *f = std::fopen(path, "wbx");   // OpenOptions::create_new(true): fail if the file exists (gen_abstraction/main.rs:372-375)
    if (!f) return fail(RS_ERR_INVALID, std::string("rs_cluster_file_write: cannot create ") + path + " (it must not exist)");
    std::vector<unsigned char> raw(n * 4);
    for (size_t i = 0; i < n; ++i) {
        raw[4 * i] = clusters[i] & 0xff;
        raw[4 * i + 1] = (clusters[i] >> 8) & 0xff;
        raw[4 * i + 2] = (clusters[i] >> 16) & 0xff;
        raw[4 * i + 3] = (clusters[i] >> 24) & 0xff;
    }
    const bool ok = std::fwrite(raw.data(), 1, raw.size(), f) == raw.size();
    std::fclose(f);
    return ok ? RS_OK : fail(RS_ERR_INVALID, "rs_cluster_file_write: short write");
}

void rs_free_u32(uint32_t *p) { std::free(p); }

// index_to_cluster (card_abstraction.rs:20-29): arr[index] with a bucket file, the index itself without (ISOMORPHIC)
int rs_index_to_cluster(const uint32_t *cluster_arr, size_t arr_len, const uint64_t *indices, size_t n, uint64_t *out) {
    if (!indices || !out) return fail(RS_ERR_INVALID, "rs_index_to_cluster: NULL argument");
    for (size_t i = 0; i < n; ++i) {
        if (!cluster_arr) out[i] = indices[i];
        else if (indices[i] < arr_len) out[i] = cluster_arr[indices[i]];
        else return fail(RS_ERR_OOB, "index out of bounds: the len is " + std::to_string(arr_len) + " but the index is " + std::to_string(indices[i]));
    }
    return RS_OK;
}

// generate_maps' consumer (card_abstraction.rs:115-123): first sighting of a bucket gets the next dense id
int rs_dense_map_create(const uint64_t *buckets, size_t n, rs_dense_map **out) {
    if ((!buckets && n) || !out) return fail(RS_ERR_INVALID, "rs_dense_map_create: NULL argument");
    rs_dense_map *m = new (std::nothrow) rs_dense_map();
    if (!m) return fail(RS_ERR_OOM, "rs_dense_map_create: out of memory");
    for (size_t i = 0; i < n; ++i)
        if (m->ids.emplace(buckets[i], uint32_t(m->keys.size())).second) m->keys.push_back(buckets[i]);
    *out = m;
    return RS_OK;
}
void rs_dense_map_destroy(rs_dense_map *m) { delete m; }
size_t rs_dense_map_size(const rs_dense_map *m) { return m ? m->keys.size() : 0; }   // get_size (card_abstraction.rs:211-213)

// get_cluster's last step: *self.cluster_map[player].get(&cluster).unwrap() (card_abstraction.rs:208)
int rs_dense_map_lookup(const rs_dense_map *m, const uint64_t *buckets, size_t n, uint32_t *dense_out) {
    if (!m || (!buckets && n) || !dense_out) return fail(RS_ERR_INVALID, "rs_dense_map_lookup: NULL argument");
    for (size_t i = 0; i < n; ++i) {
        auto it = m->ids.find(buckets[i]);
        if (it == m->ids.end())
            return fail(RS_ERR_OOB, "rs_dense_map_lookup: bucket " + std::to_string(buckets[i]) + " is not in the map (Rust: unwrap on None)");
        dense_out[i] = it->second;
    }
    return RS_OK;
}
int rs_dense_map_keys(const rs_dense_map *m, uint64_t *keys_out) {
    if (!m || !keys_out) return fail(RS_ERR_INVALID, "rs_dense_map_keys: NULL argument");
    std::memcpy(keys_out, m->keys.data(), m->keys.size() * sizeof(uint64_t));
    return RS_OK;
}

}  // extern "C"

// ---- table checkpoints: the reference never writes the trained table anywhere ---------------------------
// File = header {magic "RSTB", u32 version, u32 dtype, u32 n_nodes}, n_nodes x rs_node_desc{u32 A, u32 clusters, u32 boards,
// u8 player, u8 round, u16 0}, then per node regrets[A][lanes] and strategy_sum[A][lanes] (table element type, little
// endian, WITHOUT pitch padding, lanes = boards*clusters), then a u64 FNV-1a of everything before it.
namespace {
constexpr uint32_t kCkptMagic = 0x42545352u;   // "RSTB"
constexpr uint32_t kCkptVersion = 1;
struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void add(const void *p, size_t n) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < n; ++i) {
            h ^= b[i];
            h *= 1099511628211ull;
        }
    }
};
bool put(FILE *f, Fnv &fnv, const void *p, size_t n) {
    fnv.add(p, n);
    return std::fwrite(p, 1, n, f) == n;
}
bool get(FILE *f, Fnv &fnv, void *p, size_t n) {
    if (std::fread(p, 1, n, f) != n) return false;
    fnv.add(p, n);
    return true;
}
}  // namespace

extern "C" {

int rs_table_save(rs_table *t, const char *path) {
    if (!t || !path) return fail(RS_ERR_INVALID, "rs_table_save: NULL argument");
    if (int rc = rs::table_settle(t)) return rc;
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(RS_ERR_INVALID, std::string("rs_table_save: cannot create ") + path);
    Fnv fnv;
    const uint32_t head[4] = {kCkptMagic, kCkptVersion, uint32_t(t->dtype), uint32_t(t->nodes.size())};
    bool ok = put(f, fnv, head, sizeof(head));
    for (const rs_node_desc &d : t->nodes) {
        const uint32_t rec[4] = {d.n_actions, d.n_clusters, d.n_boards, uint32_t(d.player) | uint32_t(d.round_idx) << 8};
        ok = ok && put(f, fnv, rec, sizeof(rec));
    }
    const size_t es = elem_size(t->dtype);
    std::vector<char> buf;
    int rc = RS_OK;
    for (int n = 0; ok && rc == RS_OK && n < int(t->nodes.size()); ++n) {
        const rs_node_desc &d = t->nodes[n];
        const size_t lanes = size_t(d.n_boards) * d.n_clusters;
        buf.resize(lanes * d.n_actions * es);
        for (int which = 0; which < 2 && ok && rc == RS_OK; ++which) {
            // tile-aware: the file holds plain [A][lanes] rows whatever the node's in-memory block looks like
            rc = table_copy_node_raw(t, n, which, buf.data(), 1);
            if (rc == RS_OK) ok = put(f, fnv, buf.data(), buf.size());
        }
    }
    const uint64_t sum = fnv.h;
    ok = ok && std::fwrite(&sum, 1, 8, f) == 8;
    std::fclose(f);
    if (rc != RS_OK) return rc;
    return ok ? RS_OK : fail(RS_ERR_INVALID, "rs_table_save: short write");
}

int rs_table_load(const char *path, int device, rs_table **out) {
    if (!path || !out) return fail(RS_ERR_INVALID, "rs_table_load: NULL argument");
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(RS_ERR_INVALID, std::string("rs_table_load: cannot open ") + path);
    Fnv fnv;
    uint32_t head[4];
    if (!get(f, fnv, head, sizeof(head)) || head[0] != kCkptMagic || head[1] != kCkptVersion || head[3] == 0 || head[3] > (1u << 24)) {
        std::fclose(f);
        return fail(RS_ERR_INVALID, "rs_table_load: not a version-1 RSTB checkpoint");
    }
    std::vector<rs_node_desc> descs(head[3]);
    for (rs_node_desc &d : descs) {
        uint32_t rec[4];
        if (!get(f, fnv, rec, sizeof(rec))) {
            std::fclose(f);
            return fail(RS_ERR_INVALID, "rs_table_load: truncated header");
        }
        d.n_actions = rec[0];
        d.n_clusters = rec[1];
        d.n_boards = rec[2];
        d.player = uint8_t(rec[3] & 0xff);
        d.round_idx = uint8_t(rec[3] >> 8);
    }
    rs_table *t = nullptr;
    int rc = rs_table_create(descs.data(), int(descs.size()), int(head[2]), device, &t);
    if (rc != RS_OK) {
        std::fclose(f);
        return rc;
    }
    const size_t es = elem_size(t->dtype);
    std::vector<char> buf;
    bool ok = true;
    for (int n = 0; ok && rc == RS_OK && n < int(descs.size()); ++n) {
        const rs_node_desc &d = descs[n];
        const size_t lanes = size_t(d.n_boards) * d.n_clusters;
        buf.resize(lanes * d.n_actions * es);
        for (int which = 0; which < 2 && ok && rc == RS_OK; ++which) {
            ok = get(f, fnv, buf.data(), buf.size());
            if (!ok) break;
            rc = table_copy_node_raw(t, n, which, buf.data(), 0);
        }
    }
    uint64_t want = 0;
    const uint64_t have = fnv.h;
    ok = ok && std::fread(&want, 1, 8, f) == 8 && want == have;
    std::fclose(f);
    if (rc == RS_OK && !ok) rc = fail(RS_ERR_INVALID, "rs_table_load: truncated file or checksum mismatch");
    if (rc != RS_OK) {
        rs_table_destroy(t);
        return rc;
    }
    *out = t;
    return RS_OK;
}

}  // extern "C"
