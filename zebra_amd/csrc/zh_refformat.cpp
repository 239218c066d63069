// zh_refformat.cpp -- the VALUES of the reference's on-disk partitions, host code only.
//
// The reference keeps its index in two fjall partitions (/root/reference/src/database/index/lsh.rs:62-120):
//   "<uuid>-embeddings": key = the vector's 16 uuid bytes, value = bincode(legacy) of Embedding<N>  (lsh.rs:91-97)
//   "<uuid>-trees":      key = the tree's 16 uuid bytes,   value = bincode(legacy) of Node<N>       (lsh.rs:99-105)
// fjall's own file format (an LSM tree, a third-party crate that is not in the reference tree) is NOT read here: the
// host shim -- which links fjall anyway -- iterates the partitions and hands the raw values over; this file turns
// them into the flat forest of zh_index_set_forest and back.
//
// bincode "legacy" configuration = little endian, fixed-width integers, usize as u64.  What serde derives for the
// reference's types (lsh.rs:16-25,46-60; lib.rs:15-18), restated from the published encodings of serde / bincode 2 /
// serde_with 3 / uuid 1 (none of those crates is available in this build environment: FORMAT UNVERIFIED against
// them, pinned only by the hand-assembled known-answer bytes in tests/test_refformat.py):
//   Embedding<N>   newtype around [f32; N] serialised "as [_; N]" -> a tuple: N x f32, no length prefix   (4N bytes)
//   Hyperplane<N>  { coefficients: Embedding<N>, constant: f32 }                                           (4N+4)
//   Node<N>        enum: u32 variant index, 0 = Inner(Box<InnerNode>), 1 = Leaf(Box<LeafNode>); Box is transparent
//   InnerNode<N>   { hyperplane, left_node, right_node } in that order (pre-order, left = BELOW side, lsh.rs:260-264)
//   LeafNode       newtype around Vec<Uuid>: u64 count, then per id the uuid crate's binary form = a byte string:
//                  u64 length (16) + 16 bytes
//
// The database HEADER, the `.zebra` file itself (/root/reference/src/database/core.rs:19-29 DatabaseInner, written by
// save_database core.rs:183-190, read by open core.rs:92-102), same bincode-legacy configuration:
//   DatabaseInner<N, Met, Mod> { uuid: Uuid, model: Mod, metric: Met, index_options: LSHIndexOptions<N> } in that order
//     uuid           byte string: u64 length (16) + 16 bytes                                                   (24 bytes)
//     model          the reference's three models are unit structs (model/text.rs:11, image.rs:50, audio.rs:106) (0 bytes)
//     metric         a unit struct for 11 of the 13 metrics (distance.rs:15-158)                                (0 bytes);
//                    MinkowskiDistance / PNormDistance { power: i32 } (distance.rs:160-190)                     (4 bytes)
//     index_options  { max_node_size: usize, num_trees: usize } (lsh.rs:124-129), usize as u64                  (16 bytes)
// Met and Mod are TYPE parameters: the file does not say which they are, the opener states them (as `Database::open::<..>` does).
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

// Only the public header: this file has no HIP in it, so that it also builds stand-alone with gcc -fsanitize=address,undefined
// (tests/asan/Makefile) -- it parses untrusted on-disk bytes.
#include "../../include/zebra_hip.h"

int zh_set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));  // zh_api.hip (tests/asan: a stub)

namespace {

struct Key16 {
    uint64_t a, b;
    bool operator==(const Key16 &o) const { return a == o.a && b == o.b; }
};
struct Key16Hash {
    size_t operator()(const Key16 &k) const {
        uint64_t x = k.a * 0x9E3779B97F4A7C15ull ^ (k.b + 0xBF58476D1CE4E5B9ull + (k.a << 6) + (k.a >> 2));
        return (size_t)(x ^ (x >> 29));
    }
};

struct Reader {
    const uint8_t *p;
    size_t n, pos = 0;
    bool take(void *dst, size_t k) {
        if (n - pos < k) return false;
        memcpy(dst, p + pos, k);
        pos += k;
        return true;
    }
};

}  // namespace

struct zh_ref_forest {
    std::vector<int32_t> plane, left, right;
    std::vector<uint32_t> roots, leaf_ids;
    std::vector<float> planes, consts;
};

extern "C" int zh_ref_forest_decode(uint32_t dim, size_t n_trees, const uint8_t *const *values, const size_t *lens,
                                    size_t n_vectors, const uint8_t *uuids, zh_ref_forest **out, uint64_t *out_unknown_ids) {
    if (!out || (n_trees && (!values || !lens)) || (n_vectors && !uuids) || dim == 0)
        return zh_set_error(ZH_EINVAL, "zh_ref_forest_decode: null argument or dim == 0");
    if (n_vectors > 0xFFFFFFFFull) return zh_set_error(ZH_ELIMIT, "zh_ref_forest_decode: more than 2^32-1 vectors");
    zh_ref_forest *f = new (std::nothrow) zh_ref_forest();
    if (!f) return zh_set_error(ZH_ENOMEM, "out of host memory");
    uint64_t unknown = 0;
    try {
        std::unordered_map<Key16, uint32_t, Key16Hash> row_of;
        row_of.reserve(n_vectors * 2);
        for (size_t i = 0; i < n_vectors; i++) {
            Key16 k;
            memcpy(&k, uuids + 16 * i, 16);
            if (!row_of.emplace(k, (uint32_t)i).second) {
                delete f;
                return zh_set_error(ZH_EINVAL, "zh_ref_forest_decode: vector %zu repeats an earlier uuid", i);
            }
        }
        struct Slot { int32_t node; int side; uint32_t depth; };  // where the next decoded node's index goes
        std::vector<Slot> todo;
        for (size_t t = 0; t < n_trees; t++) {
            Reader r{values[t], lens[t]};
            f->roots.push_back(0);
            todo.clear();
            todo.push_back({-1, 0, 0});
            while (!todo.empty()) {
                const Slot s = todo.back();
                todo.pop_back();
                uint32_t tag;
                if (!r.take(&tag, 4)) { delete f; return zh_set_error(ZH_EINVAL, "tree %zu: truncated at byte %zu (node tag)", t, r.pos); }
                if (f->plane.size() >= 0x7FFFFFFFu) { delete f; return zh_set_error(ZH_ELIMIT, "more than 2^31-1 nodes"); }
                const int32_t me = (int32_t)f->plane.size();
                if (s.node < 0) f->roots[t] = (uint32_t)me;
                else (s.side ? f->right : f->left)[s.node] = me;
                if (tag == 0) {  // Inner
                    if (s.depth >= ZH_MAX_DEPTH + 3) { delete f; return zh_set_error(ZH_ELIMIT, "tree %zu: deeper than %d levels", t, ZH_MAX_DEPTH + 3); }
                    const size_t p = f->consts.size();
                    f->planes.resize((p + 1) * (size_t)dim);
                    float c;
                    if (!r.take(f->planes.data() + p * dim, 4 * (size_t)dim) || !r.take(&c, 4)) {
                        delete f;
                        return zh_set_error(ZH_EINVAL, "tree %zu: truncated at byte %zu (hyperplane)", t, r.pos);
                    }
                    f->consts.push_back(c);
                    f->plane.push_back((int32_t)p); f->left.push_back(-1); f->right.push_back(-1);
                    todo.push_back({me, 1, s.depth + 1});  // right_node follows left_node in the stream
                    todo.push_back({me, 0, s.depth + 1});
                } else if (tag == 1) {  // Leaf
                    uint64_t cnt;
                    if (!r.take(&cnt, 8) || cnt > (r.n - r.pos) / 24) { delete f; return zh_set_error(ZH_EINVAL, "tree %zu: bad leaf length at byte %zu", t, r.pos); }
                    const size_t off = f->leaf_ids.size();
                    for (uint64_t i = 0; i < cnt; i++) {
                        uint64_t blen;
                        Key16 k;
                        if (!r.take(&blen, 8) || blen != 16 || !r.take(&k, 16)) { delete f; return zh_set_error(ZH_EINVAL, "tree %zu: bad uuid at byte %zu", t, r.pos); }
                        auto it = row_of.find(k);
                        if (it == row_of.end()) unknown++;  // removed vectors stay in the reference's trees (lsh.rs:473-503)
                        else f->leaf_ids.push_back(it->second);
                    }
                    if (f->leaf_ids.size() > 0xFFFFFFFFull) { delete f; return zh_set_error(ZH_ELIMIT, "more than 2^32-1 leaf entries"); }
                    f->plane.push_back(-1); f->left.push_back((int32_t)(uint32_t)off); f->right.push_back((int32_t)(f->leaf_ids.size() - off));
                } else {
                    delete f;
                    return zh_set_error(ZH_EINVAL, "tree %zu: unknown Node variant %u at byte %zu", t, tag, r.pos - 4);
                }
            }
            if (r.pos != r.n) { delete f; return zh_set_error(ZH_EINVAL, "tree %zu: %zu trailing bytes", t, r.n - r.pos); }
        }
    } catch (const std::bad_alloc &) {
        delete f;
        return zh_set_error(ZH_ENOMEM, "out of host memory");
    }
    if (out_unknown_ids) *out_unknown_ids = unknown;
    *out = f;
    return ZH_OK;
}

extern "C" int zh_ref_forest_view(const zh_ref_forest *f, zh_forest_view *v) {
    if (!f || !v) return zh_set_error(ZH_EINVAL, "zh_ref_forest_view: null argument");
    v->n_nodes = (uint32_t)f->plane.size();
    v->n_planes = (uint32_t)f->consts.size();
    v->n_trees = (uint32_t)f->roots.size();
    v->n_leaf_ids = f->leaf_ids.size();
    v->plane = f->plane.data(); v->left = f->left.data(); v->right = f->right.data();
    v->roots = f->roots.data();
    v->planes = f->planes.data(); v->consts = f->consts.data();
    v->leaf_ids = f->leaf_ids.data();
    return ZH_OK;
}

extern "C" void zh_ref_forest_free(zh_ref_forest *f) { delete f; }

extern "C" int zh_ref_tree_encode(const zh_forest_view *v, uint32_t dim, uint32_t tree, const uint8_t *uuids, uint64_t n_rows,
                                  uint8_t *out, size_t cap, size_t *out_len) {
    if (!v || !out_len || dim == 0 || (n_rows && !uuids)) return zh_set_error(ZH_EINVAL, "zh_ref_tree_encode: null argument or dim == 0");
    if (tree >= v->n_trees) return zh_set_error(ZH_EINVAL, "zh_ref_tree_encode: tree %u of %u", tree, v->n_trees);
    size_t pos = 0;
    auto put = [&](const void *src, size_t k) {
        if (out && pos + k <= cap) memcpy(out + pos, src, k);
        pos += k;
    };
    std::vector<uint32_t> todo{v->roots[tree]};
    uint64_t visited = 0;
    while (!todo.empty()) {
        const uint32_t i = todo.back();
        todo.pop_back();
        if (i >= v->n_nodes || ++visited > v->n_nodes) return zh_set_error(ZH_EINVAL, "zh_ref_tree_encode: malformed forest");
        if (v->plane[i] >= 0) {
            const uint32_t tag = 0, p = (uint32_t)v->plane[i];
            if (p >= v->n_planes) return zh_set_error(ZH_EINVAL, "zh_ref_tree_encode: plane index out of range");
            put(&tag, 4);
            put(v->planes + (size_t)p * dim, 4 * (size_t)dim);
            put(v->consts + p, 4);
            todo.push_back((uint32_t)v->right[i]);  // left_node is written first
            todo.push_back((uint32_t)v->left[i]);
        } else {
            const uint32_t tag = 1;
            const uint64_t off = (uint32_t)v->left[i], cnt = (uint32_t)v->right[i], sixteen = 16;
            if (off + cnt > v->n_leaf_ids) return zh_set_error(ZH_EINVAL, "zh_ref_tree_encode: leaf run out of range");
            put(&tag, 4);
            put(&cnt, 8);
            for (uint64_t j = 0; j < cnt; j++) {
                const uint32_t row = v->leaf_ids[off + j];
                if (row >= n_rows) return zh_set_error(ZH_EINVAL, "zh_ref_tree_encode: row %u has no uuid (%llu given)", row, (unsigned long long)n_rows);
                put(&sixteen, 8);
                put(uuids + 16 * (size_t)row, 16);
            }
        }
    }
    *out_len = pos;
    if (out && pos > cap) return zh_set_error(ZH_EINVAL, "zh_ref_tree_encode: buffer of %zu bytes, %zu needed", cap, pos);
    return ZH_OK;
}

static bool metric_has_power(int metric) { return metric == ZH_MINKOWSKI || metric == ZH_PNORM; }

extern "C" int zh_ref_header_decode(const uint8_t *bytes, size_t len, int metric, size_t model_len, zh_ref_header *out) {
    if (!bytes || !out) return zh_set_error(ZH_EINVAL, "zh_ref_header_decode: null argument");
    if (metric < ZH_COSINE || metric > ZH_PNORM) return zh_set_error(ZH_EINVAL, "zh_ref_header_decode: unknown metric %d", metric);
    Reader r{bytes, len};
    zh_ref_header h;
    memset(&h, 0, sizeof h);
    uint64_t blen;
    if (!r.take(&blen, 8) || blen != 16 || !r.take(h.uuid, 16)) return zh_set_error(ZH_EINVAL, "header: bad uuid at byte %zu", r.pos);
    if (len - r.pos < model_len) return zh_set_error(ZH_EINVAL, "header: truncated inside the model (%zu bytes stated)", model_len);
    h.model_off = r.pos; h.model_len = model_len;
    r.pos += model_len;
    h.metric = metric;
    if (metric_has_power(metric) && !r.take(&h.power, 4)) return zh_set_error(ZH_EINVAL, "header: truncated at byte %zu (metric power)", r.pos);
    if (!r.take(&h.max_node_size, 8) || !r.take(&h.num_trees, 8)) return zh_set_error(ZH_EINVAL, "header: truncated at byte %zu (index options)", r.pos);
    if (r.pos != r.n) return zh_set_error(ZH_EINVAL, "header: %zu trailing bytes (wrong metric or model type stated?)", r.n - r.pos);
    *out = h;
    return ZH_OK;
}

extern "C" int zh_ref_header_encode(const zh_ref_header *h, const uint8_t *model_bytes, uint8_t *out, size_t cap, size_t *out_len) {
    if (!h || !out_len || (h->model_len && !model_bytes)) return zh_set_error(ZH_EINVAL, "zh_ref_header_encode: null argument");
    if (h->metric < ZH_COSINE || h->metric > ZH_PNORM) return zh_set_error(ZH_EINVAL, "zh_ref_header_encode: unknown metric %d", h->metric);
    size_t pos = 0;
    auto put = [&](const void *src, size_t k) {
        if (out && pos + k <= cap) memcpy(out + pos, src, k);
        pos += k;
    };
    const uint64_t sixteen = 16;
    put(&sixteen, 8);
    put(h->uuid, 16);
    if (h->model_len) put(model_bytes, (size_t)h->model_len);
    if (metric_has_power(h->metric)) put(&h->power, 4);
    put(&h->max_node_size, 8);
    put(&h->num_trees, 8);
    *out_len = pos;
    if (out && pos > cap) return zh_set_error(ZH_EINVAL, "zh_ref_header_encode: buffer of %zu bytes, %zu needed", cap, pos);
    return ZH_OK;
}
