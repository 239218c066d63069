// zebra.hpp -- C++17 host-side mirror of the reference crate's interface for the hot path, over the C ABI
// of zebra_hip.h.  The reference is Rust; no Rust toolchain exists in the build image, so the host layer
// above the ABI is written in C++ with the crate's names, argument meaning and error behaviour
// (anyhow::Result<T> -> zebra::Error thrown).
//
//   reference                                   here
//   Embedding<N>            src/lib.rs:15-46                     zebra::Embedding<N>
//   EmbeddingPrecision=f32  src/lib.rs:48                        zebra::EmbeddingPrecision
//   DistanceUnit = u64      src/distance.rs:13                   zebra::DistanceUnit
//   space::Metric<Embedding<N>>::distance  src/distance.rs:19-21 Met::distance(a, b) const
//   CosineDistance<N>       src/distance.rs:15-32                zebra::CosineDistance<N>
//   L2SquaredDistance<N>    src/distance.rs:34-49                zebra::L2SquaredDistance<N>
//   L2Distance<N>           src/distance.rs:99-114               zebra::L2Distance<N>
//   LSHIndexOptions<N>      src/database/index/lsh.rs:122-139    zebra::LSHIndexOptions<N>
//   LSHIndex<N>             src/database/index/lsh.rs:144-565    zebra::LSHIndex<N>
//   Database<N,Met,Mod>     src/database/core.rs:45-313          zebra::Database<N,Met> (insert_records/query_vectors)
//
// Ids are dense row numbers (uint64_t) where the crate uses Uuid v7 (lsh.rs:415); a shim that needs Uuids
// keeps a row -> Uuid table.
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "zebra_hip.h"

namespace zebra {

using EmbeddingPrecision = float;   // lib.rs:48
using DistanceUnit = std::uint64_t; // distance.rs:13
using Id = std::uint64_t;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
    if (rc != ZH_OK) throw Error(rc, zh_last_error());
}

template <std::size_t N>
struct Embedding : std::array<EmbeddingPrecision, N> {  // lib.rs:18; Default = zeros (lib.rs:30-34)
    Embedding() { this->fill(0.0f); }
    explicit Embedding(const std::array<EmbeddingPrecision, N> &a) : std::array<EmbeddingPrecision, N>(a) {}
};

// ---- src/distance.rs: zero-sized metric structs implementing space::Metric with Unit = u64 ------------
namespace detail {
template <std::size_t N>
DistanceUnit metric_pair(int metric, int mode, const Embedding<N> &a, const Embedding<N> &b, int device) {
    DistanceUnit out = 0;
    check(zh_distance_pair(metric, mode, a.data(), b.data(), N, &out, device));
    return out;
}
}  // namespace detail

template <std::size_t N>
struct CosineDistance {
    // parity = true keeps distance.rs:23-25 literally (key = bits of 1.0 - simsimd cosine distance)
    bool parity = true;
    int device = -1;
    static constexpr int metric = ZH_COSINE;
    int mode() const { return parity ? ZH_COSINE_PARITY : ZH_COSINE_CORRECTED; }
    DistanceUnit distance(const Embedding<N> &a, const Embedding<N> &b) const {
        return detail::metric_pair<N>(metric, mode(), a, b, device);
    }
};
template <std::size_t N>
struct L2SquaredDistance {
    int device = -1;
    static constexpr int metric = ZH_L2SQ;
    int mode() const { return ZH_COSINE_PARITY; }
    DistanceUnit distance(const Embedding<N> &a, const Embedding<N> &b) const {
        return detail::metric_pair<N>(metric, mode(), a, b, device);
    }
};
template <std::size_t N>
struct L2Distance {
    int device = -1;
    static constexpr int metric = ZH_L2;
    int mode() const { return ZH_COSINE_PARITY; }
    DistanceUnit distance(const Embedding<N> &a, const Embedding<N> &b) const {
        return detail::metric_pair<N>(metric, mode(), a, b, device);
    }
};

// the `distances`-crate metrics (distance.rs:51-98,116-190): key = f32 bits widened; Hamming an integer count
#define ZEBRA_SIMPLE_METRIC(NAME, CODE)                                                         \
    template <std::size_t N>                                                                    \
    struct NAME {                                                                               \
        int device = -1;                                                                        \
        static constexpr int metric = CODE;                                                     \
        int mode() const { return 0; }                                                          \
        DistanceUnit distance(const Embedding<N> &a, const Embedding<N> &b) const {             \
            return detail::metric_pair<N>(metric, 0, a, b, device);                             \
        }                                                                                       \
    };
ZEBRA_SIMPLE_METRIC(ChebyshevDistance, ZH_CHEBYSHEV)
ZEBRA_SIMPLE_METRIC(CanberraDistance, ZH_CANBERRA)
ZEBRA_SIMPLE_METRIC(BrayCurtisDistance, ZH_BRAY_CURTIS)
ZEBRA_SIMPLE_METRIC(ManhattanDistance, ZH_MANHATTAN)
ZEBRA_SIMPLE_METRIC(L3Distance, ZH_L3)
ZEBRA_SIMPLE_METRIC(L4Distance, ZH_L4)
ZEBRA_SIMPLE_METRIC(HammingDistance, ZH_HAMMING)
#undef ZEBRA_SIMPLE_METRIC
template <std::size_t N>
struct MinkowskiDistance {  // distance.rs:160-174; #[derive(Default)] -> power 0, any i32 is legal
    int power = 0;
    int device = -1;
    static constexpr int metric = ZH_MINKOWSKI;
    int mode() const { return power; }
    DistanceUnit distance(const Embedding<N> &a, const Embedding<N> &b) const {
        return detail::metric_pair<N>(metric, power, a, b, device);
    }
};
template <std::size_t N>
struct PNormDistance {  // distance.rs:176-190; #[derive(Default)] -> power 0, any i32 is legal
    int power = 0;
    int device = -1;
    static constexpr int metric = ZH_PNORM;
    int mode() const { return power; }
    DistanceUnit distance(const Embedding<N> &a, const Embedding<N> &b) const {
        return detail::metric_pair<N>(metric, power, a, b, device);
    }
};

// ---- src/database/index/lsh.rs ------------------------------------------------------------------------
template <std::size_t N>
struct LSHIndexOptions {  // lsh.rs:122-139
    std::size_t max_node_size = 5;
    std::size_t num_trees = 15;
};

template <std::size_t N>
class LSHIndex {  // Clone in the crate shares the store (lsh.rs:144-148): copies share one zh_index
  public:
    // LSHIndex::new (lsh.rs:162-167); seed/device/id_base are the knobs the crate does not have
    explicit LSHIndex(const LSHIndexOptions<N> &options = {}, std::uint64_t seed = 0x5EB2A003ull, int device = -1,
                      std::uint64_t id_base = 0) {
        zh_options o;
        zh_options_default(&o);
        o.dim = (std::uint32_t)N;
        o.max_node_size = (std::uint32_t)options.max_node_size;
        o.num_trees = (std::uint32_t)options.num_trees;
        o.seed = seed;
        o.device = device;
        o.id_base = id_base;
        zh_index *h = nullptr;
        check(zh_index_create(&o, &h));
        h_.reset(h, zh_index_destroy);
    }
    void save() const {}  // persistence is out of scope (lsh.rs:170-172)

    bool no_vectors() const { return zh_index_count(h_.get()) == 0; }     // lsh.rs:398-400
    bool no_trees() const { return zh_index_num_trees(h_.get()) == 0; }   // lsh.rs:407-409
    bool is_empty() const { return no_vectors() || no_trees(); }          // lsh.rs:389-391

    // lsh.rs:440-466 -> ids of the added vectors
    std::vector<Id> add(const std::vector<Embedding<N>> &embeddings) const {
        std::vector<Id> ids(embeddings.size());
        check(zh_index_add(h_.get(), embeddings.empty() ? nullptr : embeddings[0].data(), embeddings.size(), ids.data()));
        return ids;
    }
    void clear() const { check(zh_index_clear(h_.get())); }  // lsh.rs:506-529
    // lsh.rs:473-503 (as intended: the ids leave every tree) -> the ids that were present
    std::vector<Id> remove(const std::vector<Id> &embedding_ids) const {
        std::vector<std::uint8_t> found(embedding_ids.size());
        std::size_t n = 0;
        check(zh_index_remove(h_.get(), embedding_ids.data(), embedding_ids.size(), found.data(), &n));
        std::vector<Id> out;
        for (std::size_t i = 0; i < found.size(); i++) if (found[i]) out.push_back(embedding_ids[i]);
        return out;
    }
    // lsh.rs:270-288 -> ids removed because an earlier vector has the same bits
    std::vector<Id> deduplicate() const {
        std::vector<Id> out(zh_index_count(h_.get()) + 1);
        std::size_t n = 0;
        check(zh_index_deduplicate(h_.get(), out.data(), out.size(), &n));
        out.resize(n < out.size() ? n : out.size());
        return out;
    }

    // lsh.rs:544-565: approximate k nearest neighbours, ascending by (distance key, id)
    template <class Met>
    std::vector<std::pair<Id, DistanceUnit>> search(const Embedding<N> &query, std::size_t top_k, const Met &metric) const {
        return std::move(search_batch(std::vector<Embedding<N>>{query}, top_k, metric)[0]);
    }
    // the rayon loop of core.rs:299-303 as one call
    template <class Met>
    std::vector<std::vector<std::pair<Id, DistanceUnit>>> search_batch(const std::vector<Embedding<N>> &queries,
                                                                         std::size_t top_k, const Met &metric) const {
        const std::size_t b = queries.size();
        std::vector<Id> ids(b * top_k);
        std::vector<DistanceUnit> keys(b * top_k);
        std::vector<std::uint32_t> counts(b);
        check(zh_search_batch(h_.get(), b ? queries[0].data() : nullptr, b, top_k, Met::metric, metric.mode(), ids.data(),
                              keys.data(), counts.data()));
        std::vector<std::vector<std::pair<Id, DistanceUnit>>> out(b);
        for (std::size_t i = 0; i < b; i++)
            for (std::uint32_t j = 0; j < counts[i]; j++) out[i].emplace_back(ids[i * top_k + j], keys[i * top_k + j]);
        return out;
    }
    zh_index *handle() const { return h_.get(); }

  private:
    std::shared_ptr<zh_index> h_;
};

// ---- src/database/core.rs (the two calls on the hot path; documents live in memory) -----------------------
template <std::size_t N, class Met>
class Database {
  public:
    explicit Database(const LSHIndexOptions<N> &index_options = {}, Met metric = Met{}) : index(index_options), metric_(metric) {}
    LSHIndex<N> index;  // pub field (core.rs:62)

    // core.rs:245-254
    void insert_records(const std::vector<Embedding<N>> &embeddings, const std::vector<std::string> &documents) {
        auto ids = index.add(embeddings);
        for (std::size_t i = 0; i < ids.size() && i < documents.size(); i++) documents_[ids[i]] = documents[i];
    }
    // core.rs:205-214 / 216-225 / 194-198
    void remove(const std::vector<Id> &embedding_ids) {
        for (Id i : index.remove(embedding_ids)) documents_.erase(i);
    }
    void deduplicate() {
        for (Id i : index.deduplicate()) documents_.erase(i);
    }
    void clear_database() {
        index.clear();
        documents_.clear();
    }
    // core.rs:290-313: query index -> {id -> document}; order and distances are dropped (core.rs:304-305).
    // core.rs:303's unwrap_or_default turns a PER-QUERY search failure into an empty entry; here a batch is one call that
    // has no per-query failure mode (over-long batches are split inside zh_search_batch), so a library error -- out of
    // memory, no device, top_k > ZH_MAX_TOPK -- is thrown like in the Python mirror, never reported as "no neighbours"
    std::map<std::size_t, std::map<Id, std::string>> query_vectors(const std::vector<Embedding<N>> &vectors,
                                                                   std::size_t number_of_results) const {
        std::map<std::size_t, std::map<Id, std::string>> results;
        if (index.no_vectors()) return results;  // core.rs:295-297
        const auto nb = index.search_batch(vectors, number_of_results, metric_);
        for (std::size_t i = 0; i < nb.size(); i++) {
            auto &m = results[i];
            for (auto &p : nb[i]) {
                auto it = documents_.find(p.first);
                m[p.first] = it == documents_.end() ? std::string() : it->second;
            }
        }
        return results;
    }

  private:
    Met metric_;
    std::map<Id, std::string> documents_;
};

}  // namespace zebra
