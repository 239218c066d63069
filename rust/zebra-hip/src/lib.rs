//! Shim that keeps the reference crate's signatures for the hot path on top of `libzebra_hip.so`.
//!
//! NOT COMPILED in this repository's build image (no Rust toolchain there); it mirrors
//! `include/zebra_hip.h` one to one and is kept mechanical on purpose.  It is a PATCH to the crate, not a parallel crate:
//! without the `standalone` feature it binds the crate's own `Embedding<N>` / `DistanceUnit` (see below).  What it replaces in emmyoh/zebra:
//! `src/distance.rs` (the metric structs' `Metric::distance`), `src/database/index/lsh.rs`
//! (`LSHIndex<N>`: new / add / search / remove / deduplicate / clear / is_empty / no_vectors / no_trees) and the
//! rayon loop of `Database::query_vectors` (`src/database/core.rs:299-303`) through `search_batch`.
#![allow(non_camel_case_types)]

use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};
use std::sync::{Arc, RwLock};

use dashmap::{DashMap, DashSet};
use space::Metric;
use uuid::Uuid;

// ---- the vector type and the key type: the CRATE'S OWN (round 4) -----------------------------------------------------------
// As a patch to emmyoh/zebra this file lives INSIDE the crate (src/hip.rs, feature `hip`) and binds the
// crate's `Embedding<N>` / `DistanceUnit` -- it does not re-declare them: one type, no conversion at the boundary, and a
// `&[Embedding<N>]` handed to `insert_records` / `query_vectors` crosses the FFI as it is.  The stand-alone re-declaration
// below exists only behind the `standalone` feature, for building and testing this shim without the crate.
#[cfg(not(feature = "standalone"))]
pub use crate::{Embedding, EmbeddingPrecision}; // src/lib.rs:15-48 (as `zebra::...` when built as an external crate: `--cfg zebra_external`)
#[cfg(not(feature = "standalone"))]
pub use crate::distance::DistanceUnit; // src/distance.rs:13
// The reference's `Embedding<N>` is a plain newtype over `[f32; N]` (src/lib.rs:18) WITHOUT `#[repr(transparent)]`; the layout of a
// single-field struct is that of its field in practice, and this shim asserts what it relies on instead of assuming it:
#[cfg(not(feature = "standalone"))]
const _: () = {
    assert!(std::mem::size_of::<Embedding<4>>() == 16 && std::mem::align_of::<Embedding<4>>() == 4);
};

#[cfg(feature = "standalone")]
pub type DistanceUnit = u64; // src/distance.rs:13
#[cfg(feature = "standalone")]
pub type EmbeddingPrecision = f32; // src/lib.rs:48

/// (feature `standalone` only) `Embedding<N>` as in the reference (src/lib.rs:15-46): a newtype over `[f32; N]` with Deref / DerefMut /
/// Default / From / TryFrom.  `repr(transparent)`: a `&[Embedding<N>]` is `len * N` contiguous f32, which is what crosses the FFI.
#[cfg(feature = "standalone")]
#[repr(transparent)]
#[derive(Debug, Clone)]
pub struct Embedding<const N: usize>([EmbeddingPrecision; N]);
#[cfg(feature = "standalone")]
impl<const N: usize> std::ops::Deref for Embedding<N> {
    type Target = [EmbeddingPrecision; N];
    fn deref(&self) -> &[EmbeddingPrecision; N] {
        &self.0
    }
}
#[cfg(feature = "standalone")]
impl<const N: usize> std::ops::DerefMut for Embedding<N> {
    fn deref_mut(&mut self) -> &mut [EmbeddingPrecision; N] {
        &mut self.0
    }
}
#[cfg(feature = "standalone")]
impl<const N: usize> Default for Embedding<N> {
    fn default() -> Self {
        Self([0.0; N])
    }
}
#[cfg(feature = "standalone")]
impl<const N: usize> From<[EmbeddingPrecision; N]> for Embedding<N> {
    fn from(value: [EmbeddingPrecision; N]) -> Self {
        Self(value)
    }
}
#[cfg(feature = "standalone")]
impl<const N: usize> TryFrom<Vec<EmbeddingPrecision>> for Embedding<N> {
    type Error = Vec<EmbeddingPrecision>;
    fn try_from(value: Vec<EmbeddingPrecision>) -> Result<Self, Self::Error> {
        Ok(Self(value.try_into()?))
    }
}

pub mod ffi {
    use super::*;

    #[repr(C)]
    pub struct zh_index {
        _private: [u8; 0],
    }

    #[repr(C)]
    #[derive(Clone, Copy)]
    pub struct zh_options {
        pub dim: u32,
        pub max_node_size: u32, // LSHIndexOptions::max_node_size (lsh.rs:126)
        pub num_trees: u32,     // LSHIndexOptions::num_trees     (lsh.rs:128)
        pub seed: u64,
        pub device: i32,
        pub id_base: u64,
        pub reserve_rows: u64,
    }

    pub const ZH_COSINE: c_int = 0;
    pub const ZH_L2SQ: c_int = 1;
    pub const ZH_L2: c_int = 2;
    pub const ZH_CHEBYSHEV: c_int = 3;
    pub const ZH_CANBERRA: c_int = 4;
    pub const ZH_BRAY_CURTIS: c_int = 5;
    pub const ZH_MANHATTAN: c_int = 6;
    pub const ZH_L3: c_int = 7;
    pub const ZH_L4: c_int = 8;
    pub const ZH_HAMMING: c_int = 9;
    pub const ZH_MINKOWSKI: c_int = 10;
    pub const ZH_PNORM: c_int = 11;
    pub const ZH_COSINE_PARITY: c_int = 0; // distance.rs:23-25 literally
    pub const ZH_COSINE_CORRECTED: c_int = 1;

    extern "C" {
        pub fn zh_options_default(opt: *mut zh_options);
        pub fn zh_index_create(opt: *const zh_options, out: *mut *mut zh_index) -> c_int;
        pub fn zh_index_destroy(idx: *mut zh_index);
        pub fn zh_index_clear(idx: *mut zh_index) -> c_int;
        pub fn zh_index_add(idx: *mut zh_index, rows: *const f32, n: usize, out_row_ids: *mut u64) -> c_int;
        pub fn zh_index_build(idx: *mut zh_index) -> c_int;
        pub fn zh_index_remove(idx: *mut zh_index, ids: *const u64, n: usize, out_found: *mut u8, out_n_removed: *mut usize) -> c_int;
        pub fn zh_index_deduplicate(idx: *mut zh_index, out_ids: *mut u64, cap: usize, out_n_removed: *mut usize) -> c_int;
        pub fn zh_index_count(idx: *const zh_index) -> u64;
        pub fn zh_index_num_trees(idx: *const zh_index) -> u32;
        pub fn zh_search_batch(idx: *mut zh_index, q: *const f32, b: usize, k: usize, metric: c_int, cosine_mode: c_int,
                               out_ids: *mut u64, out_keys: *mut u64, out_counts: *mut u32) -> c_int;
        pub fn zh_search_batch_device(idx: *mut zh_index, d_q: *const f32, b: usize, k: usize, metric: c_int, cosine_mode: c_int,
                                      d_out_ids: *mut u64, d_out_keys: *mut u64, d_out_counts: *mut u32, stream: *mut c_void) -> c_int;
        pub fn zh_distance_pair(metric: c_int, cosine_mode: c_int, a: *const f32, b: *const f32, dim: usize, out_key: *mut u64,
                                device: c_int) -> c_int;
        pub fn zh_merge_topk_device(device: c_int, n_shards: u32, b: usize, k: usize, d_ids: *const u64, d_keys: *const u64,
                                    d_counts: *const u32, d_out_ids: *mut u64, d_out_keys: *mut u64, d_out_counts: *mut u32,
                                    stream: *mut c_void) -> c_int;
        // the reference's on-disk tree values <-> flat forest; host code, no GPU
        pub fn zh_ref_forest_decode(dim: u32, n_trees: usize, values: *const *const u8, lens: *const usize, n_vectors: usize,
                                    uuids: *const u8, out: *mut *mut c_void, out_unknown_ids: *mut u64) -> c_int;
        pub fn zh_ref_forest_view(forest: *const c_void, out: *mut c_void /* zh_forest_view */) -> c_int;
        pub fn zh_ref_forest_free(forest: *mut c_void);
        pub fn zh_ref_tree_encode(forest: *const c_void /* zh_forest_view */, dim: u32, tree: u32, uuids: *const u8, n_rows: u64,
                                  out: *mut u8, cap: usize, out_len: *mut usize) -> c_int;
        pub fn zh_last_error() -> *const c_char;
        pub fn zh_trim_device_memory() -> c_int;
    }
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::zh_last_error()) }.to_string_lossy().into_owned()
}

fn check(rc: c_int) -> anyhow::Result<()> {
    if rc == 0 {
        return Ok(());
    }
    Err(anyhow::anyhow!("zebra_hip error {rc}: {}", last_error()))
}

/// Owns the device-side index.  `LSHIndex` is `Clone` in the reference (lsh.rs:144): clones share it.
struct HipIndex(*mut ffi::zh_index);
unsafe impl Send for HipIndex {}
unsafe impl Sync for HipIndex {} // search calls serialise inside the library
impl Drop for HipIndex {
    fn drop(&mut self) {
        unsafe { ffi::zh_index_destroy(self.0) }
    }
}

/// What a metric struct tells the index so that the batched search can run it on the device.  Every metric struct of
/// `src/distance.rs` implements it, so every call site of the reference compiles unchanged against
/// `search<Met: Metric<Embedding<N>, Unit = DistanceUnit> + HipMetric + Send + Sync>` -- the reference's own bound
/// (lsh.rs:544-549) plus this one marker, which is the only addition to any public signature.
pub trait HipMetric {
    const METRIC: c_int;
    /// cosine mode for `CosineDistance`, `power` for Minkowski / p-norm, ignored otherwise
    fn param(&self) -> c_int {
        0
    }
}

macro_rules! simple_metric {
    ($name:ident, $code:expr, $doc:expr) => {
        #[doc = $doc]
        #[derive(Default, Debug, Clone)]
        pub struct $name<const N: usize>;
        impl<const N: usize> HipMetric for $name<N> {
            const METRIC: c_int = $code;
        }
        impl<const N: usize> Metric<Embedding<N>> for $name<N> {
            type Unit = DistanceUnit;
            fn distance(&self, a: &Embedding<N>, b: &Embedding<N>) -> DistanceUnit {
                // `Metric::distance` has no error channel (distance.rs:19-21): a failing device call must not read as key 0
                let mut key = 0u64;
                let rc = unsafe { ffi::zh_distance_pair($code, 0, a.as_ptr(), b.as_ptr(), N, &mut key, -1) };
                if rc != 0 {
                    panic!("zh_distance_pair({}): {}", stringify!($name), last_error());
                }
                key
            }
        }
    };
}
simple_metric!(CosineDistance, ffi::ZH_COSINE, "src/distance.rs:15-32 (key = bits of `1.0 - simsimd cosine`, literally)");
simple_metric!(L2SquaredDistance, ffi::ZH_L2SQ, "src/distance.rs:34-49");
simple_metric!(L2Distance, ffi::ZH_L2, "src/distance.rs:99-114");
simple_metric!(ChebyshevDistance, ffi::ZH_CHEBYSHEV, "src/distance.rs:51-61");
simple_metric!(CanberraDistance, ffi::ZH_CANBERRA, "src/distance.rs:63-73");
simple_metric!(BrayCurtisDistance, ffi::ZH_BRAY_CURTIS, "src/distance.rs:75-85");
simple_metric!(ManhattanDistance, ffi::ZH_MANHATTAN, "src/distance.rs:87-97");
simple_metric!(L3Distance, ffi::ZH_L3, "src/distance.rs:116-126");
simple_metric!(L4Distance, ffi::ZH_L4, "src/distance.rs:128-138");
simple_metric!(HammingDistance, ffi::ZH_HAMMING, "src/distance.rs:140-158");

/// src/distance.rs:160-174
#[derive(Default, Debug, Clone)]
pub struct MinkowskiDistance<const N: usize> {
    pub power: i32,
}
impl<const N: usize> HipMetric for MinkowskiDistance<N> {
    const METRIC: c_int = ffi::ZH_MINKOWSKI;
    fn param(&self) -> c_int {
        self.power
    }
}
/// src/distance.rs:176-190
#[derive(Default, Debug, Clone)]
pub struct PNormDistance<const N: usize> {
    pub power: i32,
}
impl<const N: usize> HipMetric for PNormDistance<N> {
    const METRIC: c_int = ffi::ZH_PNORM;
    fn param(&self) -> c_int {
        self.power
    }
}
macro_rules! power_metric {
    ($name:ident, $code:expr) => {
        impl<const N: usize> Metric<Embedding<N>> for $name<N> {
            type Unit = DistanceUnit;
            fn distance(&self, a: &Embedding<N>, b: &Embedding<N>) -> DistanceUnit {
                // every i32 power is served (the derived Default is 0); what can still fail is the device, and
                // `Metric::distance` has no error channel: fail loudly instead of returning key 0 for every pair
                let mut key = 0u64;
                let rc = unsafe { ffi::zh_distance_pair($code, self.power, a.as_ptr(), b.as_ptr(), N, &mut key, -1) };
                if rc != 0 {
                    panic!("zh_distance_pair({}, power {}): {}", stringify!($name), self.power, last_error());
                }
                key
            }
        }
    };
}
power_metric!(MinkowskiDistance, ffi::ZH_MINKOWSKI);
power_metric!(PNormDistance, ffi::ZH_PNORM);

/// lsh.rs:122-139
#[derive(Debug, Clone)]
pub struct LSHIndexOptions<const N: usize> {
    pub max_node_size: usize,
    pub num_trees: usize,
}
impl<const N: usize> Default for LSHIndexOptions<N> {
    fn default() -> Self {
        Self { max_node_size: 5, num_trees: 15 }
    }
}

/// `LSHIndex<N>` with the reference's public signatures (lsh.rs:144-565); ids are Uuids kept in a row table.
#[derive(Clone)]
pub struct LSHIndex<const N: usize> {
    hip: Arc<HipIndex>,
    ids: Arc<RwLock<IdTable>>,
}

/// row <-> Uuid (the library's ids are dense row numbers; Uuid::now_v7 at add time as in lsh.rs:415)
#[derive(Default)]
struct IdTable {
    of_row: Vec<Uuid>,
    row_of: HashMap<Uuid, u64>,
}

impl<const N: usize> LSHIndex<N> {
    pub fn new(_uuid: &Uuid, options: &LSHIndexOptions<N>) -> anyhow::Result<Self> {
        let mut o = unsafe { std::mem::zeroed::<ffi::zh_options>() };
        unsafe { ffi::zh_options_default(&mut o) };
        o.dim = N as u32;
        o.max_node_size = options.max_node_size as u32;
        o.num_trees = options.num_trees as u32;
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::zh_index_create(&o, &mut h) })?;
        Ok(Self { hip: Arc::new(HipIndex(h)), ids: Default::default() })
    }
    pub fn save(&self) -> anyhow::Result<()> {
        Ok(())
    }
    pub fn no_vectors(&self) -> bool {
        unsafe { ffi::zh_index_count(self.hip.0) == 0 }
    }
    pub fn no_trees(&self) -> bool {
        unsafe { ffi::zh_index_num_trees(self.hip.0) == 0 }
    }
    pub fn is_empty(&self) -> bool {
        self.no_vectors() || self.no_trees()
    }

    /// lsh.rs:440-466
    pub fn add(&self, embeddings: &Vec<Embedding<N>>) -> anyhow::Result<Vec<Uuid>> {
        let ids: Vec<Uuid> = embeddings.iter().map(|_| Uuid::now_v7()).collect();
        let mut t = self.ids.write().unwrap(); // rows are numbered in insertion order: hold the table across the call
        // The C layer stores the rows BEFORE it touches the trees and writes a row's id only once the row is stored; a failure
        // after that (a first build or an incremental insert that runs out of memory) keeps the rows -- their ids were handed
        // out.  So the row -> Uuid table follows what was STORED, not what succeeded: otherwise every later row would be
        // shifted against its Uuid (or `of_row[id]` would panic) once the index is rebuilt with zh_index_build.
        let mut rows = vec![u64::MAX; embeddings.len()];
        let rc = unsafe { ffi::zh_index_add(self.hip.0, embeddings.as_ptr() as *const f32, embeddings.len(), rows.as_mut_ptr()) };
        for (u, r) in ids.iter().zip(rows.iter()) {
            if *r != u64::MAX {
                t.row_of.insert(*u, t.of_row.len() as u64);
                t.of_row.push(*u);
            }
        }
        check(rc)?;
        Ok(ids)
    }

    /// lsh.rs:544-565
    pub fn search<Met: Metric<Embedding<N>, Unit = DistanceUnit> + HipMetric + Send + Sync>(
        &self,
        query: &Embedding<N>,
        top_k: usize,
        metric: &Met,
    ) -> anyhow::Result<Vec<(Uuid, DistanceUnit)>> {
        Ok(self.search_batch(std::slice::from_ref(query), top_k, metric)?.pop().unwrap_or_default())
    }

    /// the rayon loop of core.rs:299-303 as one call
    pub fn search_batch<Met: Metric<Embedding<N>, Unit = DistanceUnit> + HipMetric + Send + Sync>(
        &self,
        queries: &[Embedding<N>],
        top_k: usize,
        metric: &Met,
    ) -> anyhow::Result<Vec<Vec<(Uuid, DistanceUnit)>>> {
        let b = queries.len();
        let (mut ids, mut keys, mut counts) = (vec![0u64; b * top_k], vec![0u64; b * top_k], vec![0u32; b]);
        check(unsafe {
            ffi::zh_search_batch(self.hip.0, queries.as_ptr() as *const f32, b, top_k, Met::METRIC, metric.param(), ids.as_mut_ptr(),
                                 keys.as_mut_ptr(), counts.as_mut_ptr())
        })?;
        let t = self.ids.read().unwrap();
        Ok((0..b).map(|i| (0..counts[i] as usize).map(|j| (t.of_row[ids[i * top_k + j] as usize], keys[i * top_k + j])).collect()).collect())
    }

    /// lsh.rs:473-503 (as intended: the ids leave every tree)
    pub fn remove(&self, embedding_ids: &Vec<Uuid>) -> anyhow::Result<DashSet<Uuid>> {
        let t = self.ids.read().unwrap();
        let rows: Vec<u64> = embedding_ids.iter().filter_map(|u| t.row_of.get(u).copied()).collect(); // O(1) per id
        let mut found = vec![0u8; rows.len()];
        check(unsafe { ffi::zh_index_remove(self.hip.0, rows.as_ptr(), rows.len(), found.as_mut_ptr(), std::ptr::null_mut()) })?;
        let removed = DashSet::new();
        for (r, f) in rows.iter().zip(found) {
            if f != 0 {
                removed.insert(t.of_row[*r as usize]);
            }
        }
        Ok(removed)
    }

    /// lsh.rs:270-288
    pub fn deduplicate(&self) -> anyhow::Result<DashSet<Uuid>> {
        let cap = unsafe { ffi::zh_index_count(self.hip.0) } as usize + 1;
        let (mut out, mut n) = (vec![0u64; cap], 0usize);
        check(unsafe { ffi::zh_index_deduplicate(self.hip.0, out.as_mut_ptr(), cap, &mut n) })?;
        let t = self.ids.read().unwrap();
        Ok(out[..n.min(cap)].iter().map(|r| t.of_row[*r as usize]).collect())
    }

    /// lsh.rs:506-529
    pub fn clear(&self) -> anyhow::Result<()> {
        let mut t = self.ids.write().unwrap();
        t.of_row.clear();
        t.row_of.clear();
        check(unsafe { ffi::zh_index_clear(self.hip.0) })
    }
}

/// The model side of `Database<N, Met, Mod>` (src/model/core.rs:12-37) is untouched by this path; a marker stands in.
pub trait DatabaseEmbeddingModel<const N: usize> {}

/// `Database<N, Met, Mod>` (src/database/core.rs:55-62) for the two calls on the hot path.  In the reference the
/// struct also carries the header (`DatabaseInner`) and a path, and documents live in lz4 files
/// (`save_documents_to_disk` / `read_documents_from_disk`, core.rs:322-380): unchanged code, represented here by an
/// in-memory map so that the skeleton is self-contained.
#[derive(Clone)]
pub struct Database<
    const N: usize,
    Met: Metric<Embedding<N>, Unit = DistanceUnit> + HipMetric + Default + Send + Sync,
    Mod: DatabaseEmbeddingModel<N> + Default + Send + Sync,
> {
    metric: Met,
    #[allow(dead_code)]
    model: Mod,
    /// The database index used to approximate nearest-neighbour search (pub field, core.rs:62).
    pub index: LSHIndex<N>,
    documents: Arc<DashMap<Uuid, Vec<u8>>>,
}

impl<
        const N: usize,
        Met: Metric<Embedding<N>, Unit = DistanceUnit> + HipMetric + Default + Send + Sync,
        Mod: DatabaseEmbeddingModel<N> + Default + Send + Sync,
    > Database<N, Met, Mod>
{
    pub fn new(uuid: &Uuid, index_options: &LSHIndexOptions<N>) -> anyhow::Result<Self> {
        Ok(Self { metric: Met::default(), model: Mod::default(), index: LSHIndex::new(uuid, index_options)?, documents: Default::default() })
    }

    /// core.rs:245-254
    pub fn insert_records(&self, embeddings: &Vec<Embedding<N>>, documents: &Vec<Vec<u8>>) -> anyhow::Result<()> {
        let embedding_ids = self.index.add(embeddings)?;
        for (id, doc) in embedding_ids.iter().zip(documents) {
            self.documents.insert(*id, doc.clone());
        }
        Ok(())
    }

    /// core.rs:290-313: the rayon loop over queries (core.rs:299-303) is ONE batched call; order and distances are
    /// dropped as in core.rs:304-305.  A batch has no per-query failure mode, so a library error is returned, not
    /// turned into empty entries (the reference's `unwrap_or_default` is per query).
    pub fn query_vectors(&self, vectors: &Vec<Embedding<N>>, number_of_results: usize) -> anyhow::Result<DashMap<usize, DashMap<Uuid, Vec<u8>>>> {
        if self.index.no_vectors() {
            return Ok(DashMap::new()); // core.rs:295-297
        }
        let results = DashMap::new();
        for (idx, neighbours) in self.index.search_batch(vectors, number_of_results, &self.metric)?.into_iter().enumerate() {
            let docs = DashMap::new();
            for (id, _) in neighbours {
                docs.insert(id, self.documents.get(&id).map(|d| d.clone()).unwrap_or_default());
            }
            results.insert(idx, docs);
        }
        Ok(results)
    }

    /// core.rs:205-214
    pub fn remove(&self, embedding_ids: &Vec<Uuid>) -> anyhow::Result<()> {
        for id in self.index.remove(embedding_ids)?.iter() {
            self.documents.remove(&*id);
        }
        Ok(())
    }
}
