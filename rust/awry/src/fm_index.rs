//! `FmIndex` with AWRY 0.3.1's public API (reference: src/fm_index.rs:41-119,142,302-399,455-593 and
//! src/fm_index_file.rs:42,132), backed by `libawry_hip.so`.
//!
//! What differs from the reference, and why:
//! * every search runs on a GPU: `new` / `load` replicate the index onto GPU 0 (or the GPUs listed in
//!   `AWRY_DEVICES`, e.g. `0,1,2,3`); [`FmIndex::set_devices`] chooses others.  Batches are sharded over the replicas.
//! * queries on which the reference panics or is undefined (empty query, `$` / `#`, non-ASCII bytes) panic here too,
//!   with the library's message; the `try_*` variants return the error instead.
//! * hits of a multi-record index are localised with the intended semantics (largest record start <= position); the
//!   reference's `get_seq_location` does not terminate there.
//! * `FmIndex` is `Clone` (a cheap handle copy: the index is immutable) and `PartialEq` (same index content);
//!   `PartialOrd` / `Ord` / `Hash` / `MemSize` of the reference's derive list have no counterpart.
use std::ffi::{CStr, CString};
use std::os::raw::c_char;
use std::path::{Path, PathBuf};
use std::sync::Arc;

use awry_hip_sys as sys;
use rayon::iter::{IndexedParallelIterator, IntoParallelIterator, IntoParallelRefIterator, ParallelIterator};

use crate::{
    alphabet::{Symbol, SymbolAlphabet},
    search::{SearchPtr, SearchRange},
    sequence_index::LocalizedSequencePosition,
};

/// Error of a library call: the status class and the library's message.
#[derive(Debug, Clone)]
pub struct AwryError {
    pub code: i32,
    pub message: String,
}

impl std::fmt::Display for AwryError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "awry error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for AwryError {}

fn last_error(code: i32) -> AwryError {
    let message = unsafe { CStr::from_ptr(sys::awry_last_error()) }.to_string_lossy().into_owned();
    AwryError { code, message }
}

fn check(code: i32) -> Result<(), AwryError> {
    if code == sys::AWRY_OK {
        Ok(())
    } else {
        Err(last_error(code))
    }
}

fn to_io(e: AwryError) -> std::io::Error {
    let kind = match e.code {
        sys::AWRY_ERR_FORMAT => std::io::ErrorKind::InvalidData,
        sys::AWRY_ERR_ARG => std::io::ErrorKind::InvalidInput,
        _ => std::io::ErrorKind::Other,
    };
    std::io::Error::new(kind, e)
}

fn path_cstring(p: &Path) -> Result<CString, AwryError> {
    #[cfg(unix)]
    {
        use std::os::unix::ffi::OsStrExt;
        CString::new(p.as_os_str().as_bytes()).map_err(|_| AwryError { code: sys::AWRY_ERR_ARG, message: "path contains a NUL byte".into() })
    }
    #[cfg(not(unix))]
    {
        CString::new(p.to_string_lossy().as_bytes()).map_err(|_| AwryError { code: sys::AWRY_ERR_ARG, message: "path contains a NUL byte".into() })
    }
}

/// Owner of the C handle; freed when the last `FmIndex` clone goes away.
struct Handle(*mut sys::awry_index_t);
// The library's query entry points are re-entrant and the index is immutable after build / load / set_devices.
unsafe impl Send for Handle {}
unsafe impl Sync for Handle {}
impl Drop for Handle {
    fn drop(&mut self) {
        unsafe { sys::awry_free(self.0) }
    }
}

/// Primary FM-index struct.
///
/// ```no_run
/// use awry::fm_index::{FmIndex, FmBuildArgs};
/// use awry::alphabet::SymbolAlphabet;
///
/// let build_args = FmBuildArgs {
///     input_file_src: "test.fasta".into(),
///     suffix_array_output_src: None,
///     suffix_array_compression_ratio: None,
///     lookup_table_kmer_len: None,
///     alphabet: SymbolAlphabet::Nucleotide,
///     max_query_len: None,
///     remove_intermediate_suffix_array_file: false,
/// };
/// let fm_index = FmIndex::new(&build_args).expect("unable to build fm index");
/// ```
#[derive(Clone)]
pub struct FmIndex {
    handle: Arc<Handle>,
    prefix_sums: Vec<u64>,
}

impl std::fmt::Debug for FmIndex {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("FmIndex")
            .field("alphabet", &self.alphabet())
            .field("bwt_len", &self.bwt_len())
            .field("suffix_array_compression_ratio", &self.suffix_array_compression_ratio())
            .field("version_number", &self.version_number())
            .field("devices", &self.num_devices())
            .finish()
    }
}

impl PartialEq for FmIndex {
    /// Same index content: metadata, prefix sums, BWT blocks, sampled suffix array and sequence records
    /// (what the reference's `save_load_equality_test` compares field by field, src/fm_index.rs:1046-1088).
    fn eq(&self, other: &Self) -> bool {
        if Arc::ptr_eq(&self.handle, &other.handle) {
            return true;
        }
        let (a, b) = (self.raw() as *const sys::awry_index_t, other.raw() as *const sys::awry_index_t);
        unsafe {
            if sys::awry_alphabet(a) != sys::awry_alphabet(b)
                || sys::awry_bwt_len(a) != sys::awry_bwt_len(b)
                || sys::awry_version(a) != sys::awry_version(b)
                || sys::awry_sa_ratio(a) != sys::awry_sa_ratio(b)
                || sys::awry_kmer_len(a) != sys::awry_kmer_len(b)
                || self.prefix_sums != other.prefix_sums
            {
                return false;
            }
            let words = |f: unsafe extern "C" fn(*const sys::awry_index_t, *mut u64) -> *const u64, h: *const sys::awry_index_t| {
                let mut n = 0u64;
                let p = f(h, &mut n);
                std::slice::from_raw_parts(p, n as usize)
            };
            if words(sys::awry_block_words, a) != words(sys::awry_block_words, b) || words(sys::awry_sa_words, a) != words(sys::awry_sa_words, b) {
                return false;
            }
        }
        self.sequence_headers() == other.sequence_headers() && self.sequence_starts() == other.sequence_starts()
    }
}
impl Eq for FmIndex {}

/// Arguments for building an FM-index (reference: src/fm_index.rs:78-119).
#[derive(Debug, Clone)]
#[cfg_attr(feature = "serde", derive(serde::Serialize, serde::Deserialize))]
pub struct FmBuildArgs {
    /// file source for the input, either Fasta or Fastq format
    pub input_file_src: PathBuf,
    /// accepted for signature parity: no intermediate suffix-array file is written (the suffix array is built in memory,
    /// on the GPU when one is present)
    pub suffix_array_output_src: Option<PathBuf>,
    /// how much to downsample the suffix array (default 8)
    pub suffix_array_compression_ratio: Option<u64>,
    /// k-mer length of the lookup table stored in `.awry` files (default 10 nucleotide / 4 amino)
    pub lookup_table_kmer_len: Option<u8>,
    /// alphabet of the input text
    pub alphabet: SymbolAlphabet,
    /// accepted for signature parity: the full suffix array is always built
    pub max_query_len: Option<usize>,
    /// accepted for signature parity: there is no intermediate file to remove
    pub remove_intermediate_suffix_array_file: bool,
}

impl FmBuildArgs {
    pub fn new(
        input_file_src: PathBuf,
        suffix_array_output_src: Option<PathBuf>,
        suffix_array_compression_ratio: Option<u64>,
        lookup_table_kmer_len: Option<u8>,
        alphabet: SymbolAlphabet,
        max_query_len: Option<usize>,
        remove_intermediate_suffix_array_file: bool,
    ) -> Self {
        FmBuildArgs {
            input_file_src,
            suffix_array_output_src,
            suffix_array_compression_ratio,
            lookup_table_kmer_len,
            alphabet,
            max_query_len,
            remove_intermediate_suffix_array_file,
        }
    }
}

/// CSR form of a batch of queries: what `awry_count_batch` / `awry_locate_batch` take.
struct Csr {
    bytes: Vec<u8>,
    offsets: Vec<u64>,
}

fn to_csr<'a>(queries: impl ParallelIterator<Item = &'a str>) -> Csr {
    let list: Vec<&'a str> = queries.collect();
    let mut offsets = Vec::with_capacity(list.len() + 1);
    let mut at = 0u64;
    offsets.push(0);
    for q in &list {
        at += q.len() as u64;
        offsets.push(at);
    }
    let mut bytes = vec![0u8; at as usize];
    // disjoint slices of the byte buffer, filled in parallel
    let mut rest: &mut [u8] = &mut bytes;
    let mut parts: Vec<&mut [u8]> = Vec::with_capacity(list.len());
    for q in &list {
        let (head, tail) = std::mem::take(&mut rest).split_at_mut(q.len());
        parts.push(head);
        rest = tail;
    }
    parts.into_par_iter().zip(list.par_iter()).for_each(|(dst, q)| dst.copy_from_slice(q.as_bytes()));
    Csr { bytes, offsets }
}

fn default_devices() -> Vec<i32> {
    match std::env::var("AWRY_DEVICES") {
        Ok(s) => {
            let v: Vec<i32> = s.split(',').filter_map(|t| t.trim().parse().ok()).collect();
            if v.is_empty() { vec![0] } else { v }
        }
        Err(_) => vec![0],
    }
}

impl FmIndex {
    fn raw(&self) -> *mut sys::awry_index_t {
        self.handle.0
    }

    fn from_handle(h: *mut sys::awry_index_t) -> Result<Self, AwryError> {
        let handle = Arc::new(Handle(h));
        let mut n = 0u64;
        let p = unsafe { sys::awry_prefix_sums(h, &mut n) };
        let prefix_sums = unsafe { std::slice::from_raw_parts(p, n as usize) }.to_vec();
        let ix = FmIndex { handle, prefix_sums };
        ix.set_devices(&default_devices())?;
        Ok(ix)
    }

    /// Construct a new FM-index using the supplied build args (reference: src/fm_index.rs:142-268).
    pub fn new(args: &FmBuildArgs) -> Result<Self, anyhow::Error> {
        let input = path_cstring(&args.input_file_src)?;
        let sa_tmp = match &args.suffix_array_output_src {
            Some(p) => Some(path_cstring(p)?),
            None => None,
        };
        let c_args = sys::awry_build_args_t {
            input_path: input.as_ptr(),
            sa_tmp_path: sa_tmp.as_ref().map_or(std::ptr::null(), |s| s.as_ptr()),
            sa_ratio: args.suffix_array_compression_ratio.unwrap_or(0),
            kmer_len: args.lookup_table_kmer_len.unwrap_or(0),
            alphabet: args.alphabet.alphabet_id(),
            max_query_len: args.max_query_len.unwrap_or(0) as u64,
            remove_tmp: args.remove_intermediate_suffix_array_file as u8,
        };
        let mut h: *mut sys::awry_index_t = std::ptr::null_mut();
        check(unsafe { sys::awry_build(&c_args, &mut h) })?;
        Ok(Self::from_handle(h)?)
    }

    /// Loads an FM-index from an `.awry` v1 file (reference: src/fm_index_file.rs:132).
    pub fn load(fm_file_src: &Path) -> Result<FmIndex, std::io::Error> {
        let path = path_cstring(fm_file_src).map_err(to_io)?;
        let mut h: *mut sys::awry_index_t = std::ptr::null_mut();
        check(unsafe { sys::awry_load(path.as_ptr(), &mut h) }).map_err(to_io)?;
        Self::from_handle(h).map_err(to_io)
    }

    /// Saves the FM-index to an `.awry` v1 file, byte-compatible with the reference's (src/fm_index_file.rs:42).
    pub fn save(&self, file_output_src: &Path) -> Result<(), std::io::Error> {
        let path = path_cstring(file_output_src).map_err(to_io)?;
        check(unsafe { sys::awry_save(self.raw(), path.as_ptr()) }).map_err(to_io)
    }

    /// Replicates the index onto the listed GPUs (replacing earlier replicas); batches are sharded contiguously over
    /// them.  Stands in for rayon's global pool of the reference.  Not to be called while queries are running.
    pub fn set_devices(&self, device_ids: &[i32]) -> Result<(), AwryError> {
        check(unsafe { sys::awry_set_devices(self.raw(), device_ids.as_ptr(), device_ids.len() as i32) })
    }

    /// Number of GPU replicas.
    pub fn num_devices(&self) -> usize {
        unsafe { sys::awry_num_devices(self.raw()) as usize }
    }

    /// Gets the alphabet of the index.
    pub fn alphabet(&self) -> SymbolAlphabet {
        SymbolAlphabet::from_id(unsafe { sys::awry_alphabet(self.raw()) })
    }

    /// Gets the suffix array compression ratio.
    pub fn suffix_array_compression_ratio(&self) -> u64 {
        unsafe { sys::awry_sa_ratio(self.raw()) }
    }

    /// Gets the length of the BWT (text length + 1).
    pub fn bwt_len(&self) -> u64 {
        unsafe { sys::awry_bwt_len(self.raw()) }
    }

    /// Gets the version number of the FM-index.
    pub fn version_number(&self) -> u64 {
        unsafe { sys::awry_version(self.raw()) }
    }

    /// Gets a reference to the prefix sums.
    pub fn prefix_sums(&self) -> &Vec<u64> {
        &self.prefix_sums
    }

    /// Headers of the indexed sequences, in file order (the reference keeps them in its crate-private SequenceIndex).
    pub fn sequence_headers(&self) -> Vec<String> {
        let n = unsafe { sys::awry_num_sequences(self.raw()) };
        (0..n)
            .map(|i| unsafe {
                let p: *const c_char = sys::awry_sequence_header(self.raw(), i);
                if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
            })
            .collect()
    }

    /// Start position of every indexed sequence in the concatenated text.
    pub fn sequence_starts(&self) -> Vec<usize> {
        let n = unsafe { sys::awry_num_sequences(self.raw()) };
        (0..n).map(|i| unsafe { sys::awry_sequence_start(self.raw(), i) as usize }).collect()
    }

    /// Gets the initial search range for the given character.
    pub fn initial_search_range(&self, s: Symbol) -> SearchRange {
        SearchRange::new(self, s)
    }

    /// Counts of a batch of queries, in input order (reference: src/fm_index.rs:455-460).  One library call for the
    /// whole batch: the queries are gathered into one byte buffer, packed 2 bits per letter on the library's worker pool
    /// and searched on the GPU replicas.
    pub fn parallel_count<'a>(&self, queries: impl ParallelIterator<Item = &'a str>) -> Vec<u64> {
        self.try_parallel_count(queries).unwrap_or_else(|e| panic!("{}", e))
    }

    /// [`FmIndex::parallel_count`] that returns the library's error instead of panicking.
    pub fn try_parallel_count<'a>(&self, queries: impl ParallelIterator<Item = &'a str>) -> Result<Vec<u64>, AwryError> {
        let csr = to_csr(queries);
        let n = csr.offsets.len() - 1;
        let mut counts = vec![0u64; n];
        check(unsafe { sys::awry_count_batch(self.raw(), csr.bytes.as_ptr(), csr.offsets.as_ptr(), n as u64, counts.as_mut_ptr()) })?;
        Ok(counts)
    }

    /// Counts of a CSR batch (query i = `bytes[offsets[i]..offsets[i + 1]]`) into a caller-owned slice: the zero-copy
    /// form of [`FmIndex::parallel_count`] for callers that already hold their queries contiguously.
    pub fn count_batch_into(&self, bytes: &[u8], offsets: &[u64], counts_out: &mut [u64]) -> Result<(), AwryError> {
        assert!(!offsets.is_empty() && counts_out.len() == offsets.len() - 1 && *offsets.last().unwrap() as usize <= bytes.len());
        check(unsafe { sys::awry_count_batch(self.raw(), bytes.as_ptr(), offsets.as_ptr(), counts_out.len() as u64, counts_out.as_mut_ptr()) })
    }

    /// Counts of k-mers the caller already holds packed: letter j (0 = leftmost) of k-mer i in bits `[2j, 2j + 2)` of
    /// `words[i]`, A0 C1 G2 T3, `len <= 32` (no counterpart in the reference; nucleotide indexes).
    pub fn count_packed_kmers(&self, words: &[u64], len: usize, counts_out: &mut [u64]) -> Result<(), AwryError> {
        assert_eq!(words.len(), counts_out.len());
        check(unsafe { sys::awry_count_packed_kmers(self.raw(), words.as_ptr(), words.len() as u64, len as i32, counts_out.as_mut_ptr()) })
    }

    /// Locations of a batch of queries (reference: src/fm_index.rs:479-487): outer order = input order, inner order =
    /// ascending BWT row, as `locate_string` returns them.
    pub fn parallel_locate<'a>(&self, queries: impl ParallelIterator<Item = &'a str>) -> Vec<Vec<LocalizedSequencePosition>> {
        self.try_parallel_locate(queries).unwrap_or_else(|e| panic!("{}", e))
    }

    /// [`FmIndex::parallel_locate`] that returns the library's error instead of panicking.
    pub fn try_parallel_locate<'a>(&self, queries: impl ParallelIterator<Item = &'a str>) -> Result<Vec<Vec<LocalizedSequencePosition>>, AwryError> {
        let csr = to_csr(queries);
        let n = csr.offsets.len() - 1;
        let mut hit_off: *mut u64 = std::ptr::null_mut();
        let mut hits: *mut sys::awry_pos_t = std::ptr::null_mut();
        check(unsafe {
            sys::awry_locate_batch(self.raw(), csr.bytes.as_ptr(), csr.offsets.as_ptr(), n as u64, &mut hit_off, &mut hits, std::ptr::null_mut())
        })?;
        let out = unsafe {
            let off = std::slice::from_raw_parts(hit_off, n + 1);
            let total = off[n] as usize;
            let flat: &[sys::awry_pos_t] = if total == 0 { &[] } else { std::slice::from_raw_parts(hits, total) };
            (0..n)
                .into_par_iter()
                .map(|i| {
                    flat[off[i] as usize..off[i + 1] as usize]
                        .iter()
                        .map(|p| LocalizedSequencePosition::new(p.seq_idx as usize, p.local_pos as usize))
                        .collect::<Vec<_>>()
                })
                .collect::<Vec<_>>()
        };
        unsafe {
            sys::awry_free_buffer(hit_off as *mut std::os::raw::c_void);
            sys::awry_free_buffer(hits as *mut std::os::raw::c_void);
        }
        Ok(out)
    }

    /// Locations of a batch as flat arrays: `(hit_offsets[n + 1], global text positions)`; the hits of query i are
    /// `positions[hit_offsets[i]..hit_offsets[i + 1]]`, `(SA sample + steps) % bwt_len` of src/fm_index.rs:534.
    /// Passes `hits_out = NULL`: 8 bytes per hit cross PCIe instead of 24.
    pub fn locate_batch_positions(&self, bytes: &[u8], offsets: &[u64]) -> Result<(Vec<u64>, Vec<u64>), AwryError> {
        assert!(!offsets.is_empty());
        // the C side reads bytes[offsets[i]..offsets[i + 1]]: a safe fn must not let bad offsets reach it
        assert!(offsets.windows(2).all(|w| w[0] <= w[1]), "query offsets must be non-decreasing");
        assert!(*offsets.last().unwrap() as usize <= bytes.len(), "query offsets run past the byte buffer");
        let n = offsets.len() - 1;
        let mut hit_off: *mut u64 = std::ptr::null_mut();
        let mut gpos: *mut u64 = std::ptr::null_mut();
        check(unsafe { sys::awry_locate_batch(self.raw(), bytes.as_ptr(), offsets.as_ptr(), n as u64, &mut hit_off, std::ptr::null_mut(), &mut gpos) })?;
        let (off, pos) = unsafe {
            let off = std::slice::from_raw_parts(hit_off, n + 1).to_vec();
            let total = off[n] as usize;
            let pos = if total == 0 { Vec::new() } else { std::slice::from_raw_parts(gpos, total).to_vec() };
            sys::awry_free_buffer(hit_off as *mut std::os::raw::c_void);
            sys::awry_free_buffer(gpos as *mut std::os::raw::c_void);
            (off, pos)
        };
        Ok((off, pos))
    }

    /// Finds the count for the given query (reference: src/fm_index.rs:499-501).  One kernel launch per call: batches
    /// are the way to use a GPU.
    pub fn count_string(&self, query: &str) -> u64 {
        self.try_count_string(query).unwrap_or_else(|e| panic!("{}", e))
    }

    pub fn try_count_string(&self, query: &str) -> Result<u64, AwryError> {
        let mut count = 0u64;
        check(unsafe { sys::awry_count(self.raw(), query.as_ptr(), query.len() as u64, &mut count) })?;
        Ok(count)
    }

    /// The search range of the query (the reference's crate-private `get_search_range_for_string`, src/fm_index.rs:402-438).
    pub fn search_range_for_string(&self, query: &str) -> Result<SearchRange, AwryError> {
        let mut r = sys::awry_range_t::default();
        check(unsafe { sys::awry_search_range(self.raw(), query.as_ptr(), query.len() as u64, &mut r) })?;
        Ok(SearchRange { start_ptr: r.start_ptr, end_ptr: r.end_ptr })
    }

    /// Finds the locations in the original text of all instances of the given query (reference: src/fm_index.rs:516-544).
    pub fn locate_string(&self, query: &str) -> Vec<LocalizedSequencePosition> {
        self.try_locate_string(query).unwrap_or_else(|e| panic!("{}", e))
    }

    pub fn try_locate_string(&self, query: &str) -> Result<Vec<LocalizedSequencePosition>, AwryError> {
        let mut hits: *mut sys::awry_pos_t = std::ptr::null_mut();
        let mut n = 0u64;
        check(unsafe { sys::awry_locate(self.raw(), query.as_ptr(), query.len() as u64, &mut hits, std::ptr::null_mut(), &mut n) })?;
        let out = unsafe {
            let flat: &[sys::awry_pos_t] = if n == 0 { &[] } else { std::slice::from_raw_parts(hits, n as usize) };
            let v = flat.iter().map(|p| LocalizedSequencePosition::new(p.seq_idx as usize, p.local_pos as usize)).collect();
            sys::awry_free_buffer(hits as *mut std::os::raw::c_void);
            v
        };
        Ok(out)
    }

    /// Perform a single SearchRange update using a given symbol (reference: src/fm_index.rs:559-582).
    pub fn update_range_with_symbol(&self, search_range: SearchRange, query_symbol: Symbol) -> SearchRange {
        let mut out = sys::awry_range_t::default();
        let rc = unsafe {
            sys::awry_update_range(
                self.raw(),
                sys::awry_range_t { start_ptr: search_range.start_ptr, end_ptr: search_range.end_ptr },
                query_symbol.ascii(),
                &mut out,
            )
        };
        check(rc).unwrap_or_else(|e| panic!("{}", e));
        SearchRange { start_ptr: out.start_ptr, end_ptr: out.end_ptr }
    }

    /// Finds the row of the symbol that precedes the given search pointer (LF-mapping; reference: src/fm_index.rs:585-593).
    pub fn backstep(&self, search_pointer: SearchPtr) -> SearchPtr {
        let mut out = 0u64;
        check(unsafe { sys::awry_backstep(self.raw(), search_pointer, &mut out) }).unwrap_or_else(|e| panic!("{}", e));
        out
    }

    /// Global text position -> (sequence, offset): largest sequence start <= position.
    pub fn get_seq_location(&self, global_position: usize) -> LocalizedSequencePosition {
        let mut p = sys::awry_pos_t::default();
        check(unsafe { sys::awry_get_seq_location(self.raw(), global_position as u64, &mut p) }).unwrap_or_else(|e| panic!("{}", e));
        LocalizedSequencePosition::new(p.seq_idx as usize, p.local_pos as usize)
    }
}

#[cfg(test)]
mod tests {
    //! The reference's own integration tests (src/fm_index.rs:612-743,1046-1088), restated over this crate's API: counts
    //! and sorted locations of every k-mer of a small random text equal brute force; save / load round-trips.
    //! Need an MI355X and libawry_hip.so at run time.
    use super::*;
    use rayon::prelude::*;
    use std::io::Write;

    fn random_text(n: usize, letters: &[u8], mut seed: u64) -> String {
        let mut s = String::with_capacity(n);
        for _ in 0..n {
            seed = seed.wrapping_mul(6364136223846793005).wrapping_add(1442695040888963407);
            s.push(letters[((seed >> 33) % letters.len() as u64) as usize] as char);
        }
        s
    }

    fn brute(text: &str, q: &str) -> Vec<usize> {
        (0..=text.len().saturating_sub(q.len())).filter(|&i| text[i..].starts_with(q)).collect()
    }

    fn check_index(path: &Path, alphabet: SymbolAlphabet, text: &str, k: usize) {
        let args = FmBuildArgs::new(path.into(), None, Some(8), None, alphabet, None, true);
        let ix = FmIndex::new(&args).expect("unable to build fm index");
        let kmers: Vec<&str> = (0..text.len() - k).map(|i| &text[i..i + k]).collect();
        let counts = ix.parallel_count(kmers.par_iter().copied());
        let locs = ix.parallel_locate(kmers.par_iter().copied());
        for (i, q) in kmers.iter().enumerate() {
            let want = brute(text, q);
            assert_eq!(counts[i] as usize, want.len());
            let mut got: Vec<usize> = locs[i].iter().map(|p| p.local_position()).collect();
            got.sort();
            assert_eq!(got, want);
            assert_eq!(ix.count_string(q), counts[i]);
        }
        let saved = path.with_extension("awry");
        ix.save(&saved).expect("save");
        let loaded = FmIndex::load(&saved).expect("load");
        assert!(loaded == ix);
        assert_eq!(loaded.parallel_count(kmers.par_iter().copied()), counts);
    }

    #[test]
    fn nucleotide_index_matches_brute_force() {
        let text = random_text(1847, b"ACGT", 0);
        let path = std::env::temp_dir().join("awry_shim_test_nucleotide.fasta");
        writeln!(std::fs::File::create(&path).unwrap(), ">seq0\n{}", text).unwrap();
        check_index(&path, SymbolAlphabet::Nucleotide, &text, 24);
    }

    #[test]
    fn amino_index_matches_brute_force() {
        let text = random_text(300, b"ACDEFGHIKLMNPQRSTVWY", 999);
        let path = std::env::temp_dir().join("awry_shim_test_amino.fasta");
        writeln!(std::fs::File::create(&path).unwrap(), ">seq0\n{}", text).unwrap();
        check_index(&path, SymbolAlphabet::Amino, &text, 8);
    }
}
