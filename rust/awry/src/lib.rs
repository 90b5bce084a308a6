//! `awry` with its query path on MI355X GPUs.
//!
//! Same public surface as AWRY 0.3.1 (`fm_index::{FmIndex, FmBuildArgs}`, `alphabet::{Symbol, SymbolAlphabet}`,
//! `search::SearchRange`, `sequence_index::LocalizedSequencePosition`); every search runs as HIP kernels inside
//! `libawry_hip.so` (C ABI: include/awry_hip.h).  There is no CPU search path: an index answers queries on the GPUs
//! chosen with [`fm_index::FmIndex::set_devices`] (GPU 0 by default).
//!
//! Written against the reference's sources; this repository's image has no Rust toolchain, so the crate ships as
//! source only (see README.md).
pub mod alphabet;
pub mod fm_index;
pub mod search;
pub mod sequence_index;
