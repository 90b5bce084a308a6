//! The definition-level check the reference's own integration tests make (reference: src/fm_index.rs:612-664, driven by
//! test_nucleotide_index / test_amino_index :666-743), restated against this crate: for every k-mer that occurs in a
//! random single-record text, `count_string` equals the number of occurrences and the sorted `locate_string` result
//! equals the occurrence positions.  The reference reads the text back from the `.sufr` file libsufr wrote; here the
//! text is the one this test generated (no libsufr).  Needs libawry_hip.so and an MI355X:
//!
//!     AWRY_HIP_LIB_DIR=$PWD/../../awry_amd/lib cargo test --release --test compare_to_brute_force
//!
//! NOT COMPILED in the authoring image (no cargo / rustc there) -- see rust/README.md.
use std::collections::BTreeMap;
use std::io::Write;
use std::path::{Path, PathBuf};

use awry::alphabet::SymbolAlphabet;
use awry::fm_index::{FmBuildArgs, FmIndex};

/// xorshift64*: the test needs reproducible letters, not quality
struct Rng(u64);
impl Rng {
    fn next(&mut self) -> u64 {
        self.0 ^= self.0 >> 12;
        self.0 ^= self.0 << 25;
        self.0 ^= self.0 >> 27;
        self.0.wrapping_mul(0x2545_F491_4F6C_DD1D)
    }
    fn letters(&mut self, alphabet: &[u8], n: usize) -> Vec<u8> {
        (0..n).map(|_| alphabet[(self.next() % alphabet.len() as u64) as usize]).collect()
    }
}

fn write_fasta(path: &Path, header: &str, seq: &[u8], width: usize) {
    let mut f = std::fs::File::create(path).expect("cannot create the FASTA file");
    writeln!(f, ">{header}").unwrap();
    for line in seq.chunks(width) {
        f.write_all(line).unwrap();
        f.write_all(b"\n").unwrap();
    }
}

fn tmp(name: &str) -> PathBuf {
    std::env::temp_dir().join(format!("awry_shim_{}_{}", std::process::id(), name))
}

/// every k-mer of `text` -> its occurrence positions (ascending)
fn occurrences(text: &[u8], k: usize) -> BTreeMap<&[u8], Vec<usize>> {
    let mut m: BTreeMap<&[u8], Vec<usize>> = BTreeMap::new();
    for p in 0..=text.len().saturating_sub(k) {
        m.entry(&text[p..p + k]).or_default().push(p);
    }
    m
}

fn check_against_definition(index: &FmIndex, text: &[u8], k: usize) {
    for (kmer, want) in occurrences(text, k) {
        let q = std::str::from_utf8(kmer).unwrap();
        assert_eq!(index.count_string(q) as usize, want.len(), "count of {q}");
        let mut got: Vec<usize> = index.locate_string(q).iter().map(|p| p.local_position() as usize).collect();
        got.sort_unstable();
        assert_eq!(got, want, "positions of {q}");
    }
    // the batch entry points agree with the scalar ones, in input order
    let qs: Vec<String> = occurrences(text, k).keys().take(200).map(|b| String::from_utf8(b.to_vec()).unwrap()).collect();
    use rayon::prelude::*;
    let counts = index.parallel_count(qs.par_iter().map(|s| s.as_str()));
    for (q, c) in qs.iter().zip(counts) {
        assert_eq!(c, index.count_string(q));
    }
}

fn build(fasta: &Path, alphabet: SymbolAlphabet) -> FmIndex {
    FmIndex::new(&FmBuildArgs {
        input_file_src: fasta.to_path_buf(),
        suffix_array_output_src: Some(tmp("unused.sufr")),
        suffix_array_compression_ratio: None,
        lookup_table_kmer_len: None,
        alphabet,
        max_query_len: None,
        remove_intermediate_suffix_array_file: false,
    })
    .expect("FmIndex::new failed")
}

#[test]
fn nucleotide_index_matches_the_definition() {
    // the reference's sizes: 1 847 letters, 24-mers, 80 letters per FASTA line
    let seq = Rng(0x5EED_0001).letters(b"ACGT", 1847);
    let fasta = tmp("nt.fasta");
    write_fasta(&fasta, "random_nucleotides", &seq, 80);
    let index = build(&fasta, SymbolAlphabet::Nucleotide);
    check_against_definition(&index, &seq, 24);
    // save -> load -> same answers (reference: src/fm_index_file.rs:42,132)
    let file = tmp("nt.awry");
    index.save(&file).expect("save failed");
    let again = FmIndex::load(&file).expect("load failed");
    check_against_definition(&again, &seq, 24);
    let _ = std::fs::remove_file(fasta);
    let _ = std::fs::remove_file(file);
}

#[test]
fn amino_index_matches_the_definition() {
    // 300 residues, 8-mers
    let seq = Rng(0x5EED_0002).letters(b"ACDEFGHIKLMNPQRSTVWY", 300);
    let fasta = tmp("aa.fasta");
    write_fasta(&fasta, "random_residues", &seq, 80);
    let index = build(&fasta, SymbolAlphabet::Amino);
    check_against_definition(&index, &seq, 8);
    let _ = std::fs::remove_file(fasta);
}
