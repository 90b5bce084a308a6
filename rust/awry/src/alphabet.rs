//! Symbols and alphabets (reference: src/alphabet.rs:28-31,48-61,87-97,109-157,169-413).
use std::fmt::Display;

use awry_hip_sys as sys;

/// Alphabet of the indexed text.
#[derive(Clone, Debug, PartialEq, PartialOrd, Eq, Ord, Hash, Copy)]
#[cfg_attr(feature = "serde", derive(serde::Serialize, serde::Deserialize))]
pub enum SymbolAlphabet {
    Nucleotide,
    Amino,
}

impl Display for SymbolAlphabet {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "{}", match self {
            SymbolAlphabet::Nucleotide => "nucleotide",
            SymbolAlphabet::Amino => "amino",
        })
    }
}

impl SymbolAlphabet {
    pub(crate) fn alphabet_id(&self) -> u8 {
        match self {
            SymbolAlphabet::Nucleotide => 0,
            SymbolAlphabet::Amino => 1,
        }
    }
    pub(crate) fn from_id(id: i32) -> Self {
        match id {
            0 => SymbolAlphabet::Nucleotide,
            1 => SymbolAlphabet::Amino,
            _ => panic!("invalid alphabet id given"),
        }
    }
    /// How many different symbols can occur in this alphabet (`$` and the ambiguity symbol included).
    pub(crate) fn cardinality(&self) -> u8 {
        match self {
            SymbolAlphabet::Nucleotide => 6,
            SymbolAlphabet::Amino => 22,
        }
    }
}

#[derive(Debug, PartialEq, Eq, Clone, Copy)]
enum SymbolEncoding {
    Ascii(char),
    Index(u8),
}

/// A symbol of a given alphabet, given as an ASCII letter or as an index into the alphabet
/// (`$ A C G N T` / `$ A C D E F G H I K L M N P Q R S T V W X Y`).
#[derive(Debug, PartialEq, Eq, Clone, Copy)]
pub struct Symbol {
    alphabet: SymbolAlphabet,
    encoding: SymbolEncoding,
}

const NT_LETTERS: &[u8; 6] = b"$ACGNT";
const AA_LETTERS: &[u8; 22] = b"$ACDEFGHIKLMNPQRSTVWXY";

impl Symbol {
    /// Creates a Symbol from an ASCII letter (case-insensitive; unknown letters search as N / X, U as T).
    pub fn new_ascii(alphabet: SymbolAlphabet, ascii: char) -> Symbol {
        Symbol { alphabet, encoding: SymbolEncoding::Ascii(ascii.to_ascii_uppercase()) }
    }
    /// Creates a Symbol from its index into the alphabet.
    pub fn new_index(alphabet: SymbolAlphabet, index: u8) -> Symbol {
        debug_assert!(index < alphabet.cardinality());
        Symbol { alphabet, encoding: SymbolEncoding::Index(index) }
    }
    /// Index of the symbol in its alphabet: the library's `Symbol::new_ascii(..).index()` (awry_symbol_index).
    pub(crate) fn index(&self) -> u8 {
        match self.encoding {
            SymbolEncoding::Index(i) => i,
            SymbolEncoding::Ascii(c) => {
                let b = if c.is_ascii() { c as u8 } else { b'?' };
                unsafe { sys::awry_symbol_index(self.alphabet.alphabet_id() as i32, b) }
            }
        }
    }
    /// The ASCII letter of the symbol (what the C ABI's scalar entry points take).
    pub(crate) fn ascii(&self) -> u8 {
        let i = self.index() as usize;
        match self.alphabet {
            SymbolAlphabet::Nucleotide => NT_LETTERS[i.min(5)],
            SymbolAlphabet::Amino => AA_LETTERS[i.min(21)],
        }
    }
    pub(crate) fn is_sentinel(&self) -> bool {
        self.index() == 0
    }
}

impl Display for Symbol {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "{} code {}", self.alphabet, self.ascii() as char)
    }
}
