//! Search ranges (reference: src/search.rs:7,25-81).
use crate::{alphabet::Symbol, fm_index::FmIndex};

/// A position (row) in the BWT.
pub(crate) type SearchPtr = u64;

/// The closed range of BWT rows that corresponds to a query; valid while `start_ptr <= end_ptr`.
#[derive(Clone, Debug, PartialEq, PartialOrd, Eq, Ord, Hash, Default)]
#[cfg_attr(feature = "serde", derive(serde::Serialize, serde::Deserialize))]
pub struct SearchRange {
    pub start_ptr: SearchPtr,
    pub end_ptr: SearchPtr,
}

impl SearchRange {
    /// The range of all suffixes that start with `symbol`: `[C[s], C[s + 1] - 1]`.
    pub fn new(fm_index: &FmIndex, symbol: Symbol) -> Self {
        SearchRange {
            start_ptr: fm_index.prefix_sums()[symbol.index() as usize] as SearchPtr,
            end_ptr: fm_index.prefix_sums()[(symbol.index() + 1) as usize] - 1 as SearchPtr,
        }
    }
    /// An invalid range (no elements).
    pub fn zero() -> Self {
        SearchRange { start_ptr: 1, end_ptr: 0 }
    }
    /// True if the range represents no element.
    #[inline]
    pub fn is_empty(&self) -> bool {
        self.start_ptr > self.end_ptr
    }
    /// Number of elements the range represents.
    #[inline]
    pub fn len(&self) -> SearchPtr {
        match self.is_empty() {
            true => 0,
            false => self.end_ptr - self.start_ptr + 1,
        }
    }
    /// Iterator over the BWT rows of the range.
    #[inline]
    pub fn range_iter(&self) -> core::ops::Range<SearchPtr> {
        match self.is_empty() {
            true => 0..0,
            false => self.start_ptr..(self.end_ptr + 1),
        }
    }
}

#[cfg(test)]
mod tests {
    use super::SearchRange;

    #[test]
    fn zero_range_is_empty() {
        assert_eq!(SearchRange::zero().len(), 0);
        assert!(SearchRange::zero().is_empty());
        assert_eq!(SearchRange { start_ptr: 999, end_ptr: 0 }.len(), 0);
        assert_eq!(SearchRange { start_ptr: 999, end_ptr: 0 }.range_iter().count(), 0);
        assert_eq!(SearchRange { start_ptr: 3, end_ptr: 7 }.len(), 5);
        assert_eq!(SearchRange { start_ptr: 3, end_ptr: 7 }.range_iter().collect::<Vec<_>>(), vec![3, 4, 5, 6, 7]);
    }
}
