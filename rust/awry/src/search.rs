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
    /// The range of all suffixes that start with `symbol`: rows `C[s] ..= C[s + 1] - 1`, where `C` are the index's
    /// prefix sums (one sentinel row precedes every letter, so `C[s + 1] >= 1` and the subtraction cannot wrap).
    pub fn new(fm_index: &FmIndex, symbol: Symbol) -> Self {
        let c = fm_index.prefix_sums();
        let s = usize::from(symbol.index());
        let (first, next): (SearchPtr, SearchPtr) = (c[s], c[s + 1]);
        Self { start_ptr: first, end_ptr: next.wrapping_sub(1) }
    }
    /// The canonical empty range `{1, 0}`.
    pub fn zero() -> Self {
        Self { start_ptr: 1, end_ptr: 0 }
    }
    /// A range is empty exactly when its end lies before its start.
    #[inline]
    pub fn is_empty(&self) -> bool {
        self.end_ptr < self.start_ptr
    }
    /// Rows in the range (`0` for an empty one; never wraps).
    #[inline]
    pub fn len(&self) -> SearchPtr {
        self.end_ptr.checked_sub(self.start_ptr).map_or(0, |d| d + 1)
    }
    /// The rows of the range in ascending order; nothing for an empty range.
    #[inline]
    pub fn range_iter(&self) -> core::ops::Range<SearchPtr> {
        let n = self.len();
        if n == 0 { 0..0 } else { self.start_ptr..self.start_ptr + n }
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
