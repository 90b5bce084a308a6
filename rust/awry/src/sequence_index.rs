//! Localised hit positions (reference: src/sequence_index.rs:31-78).

/// A position inside one sequence of the indexed collection: which sequence, and where in it.
#[derive(Clone, Debug, PartialEq, PartialOrd, Eq, Ord, Hash, Default)]
#[cfg_attr(feature = "serde", derive(serde::Serialize, serde::Deserialize))]
pub struct LocalizedSequencePosition {
    sequence_idx: usize,
    local_position: usize,
}

impl LocalizedSequencePosition {
    pub fn new(sequence_idx: usize, local_position: usize) -> Self {
        LocalizedSequencePosition { sequence_idx, local_position }
    }
    pub fn sequence_idx(&self) -> usize {
        self.sequence_idx
    }
    pub fn local_position(&self) -> usize {
        self.local_position
    }
}

#[cfg(test)]
mod tests {
    use super::LocalizedSequencePosition;

    #[test]
    fn accessors_and_order() {
        let a = LocalizedSequencePosition::new(0, 7);
        let b = LocalizedSequencePosition::new(1, 2);
        assert_eq!((a.sequence_idx(), a.local_position()), (0, 7));
        assert!(a < b);
        assert_eq!(LocalizedSequencePosition::default(), LocalizedSequencePosition::new(0, 0));
    }
}
