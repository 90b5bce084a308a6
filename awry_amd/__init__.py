"""awry_amd -- MI355X-native FM-index search engine behind AWRY's `FmIndex` API.

The product is the C-ABI library `awry_amd/lib/libawry_hip.so` (include/awry_hip.h); this package is the
thin Python host-side binding used by tests and the bench harness.  There is no CPU search path: every
query runs HIP kernels, and importing without the built extension raises.
"""
from .fm_index import (  # noqa: F401
    AMINO,
    NUCLEOTIDE,
    AwryError,
    FmBuildArgs,
    FmIndex,
    LocalizedSequencePosition,
    SearchRange,
    SymbolAlphabet,
)
from ._lib import lib_path, load_library  # noqa: F401
