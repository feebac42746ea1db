"""rust_mdbg_amd — MI355X-native minimizer sketching and k-min-mer counting (drop-in for rust-mdbg's hot path).

The compute lives in libmdbg_hip.so (hand-written HIP for gfx950, C ABI in include/mdbg_hip.h); this package is
the thin host-side mirror of the reference's interface for that path.  There is no CPU fallback: importing
`rust_mdbg_amd.api` without the built library raises.
"""
from .api import Mdbg, MdbgError, Params, lib_path, load_library  # noqa: F401
