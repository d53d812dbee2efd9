"""terarkdb_amd: MI355X-native batched masked-CRC32C engine for TerarkDB's
record-checksum path (WAL physical records, SST block trailers).

The product is the C-ABI shared library `lib/libtcrc.so` (HIP kernels for gfx950
plus C++ writer/reader classes shaped like the reference's).  This package is the
thin ctypes view used by tests and the benchmark.  There is no CPU fallback:
importing `terarkdb_amd.capi` fails loudly when the library has not been built."""

__all__ = ["capi"]
