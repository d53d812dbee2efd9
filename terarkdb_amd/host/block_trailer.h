// SST block trailers in batches: the checksum part of BlockBasedTableBuilder::WriteRawBlock
// (table/block_based_table_builder.cc:580-646) and of BlockFetcher::CheckBlockChecksum
// (table/block_fetcher.cc:36-71), with the arithmetic done by the engine for many blocks
// at once.  On disk a block is  contents[n] | type u8 | checksum LE32  (table/format.h:190-196);
// for kCRC32c the checksum is Mask(CRC32C(contents | type)).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/tcrc.h"
#include "files.h"
#include "status.h"

namespace terarkdb_amd {

enum ChecksumType : char {  // include/rocksdb/table.h:48-53
  kNoChecksum = 0x0,
  kCRC32c = 0x1,
  kxxHash = 0x2,
  kxxHash64 = 0x3,
};

static const size_t kBlockTrailerSize = 5;  // table/format.h:190-191

struct BlockHandle {  // table/format.h:44-78 (offset and size of the block contents)
  uint64_t offset = 0;
  uint64_t size = 0;
};

// Appends blocks to a table file, each followed by its 5-byte trailer, like a run of
// WriteRawBlock calls.  contents[i] is the (possibly compressed) block and types[i] its
// CompressionType byte.  *file_offset is the file position before and after.  Only
// kCRC32c and kNoChecksum are handled; the xxHash kinds return NotSupported (they are not
// this engine's job).
Status WriteRawBlocks(tcrc_ctx* ctx, WritableFile* file, uint64_t* file_offset, const std::vector<Slice>& contents,
                      const std::vector<char>& types, ChecksumType checksum, std::vector<BlockHandle>* handles);

// Trailer values only (no file): trailers[i] = 5 bytes for block i.
Status ComputeBlockTrailers(tcrc_ctx* ctx, const std::vector<Slice>& contents, const std::vector<char>& types,
                            ChecksumType checksum, std::vector<std::string>* trailers);

// Verifies blocks of an image that holds them with their trailers: `image` is the bytes of
// the file starting at file offset image_file_offset; handles give contents offset/size in
// file coordinates.  statuses[i] is OK or the reference's Corruption text
//   "block checksum mismatch: expected <computed>, got <stored>  in <file> offset <o> size <n>".
// Returns the first non-OK status (or OK).
Status CheckBlockChecksums(tcrc_ctx* ctx, const Slice& image, uint64_t image_file_offset,
                           const std::vector<BlockHandle>& handles, ChecksumType footer_checksum,
                           const std::string& file_name, std::vector<Status>* statuses);

}  // namespace terarkdb_amd
