// Whole-file CRC32C the way the reference's backup engine computes it: the file is read
// buffer by buffer and every buffer extends one running, unmasked crc32c value
// (utilities/backupable/backupable_db.cc:1414-1452 CalculateChecksum, :1195-1283
// CopyOrCreateFile).  Here the buffers are pinned host memory and the arithmetic runs on the
// GPU through tcrc_extend; the results do not depend on the buffer size.
#pragma once
#include <stdint.h>

#include <string>

#include "../../include/tcrc.h"
#include "files.h"
#include "status.h"

namespace terarkdb_amd {

// The reference's default copy buffer is 5 MiB (kDefaultCopyFileBufferSize); a GPU call has a
// fixed cost of tens of microseconds plus the copy, so larger buffers are used here.
constexpr size_t kDefaultChecksumBufferSize = 64u << 20;

// *checksum_value = crc32c of the first size_limit bytes of src (0 = the whole file);
// *size, if given, receives the number of bytes covered.
Status CalculateChecksum(tcrc_ctx* ctx, SequentialFile* src, uint64_t size_limit, uint32_t* checksum_value,
                         uint64_t* size = nullptr, size_t buffer_size = kDefaultChecksumBufferSize);
Status CalculateChecksum(tcrc_ctx* ctx, const std::string& src, uint64_t size_limit, uint32_t* checksum_value,
                         uint64_t* size = nullptr, size_t buffer_size = kDefaultChecksumBufferSize);

// Copies src (or, when src is empty, writes `contents`) to dst, at most size_limit bytes (0 = all),
// and returns the number of bytes and their crc32c.  Exactly one of src / contents is non-empty.
Status CopyOrCreateFile(tcrc_ctx* ctx, const std::string& src, const std::string& dst, const std::string& contents,
                        bool sync, uint64_t* size, uint32_t* checksum_value, uint64_t size_limit = 0,
                        size_t buffer_size = kDefaultChecksumBufferSize);

}  // namespace terarkdb_amd
