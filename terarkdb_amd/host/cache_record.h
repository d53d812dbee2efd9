// Records of the persistent block cache's files in batches: the checksum work of CacheRecord
// (utilities/persistent_cache/block_cache_tier_file.cc:55-192).  On disk a record is
//   magic u32 | crc u32 | key size u32 | value size u32 | key | value        (:60-75, host byte order)
// with crc = crc32c::Extend over the header with its crc field zeroed, then the key, then the value
// (ComputeCRC, :108-118) - a plain CRC, not a masked one.
#pragma once
#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/tcrc.h"
#include "status.h"

namespace terarkdb_amd {

struct CacheRecordHeader {  // block_cache_tier_file.cc:65-75
  uint32_t magic;
  uint32_t crc;
  uint32_t key_size;
  uint32_t val_size;
};
static_assert(sizeof(CacheRecordHeader) == 16, "header layout");
const uint32_t kCacheRecordMagic = 0xfefa;  // :96

// Appends the records (keys[i], vals[i]) to *out the way n CacheRecord(key, val).Serialize calls lay them
// out (:77-86,120-129); the checksums of all of them come from one engine submission.
// offsets, if given, receives the position of every record in *out.
Status SerializeCacheRecords(tcrc_ctx* ctx, const std::vector<Slice>& keys, const std::vector<Slice>& vals,
                             std::string* out, std::vector<uint64_t>* offsets = nullptr);

struct CacheRecordView {
  bool ok = false;  // what CacheRecord::Deserialize returns (:162-192): size, magic and checksum all fit
  Slice key, val;   // set when the sizes fit, pointing into the image
};

// Checks the records at (offset, size) - the cache's logical block addresses - inside image, all in one
// engine submission.
Status DeserializeCacheRecords(tcrc_ctx* ctx, const Slice& image, const std::vector<std::pair<uint64_t, uint32_t>>& lbas,
                               std::vector<CacheRecordView>* records);

}  // namespace terarkdb_amd
