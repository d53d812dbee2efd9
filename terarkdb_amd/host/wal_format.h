// Log file format constants (db/log_format.h:19-44) and the two seams through which the
// writer and reader reach the checksum engine.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/tcrc.h"
#include "status.h"

namespace terarkdb_amd {
namespace log {

enum RecordType {
  kZeroType = 0,  // preallocated space
  kFullType = 1,
  kFirstType = 2,
  kMiddleType = 3,
  kLastType = 4,
  kRecyclableFullType = 5,
  kRecyclableFirstType = 6,
  kRecyclableMiddleType = 7,
  kRecyclableLastType = 8,
};
static const int kMaxRecordType = kRecyclableLastType;
static const unsigned int kBlockSize = 32768;
static const int kHeaderSize = 4 + 2 + 1;             // checksum, length, type
static const int kRecyclableHeaderSize = 4 + 2 + 1 + 4;  // + log number

// Computes the masked CRC32C of many byte spans of one staging buffer in one submission.
// The shipped implementation is DeviceRecordStamper (the GPU engine).
class RecordStamper {
 public:
  virtual ~RecordStamper() {}
  // out[i] = Mask(Value(base + spans[i].offset, spans[i].len))
  virtual Status Stamp(const char* base, size_t base_bytes, const tcrc_desc* spans, size_t n, uint32_t* out) = 0;
  // Memory the writer frames records into (pinned for the device implementation).
  virtual char* AllocStaging(size_t bytes);
  virtual void FreeStaging(char* p);
};

// Checks the checksums of the physical records of a log image.  The shipped implementation is
// DeviceRecordVerifier (the GPU engine).
class RecordVerifier {
 public:
  virtual ~RecordVerifier() {}
  // img[0..n) starts at a record boundary that is taken as the start of a 32 KiB block;
  // blocks[b] describes bytes [32768 b, 32768 (b+1)).
  virtual Status ScanBlocks(const char* img, size_t n, uint64_t log_number, tcrc_wal_block* blocks) = 0;
};

class DeviceRecordStamper : public RecordStamper {
 public:
  explicit DeviceRecordStamper(tcrc_ctx* ctx) : ctx_(ctx) {}
  Status Stamp(const char* base, size_t base_bytes, const tcrc_desc* spans, size_t n, uint32_t* out) override;
  char* AllocStaging(size_t bytes) override;
  void FreeStaging(char* p) override;

 private:
  tcrc_ctx* ctx_;
};

class DeviceRecordVerifier : public RecordVerifier {
 public:
  explicit DeviceRecordVerifier(tcrc_ctx* ctx) : ctx_(ctx) {}
  Status ScanBlocks(const char* img, size_t n, uint64_t log_number, tcrc_wal_block* blocks) override;

 private:
  tcrc_ctx* ctx_;
};

// Verdicts for a whole log file computed in one device pass, handed to a Reader so that it
// does not have to ask block by block (recovery reads a complete file).
struct VerifiedLogImage {
  uint64_t size = 0;  // bytes of the file that were verified
  std::vector<tcrc_wal_block> blocks;
  uint64_t records = 0;
};
Status VerifyLogImage(RecordVerifier* verifier, const char* img, size_t n, uint64_t log_number, VerifiedLogImage* out);

}  // namespace log
}  // namespace terarkdb_amd
