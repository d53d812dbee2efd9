// The device implementations of the writer's and reader's checksum seams.
#include <stdlib.h>
#include <string.h>

#include "wal_format.h"

namespace terarkdb_amd {
namespace log {

Status DeviceRecordStamper::Stamp(const char* base, size_t base_bytes, const tcrc_desc* spans, size_t n,
                                  uint32_t* out) {
  int rc = tcrc_host_batch_stamp(ctx_, base, base_bytes, spans, n, nullptr, TCRC_F_MASK, out);
  if (rc != TCRC_OK) return Status::IOError(std::string("checksum engine: ") + tcrc_last_error());
  return Status::OK();
}

char* DeviceRecordStamper::AllocStaging(size_t bytes) {
  char* p = static_cast<char*>(tcrc_host_alloc(ctx_, bytes));
  if (!p) abort();  // pinned allocation failure: nothing sensible to continue with
  return p;
}

void DeviceRecordStamper::FreeStaging(char* p) { tcrc_host_free(ctx_, p); }

Status DeviceRecordVerifier::ScanBlocks(const char* img, size_t n, uint64_t log_number, tcrc_wal_block* blocks) {
  int rc = tcrc_host_wal_verify(ctx_, img, n, log_number, blocks, nullptr, nullptr);
  if (rc != TCRC_OK) return Status::IOError(std::string("checksum engine: ") + tcrc_last_error());
  return Status::OK();
}

Status DeviceRecordVerifier::ScanBlocksPart(const char* img, size_t n, uint64_t log_number, uint64_t first_block,
                                            bool more_follows, tcrc_wal_block* blocks, uint64_t* counts) {
  int rc = tcrc_host_wal_verify_part(ctx_, img, n, log_number, first_block, more_follows ? 1 : 0, blocks, counts, nullptr);
  if (rc != TCRC_OK) return Status::IOError(std::string("checksum engine: ") + tcrc_last_error());
  return Status::OK();
}

char* DeviceRecordVerifier::AllocStaging(size_t bytes) { return static_cast<char*>(tcrc_host_alloc(ctx_, bytes)); }
void DeviceRecordVerifier::FreeStaging(char* p) { tcrc_host_free(ctx_, p); }

}  // namespace log
}  // namespace terarkdb_amd
