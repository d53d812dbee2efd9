#include "file_checksum.h"

#include <limits>
#include <memory>

namespace terarkdb_amd {

namespace {

// Pinned buffer owned for the duration of one call.
struct PinnedBuffer {
  tcrc_ctx* ctx;
  char* p;
  PinnedBuffer(tcrc_ctx* c, size_t n) : ctx(c), p(static_cast<char*>(tcrc_host_alloc(c, n))) {}
  ~PinnedBuffer() { tcrc_host_free(ctx, p); }
};

Status EngineError() { return Status::IOError(std::string("checksum engine: ") + tcrc_last_error()); }

}  // namespace

Status CalculateChecksum(tcrc_ctx* ctx, SequentialFile* src, uint64_t size_limit, uint32_t* checksum_value,
                         uint64_t* size, size_t buffer_size) {
  *checksum_value = 0;
  if (size) *size = 0;
  if (size_limit == 0) size_limit = std::numeric_limits<uint64_t>::max();
  if (buffer_size == 0) return Status::InvalidArgument("buffer_size must be positive");
  PinnedBuffer buf(ctx, buffer_size);
  if (!buf.p) return Status::IOError("checksum engine: cannot allocate the pinned read buffer");
  Slice data;
  do {
    const size_t to_read = buffer_size < size_limit ? buffer_size : static_cast<size_t>(size_limit);
    Status s = src->Read(to_read, &data, buf.p);
    if (!s.ok()) return s;
    size_limit -= data.size();
    if (size) *size += data.size();
    if (tcrc_extend(ctx, *checksum_value, data.data(), data.size(), checksum_value) != TCRC_OK) return EngineError();
  } while (data.size() > 0 && size_limit > 0);
  return Status::OK();
}

Status CalculateChecksum(tcrc_ctx* ctx, const std::string& src, uint64_t size_limit, uint32_t* checksum_value,
                         uint64_t* size, size_t buffer_size) {
  *checksum_value = 0;
  std::unique_ptr<SequentialFile> file;
  Status s = NewPosixSequentialFile(src, &file);
  if (!s.ok()) return s;
  return CalculateChecksum(ctx, file.get(), size_limit, checksum_value, size, buffer_size);
}

Status CopyOrCreateFile(tcrc_ctx* ctx, const std::string& src, const std::string& dst, const std::string& contents,
                        bool sync, uint64_t* size, uint32_t* checksum_value, uint64_t size_limit, size_t buffer_size) {
  if (src.empty() == contents.empty()) return Status::InvalidArgument("exactly one of src and contents must be given");
  if (size) *size = 0;
  if (checksum_value) *checksum_value = 0;
  if (size_limit == 0) size_limit = std::numeric_limits<uint64_t>::max();
  std::unique_ptr<WritableFile> dst_file;
  std::unique_ptr<SequentialFile> src_file;
  Status s = NewPosixWritableFile(dst, &dst_file);
  if (s.ok() && !src.empty()) s = NewPosixSequentialFile(src, &src_file);
  if (!s.ok()) return s;
  std::unique_ptr<PinnedBuffer> buf;
  if (!src.empty()) {
    buf.reset(new PinnedBuffer(ctx, buffer_size));
    if (!buf->p) return Status::IOError("checksum engine: cannot allocate the pinned read buffer");
  }
  Slice data;
  do {
    if (!src.empty()) {
      const size_t to_read = buffer_size < size_limit ? buffer_size : static_cast<size_t>(size_limit);
      s = src_file->Read(to_read, &data, buf->p);
    } else {
      data = Slice(contents);
    }
    size_limit -= data.size();
    if (!s.ok()) return s;
    if (size) *size += data.size();
    if (checksum_value &&
        tcrc_extend(ctx, *checksum_value, data.data(), data.size(), checksum_value) != TCRC_OK)
      return EngineError();
    s = dst_file->Append(data);
  } while (s.ok() && contents.empty() && data.size() > 0 && size_limit > 0);
  if (s.ok()) s = dst_file->Flush();
  if (s.ok() && sync) s = dst_file->Sync();
  if (s.ok()) s = dst_file->Close();
  return s;
}

}  // namespace terarkdb_amd
