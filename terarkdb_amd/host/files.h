// File interfaces the log writer/reader and the block trailer code sit on: the three
// abstract classes of include/rocksdb/env.h that the hot path touches, POSIX
// implementations with the system calls of env/io_posix.cc (fread for sequential reads
// :171-191, pread for random reads :376-387, write(2) for appends :778-800), and the
// in-memory fakes the reference's log test drives its reader and writer with
// (db/log_test.cc:49-114, util/testutil.h:201-252,370-400).
#pragma once
#include <stdint.h>

#include <memory>
#include <string>

#include "status.h"

namespace terarkdb_amd {

class SequentialFile {
 public:
  virtual ~SequentialFile() {}
  // Reads up to n bytes; *result points into scratch (or elsewhere) and is shorter than n only at end of file.
  virtual Status Read(size_t n, Slice* result, char* scratch) = 0;
  virtual Status Skip(uint64_t n) = 0;
};

class RandomAccessFile {
 public:
  virtual ~RandomAccessFile() {}
  virtual Status Read(uint64_t offset, size_t n, Slice* result, char* scratch) const = 0;
  virtual uint64_t Size() const = 0;
};

class WritableFile {
 public:
  virtual ~WritableFile() {}
  virtual Status Append(const Slice& data) = 0;
  virtual Status Flush() = 0;
  virtual Status Sync() { return Status::OK(); }
  virtual Status Close() { return Status::OK(); }
};

Status NewPosixSequentialFile(const std::string& path, std::unique_ptr<SequentialFile>* out);
Status NewPosixRandomAccessFile(const std::string& path, std::unique_ptr<RandomAccessFile>* out);
Status NewPosixWritableFile(const std::string& path, std::unique_ptr<WritableFile>* out);

// Fills buf with file bytes [offset, offset + n) using up to `threads` concurrent positional reads
// (page-cache copies are memcpy-bound per thread).  Fails if the file ends before offset + n.
Status ParallelRead(const RandomAccessFile* file, uint64_t offset, size_t n, char* buf, int threads);

// ---- in-memory fakes ---------------------------------------------------------------

// The bytes of one in-memory file, shared by a sink (writer side) and any number of sources.
struct MemFile {
  std::string contents;
};

// Appends to a MemFile; with overwrite=true it starts at offset 0 and overwrites what is
// there first, like a recycled log file.
class MemSink : public WritableFile {
 public:
  explicit MemSink(std::shared_ptr<MemFile> f, bool overwrite = false)
      : file_(std::move(f)), pos_(overwrite ? 0 : file_->contents.size()) {}
  Status Append(const Slice& data) override;
  Status Flush() override { return Status::OK(); }
  void Drop(size_t bytes);  // cut bytes off the end of the file
 private:
  std::shared_ptr<MemFile> file_;
  size_t pos_;
};

// Reads a MemFile from the start; sees bytes appended later.  force_error / force_eof make
// the Read that reaches the given position (counted in requested bytes from now on) fail
// with Corruption("read error") / return short, once.
class MemSource : public SequentialFile {
 public:
  explicit MemSource(std::shared_ptr<MemFile> f) : file_(std::move(f)) {}
  Status Read(size_t n, Slice* result, char* scratch) override;
  Status Skip(uint64_t n) override;
  void ForceError(size_t position) {
    force_error_ = true;
    force_error_position_ = position;
  }
  void ForceEOF(size_t position) {
    force_eof_ = true;
    force_eof_position_ = position;
  }
  void ClearPartial() { returned_partial_ = false; }
  bool read_after_partial() const { return read_after_partial_; }

 private:
  std::shared_ptr<MemFile> file_;
  size_t pos_ = 0;
  bool force_error_ = false, force_eof_ = false, returned_partial_ = false, read_after_partial_ = false;
  size_t force_error_position_ = 0, force_eof_position_ = 0;
};

class MemRandomAccessFile : public RandomAccessFile {
 public:
  explicit MemRandomAccessFile(std::shared_ptr<MemFile> f) : file_(std::move(f)) {}
  Status Read(uint64_t offset, size_t n, Slice* result, char* scratch) const override;
  uint64_t Size() const override { return file_->contents.size(); }

 private:
  std::shared_ptr<MemFile> file_;
};

}  // namespace terarkdb_amd
