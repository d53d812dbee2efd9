#include "files.h"

#include <thread>
#include <vector>

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

namespace terarkdb_amd {

namespace {

Status Errno(const std::string& what, const std::string& path) {
  return Status::IOError(what + " " + path + ": " + strerror(errno));
}

class PosixSequentialFile : public SequentialFile {
 public:
  PosixSequentialFile(std::string path, FILE* f) : path_(std::move(path)), f_(f) {}
  ~PosixSequentialFile() override { fclose(f_); }
  Status Read(size_t n, Slice* result, char* scratch) override {
    size_t r;
    do {
      r = fread_unlocked(scratch, 1, n, f_);
    } while (r == 0 && ferror(f_) && errno == EINTR);
    *result = Slice(scratch, r);
    if (r < n) {
      if (feof(f_)) {
        clearerr(f_);  // leave the stream usable: the file may grow (tailing readers)
      } else {
        return Errno("read", path_);
      }
    }
    return Status::OK();
  }
  Status Skip(uint64_t n) override {
    if (fseek(f_, static_cast<long>(n), SEEK_CUR)) return Errno("seek", path_);
    return Status::OK();
  }

 private:
  std::string path_;
  FILE* f_;
};

class PosixRandomAccessFile : public RandomAccessFile {
 public:
  PosixRandomAccessFile(std::string path, int fd, uint64_t size) : path_(std::move(path)), fd_(fd), size_(size) {}
  ~PosixRandomAccessFile() override { close(fd_); }
  Status Read(uint64_t offset, size_t n, Slice* result, char* scratch) const override {
    size_t left = n;
    char* p = scratch;
    while (left > 0) {
      ssize_t r = pread(fd_, p, left, static_cast<off_t>(offset));
      if (r < 0) {
        if (errno == EINTR) continue;
        *result = Slice(scratch, 0);
        return Errno("pread", path_);
      }
      if (r == 0) break;
      p += r;
      offset += static_cast<uint64_t>(r);
      left -= static_cast<size_t>(r);
    }
    *result = Slice(scratch, n - left);
    return Status::OK();
  }
  uint64_t Size() const override { return size_; }

 private:
  std::string path_;
  int fd_;
  uint64_t size_;
};

class PosixWritableFile : public WritableFile {
 public:
  PosixWritableFile(std::string path, int fd) : path_(std::move(path)), fd_(fd) {}
  ~PosixWritableFile() override {
    if (fd_ >= 0) close(fd_);
  }
  Status Append(const Slice& data) override {
    const char* p = data.data();
    size_t left = data.size();
    while (left > 0) {
      ssize_t w = write(fd_, p, left);
      if (w < 0) {
        if (errno == EINTR) continue;
        return Errno("write", path_);
      }
      p += w;
      left -= static_cast<size_t>(w);
    }
    return Status::OK();
  }
  Status Flush() override { return Status::OK(); }
  Status Sync() override {
    if (fdatasync(fd_) < 0) return Errno("fdatasync", path_);
    return Status::OK();
  }
  Status Close() override {
    int fd = fd_;
    fd_ = -1;
    if (fd >= 0 && close(fd) < 0) return Errno("close", path_);
    return Status::OK();
  }

 private:
  std::string path_;
  int fd_;
};

}  // namespace

Status NewPosixSequentialFile(const std::string& path, std::unique_ptr<SequentialFile>* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return Errno("open", path);
  out->reset(new PosixSequentialFile(path, f));
  return Status::OK();
}

Status NewPosixRandomAccessFile(const std::string& path, std::unique_ptr<RandomAccessFile>* out) {
  int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
  if (fd < 0) return Errno("open", path);
  struct stat st;
  if (fstat(fd, &st) < 0) {
    close(fd);
    return Errno("fstat", path);
  }
  out->reset(new PosixRandomAccessFile(path, fd, static_cast<uint64_t>(st.st_size)));
  return Status::OK();
}

Status NewPosixWritableFile(const std::string& path, std::unique_ptr<WritableFile>* out) {
  int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY | O_CLOEXEC, 0644);
  if (fd < 0) return Errno("open", path);
  out->reset(new PosixWritableFile(path, fd));
  return Status::OK();
}

Status MemSink::Append(const Slice& data) {
  std::string& c = file_->contents;
  if (pos_ + data.size() > c.size()) c.resize(pos_ + data.size());
  if (data.size()) memcpy(&c[pos_], data.data(), data.size());
  pos_ += data.size();
  return Status::OK();
}

void MemSink::Drop(size_t bytes) {
  std::string& c = file_->contents;
  c.resize(c.size() - bytes);
  if (pos_ > c.size()) pos_ = c.size();
}

Status MemSource::Read(size_t n, Slice* result, char* scratch) {
  if (returned_partial_) read_after_partial_ = true;
  const std::string& c = file_->contents;
  size_t avail = c.size() > pos_ ? c.size() - pos_ : 0;
  if (force_error_) {
    if (force_error_position_ >= n) {
      force_error_position_ -= n;
    } else {
      size_t k = force_error_position_ < avail ? force_error_position_ : avail;
      if (k) memcpy(scratch, c.data() + pos_, k);
      *result = Slice(scratch, k);
      pos_ += k;
      force_error_ = false;
      returned_partial_ = true;
      return Status::Corruption("read error");
    }
  }
  if (avail < n) {
    n = avail;
    returned_partial_ = true;
  }
  if (force_eof_) {
    if (force_eof_position_ >= n) {
      force_eof_position_ -= n;
    } else {
      force_eof_ = false;
      n = force_eof_position_;
      returned_partial_ = true;
    }
  }
  if (n) memcpy(scratch, c.data() + pos_, n);
  *result = Slice(scratch, n);
  pos_ += n;
  return Status::OK();
}

Status MemSource::Skip(uint64_t n) {
  const std::string& c = file_->contents;
  size_t avail = c.size() > pos_ ? c.size() - pos_ : 0;
  if (n > avail) {
    pos_ = c.size();
    return Status::NotFound("in-memory file skipped past end");
  }
  pos_ += static_cast<size_t>(n);
  return Status::OK();
}

Status MemRandomAccessFile::Read(uint64_t offset, size_t n, Slice* result, char* scratch) const {
  const std::string& c = file_->contents;
  if (offset >= c.size()) {
    *result = Slice(scratch, 0);
    return Status::OK();
  }
  size_t k = c.size() - static_cast<size_t>(offset);
  if (k > n) k = n;
  memcpy(scratch, c.data() + offset, k);
  *result = Slice(scratch, k);
  return Status::OK();
}

Status ParallelRead(const RandomAccessFile* file, uint64_t offset, size_t n, char* buf, int threads) {
  const size_t kMinPerThread = 4u << 20;
  int t = threads < 1 ? 1 : threads;
  if (n / kMinPerThread < static_cast<size_t>(t)) t = static_cast<int>(n / kMinPerThread) + 1;
  std::vector<Status> st(t);
  auto work = [&](int k) {
    const size_t lo = n * k / t, hi = n * (k + 1) / t;
    size_t done = lo;
    while (done < hi) {
      Slice got;
      Status s = file->Read(offset + done, hi - done, &got, buf + done);
      if (!s.ok()) {
        st[k] = s;
        return;
      }
      if (got.size() == 0) {
        st[k] = Status::IOError("unexpected end of file");
        return;
      }
      if (got.data() != buf + done) memcpy(buf + done, got.data(), got.size());
      done += got.size();
    }
  };
  if (t == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int k = 1; k < t; ++k) pool.emplace_back(work, k);
    work(0);
    for (auto& th : pool) th.join();
  }
  for (const Status& s : st)
    if (!s.ok()) return s;
  return Status::OK();
}

}  // namespace terarkdb_amd
