#include "db_verify.h"

#include <dirent.h>
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <memory>
#include <thread>

#include "wal_dump.h"

namespace terarkdb_amd {

namespace {

struct Named {
  uint64_t number;
  std::string path;
  uint64_t bytes = 0;
  bool operator<(const Named& o) const { return number != o.number ? number < o.number : path < o.path; }
};

// Deals files to shards so that the shards' byte totals come out even: largest file first, each to the
// shard that has the least so far (ties: the lower shard).  Every rank computes the same deal from the
// same directory listing.  files must be sorted; returns this shard's files, still sorted.
std::vector<Named> DealBySize(const std::vector<Named>& files, uint32_t shard_index, uint32_t shard_count) {
  std::vector<size_t> order(files.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return files[a].bytes > files[b].bytes; });
  std::vector<uint64_t> load(shard_count, 0);
  std::vector<char> mine(files.size(), 0);
  for (size_t k : order) {
    uint32_t best = 0;
    for (uint32_t s = 1; s < shard_count; ++s)
      if (load[s] < load[best]) best = s;
    load[best] += files[k].bytes + 1;  // empty files still take turns
    if (best == shard_index) mine[k] = 1;
  }
  std::vector<Named> out;
  for (size_t i = 0; i < files.size(); ++i)
    if (mine[i]) out.push_back(files[i]);
  return out;
}

uint64_t FileBytes(const std::string& path) {
  struct stat st;
  return stat(path.c_str(), &st) == 0 ? static_cast<uint64_t>(st.st_size) : 0;
}

// NNNNNN<suffix> -> number (util/filename.cc:306-343); false for other names
bool NumberedName(const std::string& name, const char* suffix, uint64_t* number) {
  size_t i = 0;
  uint64_t n = 0;
  for (; i < name.size() && name[i] >= '0' && name[i] <= '9'; ++i) n = n * 10 + static_cast<uint64_t>(name[i] - '0');
  if (i == 0 || name.compare(i, std::string::npos, suffix) != 0) return false;
  *number = n;
  return true;
}

Status ScanLogFile(tcrc_ctx* ctx, const std::string& path, DbVerifyReport* report) {
  std::unique_ptr<RandomAccessFile> file;
  Status s = NewPosixRandomAccessFile(path, &file);
  if (!s.ok()) return s;
  const size_t size = static_cast<size_t>(file->Size());
  ++report->log_files;
  report->log_bytes += size;
  if (size == 0) return Status::OK();
  log::DeviceRecordVerifier verifier(ctx);
  log::VerifiedLogImage image;
  // A MANIFEST is read with log number 0 (db/version_set.cc:3813-3815, 4038-4040), a write-ahead log with the
  // number in its name (db/db_impl_open.cc:771-773); it only matters for records with the recyclable header.
  const std::string base = path.substr(path.find_last_of('/') + 1);
  const uint64_t log_number = base.compare(0, 9, "MANIFEST-") == 0 ? 0 : LogNumberFromFileName(path);
  s = log::VerifyLogFile(&verifier, file.get(), log_number, &image);
  if (!s.ok()) return s;
  const std::vector<tcrc_wal_block>& blocks = image.blocks;
  const uint64_t records = image.records;
  const uint64_t counts[TCRC_CNT_WORDS] = {image.mismatches, image.first_bad_block, image.records};
  report->log_records += records;
  report->log_bad_records += counts[TCRC_CNT_MISMATCH];
  const tcrc_wal_block& tail = blocks.back();
  const bool torn = tail.end_code == TCRC_WAL_END_SHORT_AT_EOF || tail.end_code == TCRC_WAL_END_BAD_LEN_AT_EOF;
  if (counts[TCRC_CNT_MISMATCH] || torn) {
    std::string note = path + ": " + std::to_string(records) + " records";
    if (counts[TCRC_CNT_MISMATCH])
      note += ", " + std::to_string(counts[TCRC_CNT_MISMATCH]) + " with a checksum mismatch (first in 32 KiB block " +
              std::to_string(counts[TCRC_CNT_FIRST_BAD]) + ")";
    if (torn) note += ", torn record at the end of the file";
    report->log_notes.push_back(note);
  }
  return Status::OK();
}

}  // namespace

Status VerifyTableFiles(tcrc_ctx* ctx, const std::vector<std::string>& files, uint32_t shard_index, uint32_t shard_count,
                        DbVerifyReport* report, const SstVerifyOptions& options) {
  if (shard_count == 0 || shard_index >= shard_count) return Status::InvalidArgument("bad shard index / count");
  for (size_t i = 0; i < files.size(); ++i) {
    if (i % shard_count != shard_index) continue;
    SstVerifyStats st;
    Status s = VerifySstFileChecksum(ctx, files[i], options, &st);
    if (s.code() == Status::kNotSupported) {
      ++report->table_files_declined;
      continue;
    }
    ++report->table_files;
    report->table_blocks += st.meta_blocks + st.index_blocks + st.data_blocks;
    report->table_bytes_covered += st.bytes_covered;
    if (!s.ok()) {
      report->first_bad_file = files[i];
      return s;
    }
  }
  return Status::OK();
}

Status VerifyDbDirectory(tcrc_ctx* ctx, const std::string& dir, uint32_t shard_index, uint32_t shard_count,
                         DbVerifyReport* report, const SstVerifyOptions& options) {
  if (shard_count == 0 || shard_index >= shard_count) return Status::InvalidArgument("bad shard index / count");
  DIR* d = opendir(dir.c_str());
  if (!d) return Status::IOError("cannot list " + dir + ": " + strerror(errno));
  std::vector<Named> tables, logs;
  while (struct dirent* e = readdir(d)) {
    const std::string name = e->d_name;
    uint64_t num = 0;
    if (NumberedName(name, ".sst", &num) || NumberedName(name, ".ldb", &num)) {
      tables.push_back(Named{num, dir + "/" + name, 0});
    } else if (NumberedName(name, ".log", &num)) {
      logs.push_back(Named{num, dir + "/" + name, 0});
    } else if (name.compare(0, 9, "MANIFEST-") == 0 && NumberedName(name.substr(9), "", &num)) {
      logs.push_back(Named{num, dir + "/" + name, 0});
    }
  }
  closedir(d);
  std::sort(tables.begin(), tables.end());
  std::sort(logs.begin(), logs.end());
  for (Named& t : tables) t.bytes = FileBytes(t.path);
  for (Named& l : logs) l.bytes = FileBytes(l.path);
  std::vector<std::string> files;
  for (const Named& t : DealBySize(tables, shard_index, shard_count)) files.push_back(t.path);
  Status s = VerifyTableFiles(ctx, files, 0, 1, report, options);
  if (!s.ok()) return s;
  for (const Named& l : DealBySize(logs, shard_index, shard_count)) {
    s = ScanLogFile(ctx, l.path, report);
    if (!s.ok()) {
      report->first_bad_file = l.path;
      return s;
    }
  }
  return Status::OK();
}

Status VerifyDbDirectoryGroup(tcrc_group* group, const std::string& dir, DbVerifyReport* report,
                              const SstVerifyOptions& options) {
  const int members = tcrc_group_size(group);
  if (members <= 0) return Status::InvalidArgument("empty group");
  std::vector<DbVerifyReport> reps(members);
  std::vector<Status> sts(members);
  auto run = [&](int m) {
    sts[m] = VerifyDbDirectory(tcrc_group_ctx(group, m), dir, static_cast<uint32_t>(m), static_cast<uint32_t>(members),
                               &reps[m], options);
  };
  std::vector<std::thread> th;
  for (int m = 1; m < members; ++m) th.emplace_back(run, m);
  run(0);
  for (auto& t : th) t.join();
  int worst = -1;
  for (int m = 0; m < members; ++m) {
    const DbVerifyReport& r = reps[m];
    report->table_files += r.table_files;
    report->table_files_declined += r.table_files_declined;
    report->table_blocks += r.table_blocks;
    report->table_bytes_covered += r.table_bytes_covered;
    report->log_files += r.log_files;
    report->log_bytes += r.log_bytes;
    report->log_records += r.log_records;
    report->log_bad_records += r.log_bad_records;
    report->log_notes.insert(report->log_notes.end(), r.log_notes.begin(), r.log_notes.end());
    if (!sts[m].ok()) {
      // file names are NNNNNN.sst: shorter or lexically smaller base name = smaller number
      auto key = [](const std::string& path) {
        const std::string base = path.substr(path.find_last_of('/') + 1);
        return std::make_pair(base.size(), base);
      };
      if (worst < 0 || key(r.first_bad_file) < key(reps[worst].first_bad_file)) worst = m;
    }
  }
  std::sort(report->log_notes.begin(), report->log_notes.end());
  if (worst < 0) return Status::OK();
  report->first_bad_file = reps[worst].first_bad_file;
  return sts[worst];
}

}  // namespace terarkdb_amd
