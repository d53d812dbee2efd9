// Checksum verification of a whole database directory at the file level: the loop of
// DBImpl::VerifyChecksum (db/db_impl.cc:4657-4690), which calls VerifySstFileChecksum
// (db/convenience.cc:35-64) for every live table file and stops at the first error, plus a scan of
// the write-ahead logs and MANIFEST files the way recovery would meet them
// (db/log_reader.cc:314-406).  The list of live files comes from the caller or, for the directory
// form, from the file names (util/filename.cc:229-343: NNNNNN.sst, NNNNNN.log, MANIFEST-NNNNNN);
// reading the version set is not this engine's business.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "sst_verify.h"
#include "wal_format.h"

namespace terarkdb_amd {

struct DbVerifyReport {
  uint64_t table_files = 0, table_files_declined = 0, table_blocks = 0, table_bytes_covered = 0;
  // log_bad_records counts EVERY physical record whose checksum fails; the reference's reader stops looking at a
  // 32 KiB block at its first bad record (db/log_reader.cc:391-397), so it can meet fewer.
  uint64_t log_files = 0, log_bytes = 0, log_records = 0, log_bad_records = 0;
  std::string first_bad_file;  // the file the returned status is about
  std::vector<std::string> log_notes;  // one line per log file with a bad record or a torn tail
};

// Every file of `files` whose index i satisfies i % shard_count == shard_index (files can be split over
// several GPUs this way) is verified in order; the first non-OK table status ends the run, as in the
// reference.  Tables the engine declines (xxHash checksums, other table formats) are counted and skipped.
Status VerifyTableFiles(tcrc_ctx* ctx, const std::vector<std::string>& files, uint32_t shard_index, uint32_t shard_count,
                        DbVerifyReport* report, const SstVerifyOptions& options = SstVerifyOptions());

// All table files of `dir` in file-number order, then its logs and MANIFEST files.  With several shards the
// files are dealt by size (largest first, to the shard with the fewest bytes so far), so that the shards of a
// directory with files of very different sizes finish together; every shard computes the same deal.  A bad log record is
// not an error of the call (recovery policy decides what it means): it is counted and noted.
Status VerifyDbDirectory(tcrc_ctx* ctx, const std::string& dir, uint32_t shard_index, uint32_t shard_count,
                         DbVerifyReport* report, const SstVerifyOptions& options = SstVerifyOptions());

// The same over a group of GPUs in one process (tcrc_group_init): shard m of the directory's files goes to member m,
// every member on a host thread of its own, the way several processes with --shard=m/n would do it; the reports
// add up.  If several shards fail, the status is that of the failing table with the smallest file number - the
// first one DBImpl::VerifyChecksum's loop over the files (db/db_impl.cc:4667-4686) would have met among them.
Status VerifyDbDirectoryGroup(tcrc_group* group, const std::string& dir, DbVerifyReport* report,
                              const SstVerifyOptions& options = SstVerifyOptions());

}  // namespace terarkdb_amd
