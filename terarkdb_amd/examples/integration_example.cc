// The call sequences a TerarkDB maintainer would write when binding the engine, as one runnable
// program: group-commit log writing, recovery of that log through the reference-shaped reader,
// table block trailers (stamp, then verify with one damaged block), whole-file table verification
// and the backup engine's file checksum.  Each step checks its own result; the exit status is 0 when
// everything held.  Usage: integration_example <scratch directory>
#include <stdio.h>
#include <string.h>

#include <memory>
#include <string>
#include <vector>

#include "../../include/tcrc.h"
#include "../host/block_trailer.h"
#include "../host/file_checksum.h"
#include "../host/sst_verify.h"
#include "../host/wal_reader.h"
#include "../host/wal_writer.h"

namespace ta = terarkdb_amd;

#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #cond); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

namespace {
struct Collecting : ta::log::Reader::Reporter {
  size_t dropped = 0;
  std::string last;
  void Corruption(size_t bytes, const ta::Status& s) override {
    dropped += bytes;
    last = s.ToString();
  }
};
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <scratch directory>\n", argv[0]);
    return 2;
  }
  const std::string dir = argv[1];
  tcrc_ctx* ctx = nullptr;
  if (tcrc_init(0, &ctx) != TCRC_OK) {
    fprintf(stderr, "tcrc unavailable: %s\n", tcrc_last_error());
    return 3;
  }
  printf("CRC engine: %s\n", tcrc_describe(ctx));

  // 1. write-ahead log, group commit (db/db_impl_write.cc:834-939 shape)
  const std::string log_path = dir + "/000042.log";
  const uint64_t log_number = 42;
  std::vector<std::string> batches;
  for (int i = 0; i < 500; ++i) batches.push_back(std::string(20 + (i * 37) % 5000, static_cast<char>('a' + i % 26)));
  batches.push_back(std::string(100000, 'z'));  // spans several 32 KiB blocks
  {
    ta::log::DeviceRecordStamper stamper(ctx);
    std::unique_ptr<ta::WritableFile> file;
    CHECK(ta::NewPosixWritableFile(log_path, &file).ok());
    ta::log::Writer w(std::move(file), log_number, /*recycle_log_files=*/false, /*manual_flush=*/true, &stamper);
    for (const std::string& b : batches) CHECK(w.AddRecord(ta::Slice(b)).ok());
    CHECK(w.WriteBuffer().ok());  // one submission for the whole group
    printf("log: %zu records in, %llu physical records written\n", batches.size(),
           (unsigned long long)w.physical_records());
  }

  // 2. recovery: the file is checked by the engine first, then the reference's read loop (db/db_impl_open.cc:737-784)
  {
    ta::log::DeviceRecordVerifier verifier(ctx);
    std::unique_ptr<ta::RandomAccessFile> ra;
    CHECK(ta::NewPosixRandomAccessFile(log_path, &ra).ok());
    const size_t size = static_cast<size_t>(ra->Size());
    auto verdicts = std::make_shared<ta::log::VerifiedLogImage>();
    // the file goes through the engine in pieces, the read of one overlapping the engine's work on the previous one
    CHECK(ta::log::VerifyLogFile(&verifier, ra.get(), log_number, verdicts.get()).ok());
    CHECK(verdicts->mismatches == 0 && verdicts->size == size);
    std::unique_ptr<ta::SequentialFile> seq;
    CHECK(ta::NewPosixSequentialFile(log_path, &seq).ok());
    Collecting reporter;
    ta::log::Reader reader(std::move(seq), &reporter, /*checksum=*/true, log_number, /*retry_after_eof=*/false, &verifier);
    reader.AttachVerifiedImage(verdicts);
    std::string scratch;
    ta::Slice record;
    size_t n = 0;
    while (reader.ReadRecord(&record, &scratch, ta::WALRecoveryMode::kPointInTimeRecovery)) {
      CHECK(n < batches.size() && record.ToString() == batches[n]);
      ++n;
    }
    CHECK(n == batches.size() && reporter.dropped == 0 && reader.verifier_calls() == 0);
    printf("recovery: %zu records back, verdicts from the engine for %zu bytes\n", n, size);
  }

  // 3. table block trailers: WriteRawBlock / CheckBlockChecksum in batches
  const std::string sst_path = dir + "/000043.blocks";
  std::vector<std::string> blocks;
  for (int i = 0; i < 200; ++i) blocks.push_back(std::string(3000 + (i * 97) % 2000, static_cast<char>(i)));
  std::vector<ta::BlockHandle> handles;
  uint64_t offset = 0;
  {
    std::vector<ta::Slice> contents(blocks.begin(), blocks.end());
    std::vector<char> types(blocks.size(), 0);  // kNoCompression
    std::unique_ptr<ta::WritableFile> file;
    CHECK(ta::NewPosixWritableFile(sst_path, &file).ok());
    CHECK(ta::WriteRawBlocks(ctx, file.get(), &offset, contents, types, ta::kCRC32c, &handles).ok());
    CHECK(file->Flush().ok() && file->Close().ok());
  }
  {
    std::unique_ptr<ta::RandomAccessFile> ra;
    CHECK(ta::NewPosixRandomAccessFile(sst_path, &ra).ok());
    std::string image(static_cast<size_t>(ra->Size()), '\0');
    CHECK(ta::ParallelRead(ra.get(), 0, image.size(), &image[0], 2).ok());
    std::vector<ta::Status> per_block;
    CHECK(ta::CheckBlockChecksums(ctx, ta::Slice(image), 0, handles, ta::kCRC32c, sst_path, &per_block).ok());
    image[handles[17].offset + 5] ^= 0x10;
    ta::Status first = ta::CheckBlockChecksums(ctx, ta::Slice(image), 0, handles, ta::kCRC32c, sst_path, &per_block);
    CHECK(first.IsCorruption() && !per_block[17].ok() && per_block[16].ok());
    CHECK(first.ToString().find("block checksum mismatch: expected ") != std::string::npos);
    printf("blocks: %zu trailers stamped and verified; damaged block reported as: %s\n", blocks.size(),
           first.ToString().c_str());
  }

  // 4. whole-file checksum (backup engine) and whole-file table verification of something that is not a table
  {
    uint32_t crc = 0;
    uint64_t size = 0;
    CHECK(ta::CalculateChecksum(ctx, sst_path, 0, &crc, &size).ok() && size == offset);
    uint32_t crc2 = 0;
    CHECK(ta::CalculateChecksum(ctx, sst_path, 1000, &crc2, &size).ok() && size == 1000 && crc2 != crc);
    ta::Status s = ta::VerifySstFileChecksum(ctx, sst_path);
    CHECK(s.IsCorruption());  // no footer: "Bad table magic number ..."
    printf("file checksum %u over %llu bytes; as a table file: %s\n", crc, (unsigned long long)offset, s.ToString().c_str());
  }
  tcrc_shutdown(ctx);
  printf("integration example ok\n");
  return 0;
}
