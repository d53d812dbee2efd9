// Command-line front end of the engine, shaped after the reference's tools for this path:
//   tcrc crc32c [--block_size=N] [--bytes=N] [--device=D]      db_bench "crc32c" (tools/db_bench_tool.cc:3069-3088)
//   tcrc verify FILE.sst...                                    sst_dump --command=verify (tools/sst_dump_tool.cc:656-665)
//   tcrc dump_wal --walfile=F [--header] [--print_value]       ldb dump_wal (tools/ldb_cmd.cc:1967-2037)
//   tcrc verify_db DIR [--shard=I/N | --gpus=0,1,..]                           DBImpl::VerifyChecksum over the files of a
//                                                              database directory (db/db_impl.cc:4657-4690), plus a
//                                                              scan of its logs and MANIFEST files
//   tcrc file_checksum FILE...                                 backup engine's CalculateChecksum
//                                                              (utilities/backupable/backupable_db.cc:1414-1452)
// All arithmetic runs on the GPU; without one every command fails at start-up.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "../../include/tcrc.h"
#include "../host/db_verify.h"
#include "../host/file_checksum.h"
#include "../host/sst_verify.h"
#include "../host/wal_dump.h"

using namespace terarkdb_amd;

namespace {

double NowMicros() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool Flag(const char* arg, const char* name, std::string* value) {
  const size_t n = strlen(name);
  if (strncmp(arg, name, n) != 0) return false;
  if (arg[n] == '\0') {
    value->clear();
    return true;
  }
  if (arg[n] != '=') return false;
  *value = arg + n + 1;
  return true;
}

// Stats::Report of db_bench (tools/db_bench_tool.cc:1867-1888)
void Report(const char* name, double elapsed_micros, uint64_t done, uint64_t bytes, const std::string& message) {
  if (done < 1) done = 1;
  const double elapsed = elapsed_micros * 1e-6;
  char rate[100];
  snprintf(rate, sizeof(rate), "%6.1f MB/s", (bytes / 1048576.0) / elapsed);
  std::string extra = rate;
  if (!message.empty()) extra += " " + message;
  printf("%-12s : %11.3f micros/op %ld ops/sec; %s\n", name, elapsed * 1e6 / done, (long)(done / elapsed),
         extra.c_str());
}

int CmdCrc32c(tcrc_ctx* ctx, int argc, char** argv) {
  uint64_t block_size = 4096, total = 500ull * 1048576;
  std::string v;
  for (int i = 0; i < argc; ++i) {
    if (Flag(argv[i], "--block_size", &v)) block_size = strtoull(v.c_str(), nullptr, 10);
    else if (Flag(argv[i], "--bytes", &v)) total = strtoull(v.c_str(), nullptr, 10);
  }
  if (block_size == 0 || block_size >= 0xffffffffull) {
    fprintf(stderr, "bad --block_size\n");
    return 2;
  }
  const std::string label = "(" + std::to_string(block_size) + " per op)";
  const uint64_t ops = (total + block_size - 1) / block_size;

  // 1. one call per block, like the reference's loop: every op pays a host-device round trip
  {
    std::string data(block_size, 'x');
    const uint64_t few = ops < 2000 ? ops : 2000;
    uint32_t crc = 0;
    tcrc_value(ctx, data.data(), data.size(), &crc);  // first call sets up streams and buffers
    const double t0 = NowMicros();
    for (uint64_t i = 0; i < few; ++i)
      if (tcrc_value(ctx, data.data(), data.size(), &crc) != TCRC_OK) {
        fprintf(stderr, "%s\n", tcrc_last_error());
        return 1;
      }
    const double t1 = NowMicros();
    fprintf(stderr, "... crc=0x%x\r", crc);
    Report("crc32c", t1 - t0, few, few * block_size, label + " one call per op");
  }
  // 2. the same ~500 MB as one submission of `ops` blocks in host memory
  {
    char* buf = static_cast<char*>(tcrc_host_alloc(ctx, ops * block_size));
    std::vector<uint32_t> out(ops);
    if (!buf) {
      fprintf(stderr, "cannot allocate %llu bytes of pinned memory\n", (unsigned long long)(ops * block_size));
      return 1;
    }
    memset(buf, 'x', ops * block_size);
    for (int pass = 0; pass < 2; ++pass) {  // the first pass grows the device staging
      const double t0 = NowMicros();
      if (tcrc_host_blocks_stamp(ctx, buf, ops, (uint32_t)block_size, block_size, 0, -1, 0, out.data()) != TCRC_OK) {
        fprintf(stderr, "%s\n", tcrc_last_error());
        return 1;
      }
      const double t1 = NowMicros();
      if (pass == 1) {
        fprintf(stderr, "... crc=0x%x\r", out[ops - 1]);
        Report("crc32c_batch", t1 - t0, ops, ops * block_size, label + " one submission, host memory to host memory");
      }
    }
    tcrc_host_free(ctx, buf);
  }
  return 0;
}

int CmdVerify(tcrc_ctx* ctx, int argc, char** argv) {
  int bad = 0;
  for (int i = 0; i < argc; ++i) {
    if (argv[i][0] == '-') continue;
    const std::string filename = argv[i];
    SstVerifyStats st;
    const double t0 = NowMicros();
    Status s = VerifySstFileChecksum(ctx, filename, SstVerifyOptions(), &st);
    const double t1 = NowMicros();
    if (!s.ok()) {
      fprintf(stderr, "%s is corrupted: %s\n", filename.c_str(), s.ToString().c_str());
      ++bad;
    } else {
      fprintf(stdout, "The file is ok\n");
    }
    fprintf(stderr, "%s: %llu blocks, %llu bytes covered, %.1f MB/s\n", filename.c_str(),
            (unsigned long long)(st.meta_blocks + st.index_blocks + st.data_blocks),
            (unsigned long long)st.bytes_covered, st.bytes_covered / 1048576.0 / ((t1 - t0) * 1e-6));
  }
  return bad ? 1 : 0;
}

int CmdDumpWal(tcrc_ctx* ctx, int argc, char** argv) {
  std::string walfile, v;
  bool header = false, values = false;
  for (int i = 0; i < argc; ++i) {
    if (Flag(argv[i], "--walfile", &v)) walfile = v;
    else if (Flag(argv[i], "--header", &v)) header = true;
    else if (Flag(argv[i], "--print_value", &v)) values = true;
  }
  if (walfile.empty()) {
    fprintf(stderr, "Argument --walfile must be specified.\n");
    return 2;
  }
  log::DeviceRecordVerifier verifier(ctx);
  std::string out, err;
  Status s = DumpWalFile(&verifier, walfile, header, values, &out, &err);
  fputs(out.c_str(), stdout);
  fputs(err.c_str(), stderr);
  if (!s.ok()) {
    fprintf(stderr, "%s\n", s.ToString().c_str());
    return 1;
  }
  return 0;
}

int CmdVerifyDb(tcrc_ctx* ctx, int argc, char** argv) {
  std::string dir, v;
  uint32_t shard = 0, shards = 1;
  std::vector<int> gpus;  // --gpus=0,1,...: the directory's files over a group of GPUs in this process
  for (int i = 0; i < argc; ++i) {
    if (Flag(argv[i], "--shard", &v)) {
      if (sscanf(v.c_str(), "%u/%u", &shard, &shards) != 2) {
        fprintf(stderr, "--shard=I/N\n");
        return 2;
      }
    } else if (Flag(argv[i], "--gpus", &v)) {
      for (size_t at = 0; at < v.size();) {
        gpus.push_back(atoi(v.c_str() + at));
        const size_t comma = v.find(',', at);
        if (comma == std::string::npos) break;
        at = comma + 1;
      }
    } else if (argv[i][0] != '-') {
      dir = argv[i];
    }
  }
  if (dir.empty()) {
    fprintf(stderr, "verify_db needs a directory\n");
    return 2;
  }
  DbVerifyReport rep;
  const double t0 = NowMicros();
  Status s;
  if (!gpus.empty()) {
    tcrc_group* group = nullptr;
    if (tcrc_group_init(gpus.data(), (int)gpus.size(), &group) != TCRC_OK) {
      fprintf(stderr, "%s\n", tcrc_last_error());
      return 2;
    }
    s = VerifyDbDirectoryGroup(group, dir, &rep);
    tcrc_group_shutdown(group);
  } else {
    s = VerifyDbDirectory(ctx, dir, shard, shards, &rep);
  }
  const double t1 = NowMicros();
  printf("table files: %llu verified (%llu blocks, %llu bytes covered), %llu declined\n",
         (unsigned long long)rep.table_files, (unsigned long long)rep.table_blocks,
         (unsigned long long)rep.table_bytes_covered, (unsigned long long)rep.table_files_declined);
  printf("log files: %llu (%llu bytes, %llu records, %llu with a checksum mismatch)\n", (unsigned long long)rep.log_files,
         (unsigned long long)rep.log_bytes, (unsigned long long)rep.log_records, (unsigned long long)rep.log_bad_records);
  for (const std::string& n : rep.log_notes) printf("  %s\n", n.c_str());
  fprintf(stderr, "%.1f MB/s\n", (rep.table_bytes_covered + rep.log_bytes) / 1048576.0 / ((t1 - t0) * 1e-6));
  if (!s.ok()) {
    fprintf(stderr, "%s is corrupted: %s\n", rep.first_bad_file.c_str(), s.ToString().c_str());
    return 1;
  }
  printf("OK\n");
  return rep.log_bad_records ? 1 : 0;
}

// Checksum scan of log files (NNNNNN.log, MANIFEST-NNNNNN): what recovery's reader would accept and drop.
int CmdVerifyWal(tcrc_ctx* ctx, int argc, char** argv) {
  int bad = 0;
  for (int i = 0; i < argc; ++i) {
    if (argv[i][0] == '-') continue;
    std::unique_ptr<RandomAccessFile> file;
    Status s = NewPosixRandomAccessFile(argv[i], &file);
    log::VerifiedLogImage image;
    const double t0 = NowMicros();
    if (s.ok()) {
      log::DeviceRecordVerifier verifier(ctx);
      s = log::VerifyLogFile(&verifier, file.get(), LogNumberFromFileName(argv[i]), &image);
    }
    const double t1 = NowMicros();
    if (!s.ok()) {
      fprintf(stderr, "%s: %s\n", argv[i], s.ToString().c_str());
      ++bad;
      continue;
    }
    const uint32_t end = image.blocks.empty() ? TCRC_WAL_END_CLEAN : image.blocks.back().end_code;
    const bool torn = end == TCRC_WAL_END_SHORT_AT_EOF || end == TCRC_WAL_END_BAD_LEN_AT_EOF;
    printf("%s: %llu bytes, %llu physical records, %llu with a checksum mismatch", argv[i],
           (unsigned long long)image.size, (unsigned long long)image.records, (unsigned long long)image.mismatches);
    if (image.mismatches) printf(" (first in 32 KiB block %llu)", (unsigned long long)image.first_bad_block);
    printf("%s\n", torn ? ", torn record at the end of the file" : "");
    fprintf(stderr, "%s: %.1f MB/s\n", argv[i], image.size / 1048576.0 / ((t1 - t0) * 1e-6));
    if (image.mismatches) ++bad;
  }
  return bad ? 1 : 0;
}

int CmdFileChecksum(tcrc_ctx* ctx, int argc, char** argv) {
  int bad = 0;
  for (int i = 0; i < argc; ++i) {
    if (argv[i][0] == '-') continue;
    uint32_t crc = 0;
    uint64_t size = 0;
    const double t0 = NowMicros();
    Status s = CalculateChecksum(ctx, std::string(argv[i]), 0, &crc, &size);
    const double t1 = NowMicros();
    if (!s.ok()) {
      fprintf(stderr, "%s: %s\n", argv[i], s.ToString().c_str());
      ++bad;
      continue;
    }
    printf("%s crc32c %u size %llu\n", argv[i], crc, (unsigned long long)size);
    fprintf(stderr, "%s: %.1f MB/s\n", argv[i], size / 1048576.0 / ((t1 - t0) * 1e-6));
  }
  return bad ? 1 : 0;
}

void Usage() {
  fprintf(stderr,
          "usage: tcrc [--device=N] <command> [args]\n"
          "  crc32c [--block_size=N] [--bytes=N]\n"
          "  verify FILE.sst ...\n"
          "  dump_wal --walfile=FILE [--header] [--print_value]\n"
          "  verify_wal FILE.log ...\n"
          "  verify_db DIR [--shard=I/N | --gpus=0,1,...]\n"
          "  file_checksum FILE ...\n");
}

}  // namespace

int main(int argc, char** argv) {
  int device = 0, at = 1;
  std::string v;
  if (at < argc && Flag(argv[at], "--device", &v)) {
    device = atoi(v.c_str());
    ++at;
  }
  if (at >= argc) {
    Usage();
    return 2;
  }
  const std::string cmd = argv[at++];
  if (cmd != "crc32c" && cmd != "verify" && cmd != "dump_wal" && cmd != "file_checksum" && cmd != "verify_db" &&
      cmd != "verify_wal") {
    Usage();
    return 2;
  }
  tcrc_ctx* ctx = nullptr;
  if (tcrc_init(device, &ctx) != TCRC_OK) {
    fprintf(stderr, "tcrc: %s\n", tcrc_last_error());
    return 3;
  }
  int rc;
  if (cmd == "crc32c") rc = CmdCrc32c(ctx, argc - at, argv + at);
  else if (cmd == "verify") rc = CmdVerify(ctx, argc - at, argv + at);
  else if (cmd == "dump_wal") rc = CmdDumpWal(ctx, argc - at, argv + at);
  else if (cmd == "verify_db") rc = CmdVerifyDb(ctx, argc - at, argv + at);
  else if (cmd == "verify_wal") rc = CmdVerifyWal(ctx, argc - at, argv + at);
  else rc = CmdFileChecksum(ctx, argc - at, argv + at);
  tcrc_shutdown(ctx);
  return rc;
}
