#include "wal_dump.h"

#include <memory>
#include <vector>

#include "wal_reader.h"

namespace terarkdb_amd {

namespace {

class TextReporter : public log::Reader::Reporter {
 public:
  explicit TextReporter(std::string* err) : err_(err) {}
  void Corruption(size_t /*bytes*/, const Status& s) override {
    *err_ += "Corruption detected in log file " + s.ToString() + "\n";
  }

 private:
  std::string* err_;
};

std::string Hex(const Slice& s) {  // LDBCommand::StringToHex (tools/ldb_cmd.cc:721-725)
  static const char tab[] = "0123456789ABCDEF";
  std::string r("0x");
  for (size_t i = 0; i < s.size(); ++i) {
    const unsigned char c = static_cast<unsigned char>(s.data()[i]);
    r.push_back(tab[c >> 4]);
    r.push_back(tab[c & 15]);
  }
  return r;
}

bool GetVarint32(Slice* in, uint32_t* v) {
  uint32_t result = 0;
  const char* p = in->data();
  const char* limit = p + in->size();
  for (uint32_t shift = 0; shift <= 28 && p < limit; shift += 7) {
    const uint32_t byte = static_cast<unsigned char>(*p++);
    if (byte & 128) {
      result |= (byte & 127) << shift;
    } else {
      *v = result | (byte << shift);
      in->remove_prefix(p - in->data());
      return true;
    }
  }
  return false;
}

bool GetLengthPrefixed(Slice* in, Slice* out) {
  uint32_t len;
  if (!GetVarint32(in, &len) || in->size() < len) return false;
  *out = Slice(in->data(), len);
  in->remove_prefix(len);
  return true;
}

uint64_t Fixed64(const char* p) {
  return static_cast<uint64_t>(DecodeFixed32(p)) | (static_cast<uint64_t>(DecodeFixed32(p + 4)) << 32);
}

}  // namespace

uint64_t LogNumberFromFileName(const std::string& path) {
  std::string name = path;
  const size_t slash = name.rfind('/');
  if (slash != std::string::npos) name = name.substr(slash + 1);
  bool manifest = false;
  if (name.compare(0, 9, "MANIFEST-") == 0) {  // parses as a descriptor file with its number
    name = name.substr(9);
    manifest = true;
  }
  uint64_t num = 0;
  size_t i = 0;
  for (; i < name.size() && name[i] >= '0' && name[i] <= '9'; ++i) {
    const uint64_t d = static_cast<uint64_t>(name[i] - '0');
    if (num > (~0ull - d) / 10) return 0;  // overflow: ConsumeDecimalNumber fails
    num = num * 10 + d;
  }
  if (i == 0) return 0;
  if (manifest) return i == name.size() ? num : 0;
  // the reference accepts any known suffix (log, sst, ldb, dbtmp) and keeps the number
  const std::string suffix = name.substr(i);
  if (suffix == ".log" || suffix == ".sst" || suffix == ".ldb" || suffix == ".dbtmp") return num;
  return 0;
}

void WriteBatchRow(const Slice& rep, uint64_t physical_offset, bool print_values, std::string* row) {
  // WriteBatchInternal::Sequence / Count / ByteSize: fixed64, fixed32, size of the representation
  *row += std::to_string(Fixed64(rep.data())) + ",";
  *row += std::to_string(DecodeFixed32(rep.data() + 8)) + ",";
  *row += std::to_string(rep.size()) + ",";
  *row += std::to_string(physical_offset) + ",";
  Slice in(rep.data() + 12, rep.size() - 12);
  bool empty_batch = true;
  auto put_merge = [&](const Slice& key, const Slice& value) {  // InMemoryHandler::commonPutMerge
    if (print_values) {
      *row += Hex(key) + " : " + Hex(value) + " ";
    } else {
      *row += Hex(key) + " ";
    }
  };
  while (!in.empty()) {
    const unsigned char tag = static_cast<unsigned char>(in.data()[0]);
    in.remove_prefix(1);
    uint32_t cf = 0;
    Slice key, value;
    switch (tag) {
      case 0x5:  // kTypeColumnFamilyValue
        if (!GetVarint32(&in, &cf)) return;
        // fall through
      case 0x1:  // kTypeValue
        if (!GetLengthPrefixed(&in, &key) || !GetLengthPrefixed(&in, &value)) return;
        *row += "PUT(" + std::to_string(cf) + ") : ";
        put_merge(key, value);
        empty_batch = false;
        break;
      case 0x4:  // kTypeColumnFamilyDeletion
        if (!GetVarint32(&in, &cf)) return;
        // fall through
      case 0x0:  // kTypeDeletion
        if (!GetLengthPrefixed(&in, &key)) return;
        *row += "DELETE(" + std::to_string(cf) + ") : " + Hex(key) + " ";
        empty_batch = false;
        break;
      case 0x8:  // kTypeColumnFamilySingleDeletion
        if (!GetVarint32(&in, &cf)) return;
        // fall through
      case 0x7:  // kTypeSingleDeletion
        if (!GetLengthPrefixed(&in, &key)) return;
        *row += "SINGLE_DELETE(" + std::to_string(cf) + ") : " + Hex(key) + " ";
        empty_batch = false;
        break;
      case 0xE:  // kTypeColumnFamilyRangeDeletion
        if (!GetVarint32(&in, &cf)) return;
        // fall through
      case 0xF:  // kTypeRangeDeletion
        if (!GetLengthPrefixed(&in, &key) || !GetLengthPrefixed(&in, &value)) return;
        *row += "DELETE_RANGE(" + std::to_string(cf) + ") : " + Hex(key) + " " + Hex(value) + " ";
        empty_batch = false;
        break;
      case 0x6:  // kTypeColumnFamilyMerge
        if (!GetVarint32(&in, &cf)) return;
        // fall through
      case 0x2:  // kTypeMerge
        if (!GetLengthPrefixed(&in, &key) || !GetLengthPrefixed(&in, &value)) return;
        *row += "MERGE(" + std::to_string(cf) + ") : ";
        put_merge(key, value);
        empty_batch = false;
        break;
      case 0x3:  // kTypeLogData: the dump handler prints nothing for it
        if (!GetLengthPrefixed(&in, &value)) return;
        empty_batch = false;
        break;
      case 0xD:  // kTypeNoop
        *row += "NOOP ";
        empty_batch = true;
        break;
      case 0x9:   // kTypeBeginPrepareXID
      case 0x12:  // kTypeBeginPersistedPrepareXID
        *row += "BEGIN_PREPARE(false) ";
        empty_batch = false;
        break;
      case 0x13:  // kTypeBeginUnprepareXID
        *row += "BEGIN_PREPARE(true) ";
        empty_batch = false;
        break;
      case 0xA:  // kTypeEndPrepareXID
        if (!GetLengthPrefixed(&in, &value)) return;
        *row += "END_PREPARE(" + Hex(value) + ") ";
        empty_batch = true;
        break;
      case 0xB:  // kTypeCommitXID
        if (!GetLengthPrefixed(&in, &value)) return;
        *row += "COMMIT(" + Hex(value) + ") ";
        empty_batch = true;
        break;
      case 0xC:  // kTypeRollbackXID
        if (!GetLengthPrefixed(&in, &value)) return;
        *row += "ROLLBACK(" + Hex(value) + ") ";
        empty_batch = true;
        break;
      default:  // unknown tag, kTypeValueIndex, kTypeMergeIndex: Iterate stops with Corruption, the dump goes on
        return;
    }
  }
  (void)empty_batch;
}

Status DumpWalFile(log::RecordVerifier* verifier, const std::string& wal_file, bool print_header, bool print_values,
                   std::string* out, std::string* err) {
  // The whole file goes through the device pass in large pieces, then the reader reads it again for the payloads.
  std::unique_ptr<RandomAccessFile> ra;
  Status s = NewPosixRandomAccessFile(wal_file, &ra);
  if (!s.ok()) {
    *err += "Error: Failed to open WAL file " + s.ToString() + "\n";
    return s;
  }
  const uint64_t log_number = LogNumberFromFileName(wal_file);
  auto image = std::make_shared<log::VerifiedLogImage>();
  s = log::VerifyLogFile(verifier, ra.get(), log_number, image.get());
  if (!s.ok()) return s;

  std::unique_ptr<SequentialFile> file;
  s = NewPosixSequentialFile(wal_file, &file);
  if (!s.ok()) return s;
  TextReporter reporter(err);
  log::Reader reader(std::move(file), &reporter, true /* checksum */, log_number, false /* retry_after_eof */,
                     verifier);
  reader.AttachVerifiedImage(image);
  if (print_header) {
    *out += "Sequence,Count,ByteSize,Physical Offset,Key(s)";
    if (print_values) *out += " : value ";
    *out += "\n";
  }
  std::string scratch;
  Slice record;
  while (reader.ReadRecord(&record, &scratch)) {
    if (record.size() < 12) {  // WriteBatchInternal::kHeader
      reporter.Corruption(record.size(), Status::Corruption("log record too small"));
    } else {
      WriteBatchRow(record, reader.LastRecordOffset(), print_values, out);
      *out += "\n";
    }
  }
  return Status::OK();
}

}  // namespace terarkdb_amd
