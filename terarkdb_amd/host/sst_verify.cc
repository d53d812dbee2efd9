#include "sst_verify.h"

#include <string.h>

#include <algorithm>
#include <memory>
#include <thread>

namespace terarkdb_amd {

namespace {

// ---- varints (util/coding.h:86-100, util/coding.cc GetVarint64Ptr) -------------------------------

const char* GetVarint64Ptr(const char* p, const char* limit, uint64_t* value) {
  uint64_t result = 0;
  for (uint32_t shift = 0; shift <= 63 && p < limit; shift += 7) {
    const uint64_t byte = static_cast<unsigned char>(*p++);
    if (byte & 128) {
      result |= (byte & 127) << shift;
    } else {
      *value = result | (byte << shift);
      return p;
    }
  }
  return nullptr;
}

const char* GetVarint32Ptr(const char* p, const char* limit, uint32_t* value) {
  uint32_t result = 0;
  for (uint32_t shift = 0; shift <= 28 && p < limit; shift += 7) {
    const uint32_t byte = static_cast<unsigned char>(*p++);
    if (byte & 128) {
      result |= (byte & 127) << shift;
    } else {
      *value = result | (byte << shift);
      return p;
    }
  }
  return nullptr;
}

bool GetVarint64(Slice* input, uint64_t* value) {
  const char* q = GetVarint64Ptr(input->data(), input->data() + input->size(), value);
  if (!q) return false;
  input->remove_prefix(q - input->data());
  return true;
}

void PutVarint64(std::string* dst, uint64_t v) {
  while (v >= 128) {
    dst->push_back(static_cast<char>((v & 127) | 128));
    v >>= 7;
  }
  dst->push_back(static_cast<char>(v));
}

// BlockHandle::DecodeFrom (table/format.cc:60-69)
Status DecodeHandle(Slice* input, BlockHandle* h) {
  if (GetVarint64(input, &h->offset) && GetVarint64(input, &h->size)) return Status::OK();
  h->offset = 0;
  h->size = 0;
  return Status::Corruption("bad block handle");
}

int64_t ZigzagToI64(uint64_t n) { return static_cast<int64_t>(n >> 1) ^ -static_cast<int64_t>(n & 1); }

Status EngineError() { return Status::IOError(std::string("checksum engine: ") + tcrc_last_error()); }

// table/block_fetcher.cc:64-69
Status MismatchStatus(uint32_t computed_masked, uint32_t stored_masked, const std::string& file_name,
                      const BlockHandle& h) {
  return Status::Corruption("block checksum mismatch: expected " + std::to_string(tcrc_unmask(computed_masked)) +
                            ", got " + std::to_string(tcrc_unmask(stored_masked)) + "  in " + file_name + " offset " +
                            std::to_string(h.offset) + " size " + std::to_string(h.size));
}

// What BlockFetcher::ReadBlockContents reports when the file ends inside the block
// (table/block_fetcher.cc:230-236).  Returns OK when the block and its trailer fit.
Status BoundsStatus(const BlockHandle& h, uint64_t file_size, const std::string& file_name) {
  const uint64_t avail = h.offset < file_size ? file_size - h.offset : 0;
  if (h.size <= avail && kBlockTrailerSize <= avail - h.size) return Status::OK();
  // wrap-around of size + 5 is what the reference's size_t arithmetic would print too
  return Status::Corruption("truncated block read from " + file_name + " offset " + std::to_string(h.offset) +
                            ", expected " + std::to_string(h.size + kBlockTrailerSize) + " bytes, got " +
                            std::to_string(avail));
}

// contents + type byte must fit one engine span
const uint64_t kMaxBlockBytes = 1ull << 30;

// One block on the verification list.
struct Item {
  BlockHandle handle;
  Status status;         // bounds or checksum verdict
  bool checked = false;  // went through the engine
};

struct Run {
  BlockBatchChecker* checker;
  const RandomAccessFile* file;
  std::string file_name;
  uint64_t file_size;
  int checksum;  // footer value
  SstVerifyOptions options;
  SstVerifyStats* stats;
};

// unknown checksum type: table/block_fetcher.cc:58-62
Status UnknownChecksumStatus(const Run& r, const BlockHandle& h) {
  return Status::Corruption("unknown checksum type " + std::to_string(r.checksum) + " in " + r.file_name + " offset " +
                            std::to_string(h.offset) + " size " + std::to_string(h.size));
}

// Reads the listed blocks one by one into a packed image and checks them in one engine call.
// contents, when given, receives every block's bytes with the type byte (empty if unreadable).
Status CheckPacked(const Run& r, std::vector<Item>* items, std::vector<std::string>* contents) {
  const size_t n = items->size();
  if (contents) contents->assign(n, std::string());
  std::string image;
  std::vector<tcrc_desc> spans;
  std::vector<size_t> which;
  for (size_t i = 0; i < n; ++i) {
    Item& it = (*items)[i];
    it.status = BoundsStatus(it.handle, r.file_size, r.file_name);
    if (!it.status.ok()) continue;
    if (it.handle.size >= kMaxBlockBytes) {
      it.status = Status::NotSupported("block of 1 GiB or more");
      continue;
    }
    const size_t want = static_cast<size_t>(it.handle.size) + kBlockTrailerSize;
    const size_t at = image.size();
    image.resize(at + want);
    Slice got;
    Status s = r.file->Read(it.handle.offset, want, &got, &image[at]);
    if (!s.ok()) return s;
    if (got.size() != want) {
      image.resize(at);
      it.status = Status::Corruption("truncated block read from " + r.file_name + " offset " +
                                     std::to_string(it.handle.offset) + ", expected " + std::to_string(want) +
                                     " bytes, got " + std::to_string(got.size()));
      continue;
    }
    if (got.data() != &image[at]) memcpy(&image[at], got.data(), want);
    if (contents) (*contents)[i].assign(&image[at], static_cast<size_t>(it.handle.size) + 1);
    tcrc_desc d;
    d.offset = at;
    d.len = static_cast<uint32_t>(it.handle.size + 1);
    d.init = 0;
    spans.push_back(d);
    which.push_back(i);
  }
  if (spans.empty() || r.checksum == kNoChecksum) return Status::OK();
  if (r.checksum != kCRC32c) {
    for (size_t k : which) (*items)[k].status = UnknownChecksumStatus(r, (*items)[k].handle);
    return Status::OK();
  }
  std::vector<uint32_t> computed(spans.size());
  std::vector<uint8_t> ok(spans.size());
  Status s = r.checker->Check(image.data(), image.size(), spans.data(), spans.size(), computed.data(), ok.data());
  if (!s.ok()) return s;
  if (r.stats) ++r.stats->engine_calls;
  for (size_t k = 0; k < spans.size(); ++k) {
    Item& it = (*items)[which[k]];
    it.checked = true;
    if (r.stats) r.stats->bytes_covered += spans[k].len;
    if (ok[k]) continue;
    if (r.stats) ++r.stats->mismatches;
    const uint32_t stored = DecodeFixed32(image.data() + spans[k].offset + spans[k].len);
    it.status = MismatchStatus(computed[k], stored, r.file_name, it.handle);
  }
  return Status::OK();
}

// Data blocks: runs of handles that lie within extent_bytes of each other are read with one
// large read and checked with one engine call.  Two staging buffers alternate, so the file read
// of the next extent runs while the engine works on the current one.
struct Extent {
  uint64_t lo = 0;
  size_t bytes = 0;
  std::vector<tcrc_desc> spans;
  std::vector<size_t> which;
};

Status CheckExtents(const Run& r, std::vector<Item>* items) {
  const size_t n = items->size();
  // plan
  std::vector<Extent> extents;
  for (size_t i = 0; i < n;) {
    Item& first = (*items)[i];
    first.status = BoundsStatus(first.handle, r.file_size, r.file_name);
    if (first.status.ok() && first.handle.size >= kMaxBlockBytes)
      first.status = Status::NotSupported("block of 1 GiB or more");
    if (!first.status.ok()) {  // cannot be read at all: reported in order later
      ++i;
      continue;
    }
    Extent e;
    e.lo = first.handle.offset;
    uint64_t hi = e.lo + first.handle.size + kBlockTrailerSize;
    size_t j = i;
    for (; j < n; ++j) {
      Item& it = (*items)[j];
      if (j > i) {
        it.status = BoundsStatus(it.handle, r.file_size, r.file_name);
        if (it.status.ok() && it.handle.size >= kMaxBlockBytes)
          it.status = Status::NotSupported("block of 1 GiB or more");
        if (!it.status.ok()) continue;  // does not break the run
        const uint64_t end = it.handle.offset + it.handle.size + kBlockTrailerSize;
        if (it.handle.offset < e.lo || std::max(end, hi) - e.lo > r.options.extent_bytes) break;
        hi = std::max(hi, end);
      }
      tcrc_desc d;
      d.offset = it.handle.offset - e.lo;
      d.len = static_cast<uint32_t>(it.handle.size + 1);
      d.init = 0;
      e.spans.push_back(d);
      e.which.push_back(j);
    }
    e.bytes = static_cast<size_t>(hi - e.lo);
    extents.push_back(std::move(e));
    i = j;
  }
  if (r.checksum == kNoChecksum || extents.empty()) return Status::OK();
  if (r.checksum != kCRC32c) {
    for (const Extent& e : extents)
      for (size_t k : e.which) (*items)[k].status = UnknownChecksumStatus(r, (*items)[k].handle);
    return Status::OK();
  }
  size_t cap = 0;
  for (const Extent& e : extents) cap = std::max(cap, e.bytes);
  char* stage[2] = {r.checker->AllocStaging(cap), extents.size() > 1 ? r.checker->AllocStaging(cap) : nullptr};
  if (!stage[0] || (extents.size() > 1 && !stage[1])) {
    if (stage[0]) r.checker->FreeStaging(stage[0]);
    if (stage[1]) r.checker->FreeStaging(stage[1]);
    return Status::IOError("cannot allocate the staging buffers for file extents");
  }
  Status result, read_status[2];
  std::vector<uint32_t> computed;
  std::vector<uint8_t> ok;
  read_status[0] = ParallelRead(r.file, extents[0].lo, extents[0].bytes, stage[0], r.options.read_threads);
  for (size_t x = 0; x < extents.size() && result.ok(); ++x) {
    const Extent& e = extents[x];
    char* cur = stage[x & 1];
    std::thread next;
    if (x + 1 < extents.size()) {
      const Extent* ne = &extents[x + 1];
      Status* st = &read_status[(x + 1) & 1];
      char* dst = stage[(x + 1) & 1];
      next = std::thread([&r, ne, st, dst] { *st = ParallelRead(r.file, ne->lo, ne->bytes, dst, r.options.read_threads); });
    }
    result = read_status[x & 1];
    if (result.ok()) {
      computed.resize(e.spans.size());
      ok.resize(e.spans.size());
      result = r.checker->Check(cur, e.bytes, e.spans.data(), e.spans.size(), computed.data(), ok.data());
    }
    if (result.ok()) {
      if (r.stats) ++r.stats->engine_calls;
      for (size_t k = 0; k < e.spans.size(); ++k) {
        Item& it = (*items)[e.which[k]];
        it.checked = true;
        if (r.stats) r.stats->bytes_covered += e.spans[k].len;
        if (ok[k]) continue;
        if (r.stats) ++r.stats->mismatches;
        const uint32_t stored = DecodeFixed32(cur + e.spans[k].offset + e.spans[k].len);
        it.status = MismatchStatus(computed[k], stored, r.file_name, it.handle);
      }
    }
    if (next.joinable()) next.join();
  }
  r.checker->FreeStaging(stage[0]);
  if (stage[1]) r.checker->FreeStaging(stage[1]);
  return result;
}

Status FirstBad(const std::vector<Item>& items) {
  for (const Item& it : items)
    if (!it.status.ok()) return it.status;
  return Status::OK();
}

// Handles an index block lists, in order.  Parse errors end the list silently: the
// reference's loop stops when the iterator turns invalid and never looks at its status
// (table/block_based_table_reader.cc:2650-2667).
void IndexHandles(const std::string& block_with_type, bool delta_encoded, std::vector<BlockHandle>* out) {
  BlockCursor c(Slice(block_with_type.data(), block_with_type.size() - 1), delta_encoded);
  while (c.Next()) {
    if (delta_encoded) {
      out->push_back(c.handle());
    } else {
      BlockHandle h;
      Slice v = c.value();
      if (!DecodeHandle(&v, &h).ok()) break;  // IndexBlockIter::value() asserts; treat as the end
      out->push_back(h);
    }
  }
}

}  // namespace

// ---- footer --------------------------------------------------------------------------------------

Status Footer::DecodeFrom(Slice* input) {
  const char* magic_ptr = input->data() + input->size() - 8;
  uint64_t magic = static_cast<uint64_t>(DecodeFixed32(magic_ptr)) |
                   (static_cast<uint64_t>(DecodeFixed32(magic_ptr + 4)) << 32);
  const bool legacy = magic == kLegacyBlockBasedTableMagicNumber;
  if (legacy) magic = kBlockBasedTableMagicNumber;
  table_magic_number = magic;
  if (legacy) {
    input->remove_prefix(input->size() - kVersion0EncodedLength);
    version = 0;
    checksum = kCRC32c;
  } else {
    version = DecodeFixed32(magic_ptr - 4);
    if (input->size() < kNewVersionsEncodedLength) return Status::Corruption("input is too short to be an sstable");
    input->remove_prefix(input->size() - kNewVersionsEncodedLength);
    uint32_t chksum;
    const char* q = GetVarint32Ptr(input->data(), input->data() + input->size(), &chksum);
    if (!q) return Status::Corruption("bad checksum type");
    input->remove_prefix(q - input->data());
    checksum = static_cast<int>(static_cast<char>(chksum));  // the reference narrows to ChecksumType : char
  }
  Status result = DecodeHandle(input, &metaindex_handle);
  if (result.ok()) result = DecodeHandle(input, &index_handle);
  if (result.ok()) {
    const char* end = magic_ptr + 8;
    *input = Slice(end, input->data() + input->size() - end);
  }
  return result;
}

void Footer::EncodeTo(std::string* dst, bool legacy) const {
  const size_t original = dst->size();
  if (!legacy) dst->push_back(static_cast<char>(checksum));
  PutVarint64(dst, metaindex_handle.offset);
  PutVarint64(dst, metaindex_handle.size);
  PutVarint64(dst, index_handle.offset);
  PutVarint64(dst, index_handle.size);
  dst->resize(original + (legacy ? 40 : 41));  // padding
  const uint64_t magic = legacy ? kLegacyBlockBasedTableMagicNumber : kBlockBasedTableMagicNumber;
  char buf[4];
  if (!legacy) {
    EncodeFixed32(buf, version);
    dst->append(buf, 4);
  }
  EncodeFixed32(buf, static_cast<uint32_t>(magic & 0xffffffffu));
  dst->append(buf, 4);
  EncodeFixed32(buf, static_cast<uint32_t>(magic >> 32));
  dst->append(buf, 4);
}

Status ReadFooterFromFile(const RandomAccessFile* file, uint64_t file_size, const std::string& file_name,
                          Footer* footer, uint64_t enforce_table_magic_number) {
  // table/format.cc:235-282
  if (file_size < Footer::kMinEncodedLength)
    return Status::Corruption("file is too short (" + std::to_string(file_size) + " bytes) to be an sstable: " +
                              file_name);
  char space[Footer::kMaxEncodedLength];
  Slice input;
  const uint64_t read_offset = file_size > Footer::kMaxEncodedLength ? file_size - Footer::kMaxEncodedLength : 0;
  Status s = file->Read(read_offset, Footer::kMaxEncodedLength, &input, space);
  if (!s.ok()) return s;
  if (input.size() < Footer::kMinEncodedLength)
    return Status::Corruption("file is too short (" + std::to_string(file_size) + " bytes) to be an sstable" +
                              file_name);
  s = footer->DecodeFrom(&input);
  if (!s.ok()) return s;
  if (enforce_table_magic_number != 0 && enforce_table_magic_number != footer->table_magic_number)
    return Status::Corruption("Bad table magic number: expected " + std::to_string(enforce_table_magic_number) +
                              ", found " + std::to_string(footer->table_magic_number) + " in " + file_name);
  return Status::OK();
}

// ---- block entries -------------------------------------------------------------------------------

BlockCursor::BlockCursor(const Slice& contents, bool index_values_delta_encoded)
    : data_(contents.data()), delta_(index_values_delta_encoded) {
  // Block::Block / Block::NumRestarts (table/block.cc:736-833), binary-search layout only
  const size_t size = contents.size();
  if (size < 2 * sizeof(uint32_t)) {
    status_ = Status::Corruption("bad block contents");
    return;
  }
  uint32_t num_restarts = DecodeFixed32(data_ + size - 4);
  if (size <= 65536) {
    if (num_restarts & 0x80000000u) {  // a data block with the hash index appended: not an index or meta block
      status_ = Status::Corruption("bad block contents");
      return;
    }
  }
  const uint64_t tail = (1ull + num_restarts) * 4;
  if (tail > size) {
    status_ = Status::Corruption("bad block contents");
    return;
  }
  restarts_ = static_cast<uint32_t>(size - tail);
  if (num_restarts == 0) next_ = restarts_;  // empty block
}

bool BlockCursor::Next() {
  if (!status_.ok()) return false;
  const char* p = data_ + next_;
  const char* limit = data_ + restarts_;
  if (p >= limit) return false;
  uint32_t shared = 0, non_shared = 0, value_length = 0;
  p = GetVarint32Ptr(p, limit, &shared);
  if (p) p = GetVarint32Ptr(p, limit, &non_shared);
  if (p && !delta_) p = GetVarint32Ptr(p, limit, &value_length);
  if (!p || key_.size() < shared || static_cast<uint64_t>(limit - p) < static_cast<uint64_t>(non_shared) + value_length) {
    status_ = Status::Corruption("bad entry in block");
    next_ = restarts_;
    return false;
  }
  key_.resize(shared);
  key_.append(p, non_shared);
  p += non_shared;
  if (!delta_) {
    value_ = Slice(p, value_length);
    next_ = static_cast<uint32_t>(p + value_length - data_);
    return true;
  }
  // IndexBlockIter::DecodeCurrentValue (table/block.cc:586-606)
  const char* q;
  if (shared == 0) {
    uint64_t o = 0, s = 0;
    q = GetVarint64Ptr(p, limit, &o);
    if (q) q = GetVarint64Ptr(q, limit, &s);
    if (q) {
      handle_.offset = o;
      handle_.size = s;
    }
  } else {
    uint64_t u = 0;
    q = GetVarint64Ptr(p, limit, &u);
    if (q) {
      const uint64_t base = handle_.offset + handle_.size + kBlockTrailerSize;
      handle_.size = handle_.size + static_cast<uint64_t>(ZigzagToI64(u));
      handle_.offset = base;
    }
  }
  if (!q) {
    status_ = Status::Corruption("bad entry in block");
    next_ = restarts_;
    return false;
  }
  value_ = Slice(p, q - p);
  next_ = static_cast<uint32_t>(q - data_);
  return true;
}

// ---- the engine-backed checker -------------------------------------------------------------------

Status DeviceBlockBatchChecker::Check(const char* image, size_t image_bytes, const tcrc_desc* spans, size_t n,
                                      uint32_t* computed_masked, uint8_t* ok) {
  uint64_t counts[TCRC_CNT_WORDS];
  if (tcrc_host_batch_verify(ctx_, image, image_bytes, spans, n, nullptr, TCRC_F_EXPECT_AFTER, nullptr, computed_masked,
                             ok, counts) != TCRC_OK)
    return EngineError();
  return Status::OK();
}

char* DeviceBlockBatchChecker::AllocStaging(size_t bytes) { return static_cast<char*>(tcrc_host_alloc(ctx_, bytes)); }
void DeviceBlockBatchChecker::FreeStaging(char* p) { tcrc_host_free(ctx_, p); }

// ---- the driver ----------------------------------------------------------------------------------

Status VerifySstFileChecksum(BlockBatchChecker* checker, const RandomAccessFile* file, const std::string& file_name,
                             const SstVerifyOptions& options, SstVerifyStats* stats) {
  if (stats) *stats = SstVerifyStats();
  Run r{checker, file, file_name, file->Size(), kCRC32c, options, stats};
  if (r.options.extent_bytes < (1u << 20)) r.options.extent_bytes = 1u << 20;
  if (stats) stats->file_size = r.file_size;

  // Tables of the reference's other formats are not this driver's: say so before the block-based
  // open below would call their magic numbers corrupt (table/plain_table_builder.cc:53-54,
  // table/cuckoo_table_builder.cc:50, table/terark_zip_table.cc:64).
  if (r.file_size >= 8) {
    char tail[8];
    Slice m;
    Status ms = file->Read(r.file_size - 8, 8, &m, tail);
    if (ms.ok() && m.size() == 8) {
      const uint64_t magic = static_cast<uint64_t>(DecodeFixed32(m.data())) |
                             (static_cast<uint64_t>(DecodeFixed32(m.data() + 4)) << 32);
      if (magic == 0x8242229663bf9564ull || magic == 0x4f3418eb7a8f13b8ull || magic == 0x926789d0c5f17873ull ||
          magic == 0x1122334455667788ull)
        return Status::NotSupported("not a block-based table (plain, cuckoo or TerarkZip format): " + file_name);
    }
  }

  // BlockBasedTable::Open (table/block_based_table_reader.cc:814-823)
  Footer footer;
  Status s = ReadFooterFromFile(file, r.file_size, file_name, &footer, kBlockBasedTableMagicNumber);
  if (!s.ok()) return s;
  if (footer.version > 4)  // BlockBasedTableSupportedVersion (table/format.h:95-97)
    return Status::Corruption("Unknown Footer version. Maybe this file was created with newer version of RocksDB?");
  r.checksum = footer.checksum;
  if (r.checksum == kxxHash || r.checksum == kxxHash64)
    return Status::NotSupported("block checksum type is not CRC32c: " + file_name);

  // 1. the metaindex block (ReadMetaBlock, then VerifyChecksumInBlocks over its entries)
  std::vector<Item> meta(1);
  meta[0].handle = footer.metaindex_handle;
  std::vector<std::string> contents;
  s = CheckPacked(r, &meta, &contents);
  if (!s.ok()) return s;
  if (stats) stats->meta_blocks = 1;
  if (!meta[0].status.ok()) return meta[0].status;
  if (contents[0].back() != 0) return Status::NotSupported("compressed metaindex block in " + file_name);

  // 2. the blocks it names, in key order, and the index block in the same engine call
  std::vector<Item> named;
  size_t props_at = static_cast<size_t>(-1);
  {
    BlockCursor c(Slice(contents[0].data(), contents[0].size() - 1));
    while (c.Next()) {
      Item it;
      Slice v = c.value();
      (void)DecodeHandle(&v, &it.handle);  // a bad handle is checked as (0, 0), like the reference (:2681)
      if (c.key() == "rocksdb.properties" || (c.key() == "rocksdb.stats" && props_at == static_cast<size_t>(-1)))
        props_at = named.size();
      named.push_back(it);
    }
  }
  const size_t n_named = named.size();
  Item index_item;
  index_item.handle = footer.index_handle;
  named.push_back(index_item);
  s = CheckPacked(r, &named, &contents);
  if (!s.ok()) return s;
  if (stats) {
    stats->meta_blocks += n_named;
    stats->index_blocks = 1;
  }
  for (size_t i = 0; i < n_named; ++i)
    if (!named[i].status.ok()) return named[i].status;

  // index layout from the properties block (table/block_based_table_reader.cc:2729-2744,2767-2825)
  int index_type = r.options.index_type_hint;
  bool delta_encoded = false;
  if (props_at != static_cast<size_t>(-1) && !contents[props_at].empty() && contents[props_at].back() == 0) {
    BlockCursor c(Slice(contents[props_at].data(), contents[props_at].size() - 1));
    while (c.Next()) {
      Slice v = c.value();
      if (c.key() == "rocksdb.block.based.table.index.type" && v.size() >= 4) {
        index_type = static_cast<int>(DecodeFixed32(v.data()));
      } else if (c.key() == "rocksdb.index.value.is.delta.encoded") {
        uint64_t x = 0;
        if (GetVarint64(&v, &x)) delta_encoded = x != 0;
      }
    }
  }
  if (index_type < 0 || index_type > 2)
    return Status::InvalidArgument("Unrecognized index type: " + std::to_string(index_type));

  // 3. the index block
  const Item& idx = named[n_named];
  if (!idx.status.ok()) return idx.status;
  const std::string& index_block = contents[n_named];
  if (index_block.back() != 0)
    return Status::NotSupported("compressed index block (type " + std::to_string(static_cast<int>(index_block.back())) +
                                ") in " + file_name);
  {
    BlockCursor probe(Slice(index_block.data(), index_block.size() - 1), delta_encoded);
    if (!probe.status().ok()) return probe.status();  // NewIndexIterator's status (:2639-2642)
  }
  std::vector<BlockHandle> handles;
  IndexHandles(index_block, delta_encoded, &handles);

  // 4. data blocks (through the partitions of a two-level index)
  std::vector<Item> data;
  Status after_data;  // a bad partition block is reported after the data blocks of the partitions before it
  if (index_type == 2) {
    std::vector<Item> parts(handles.size());
    for (size_t i = 0; i < handles.size(); ++i) parts[i].handle = handles[i];
    std::vector<std::string> part_contents;
    s = CheckPacked(r, &parts, &part_contents);
    if (!s.ok()) return s;
    if (stats) stats->index_blocks += parts.size();
    for (size_t i = 0; i < parts.size(); ++i) {
      if (!parts[i].status.ok()) {
        after_data = parts[i].status;
        break;
      }
      if (part_contents[i].back() != 0) {
        after_data = Status::NotSupported("compressed index partition in " + file_name);
        break;
      }
      std::vector<BlockHandle> hs;
      IndexHandles(part_contents[i], delta_encoded, &hs);
      for (const BlockHandle& h : hs) {
        Item it;
        it.handle = h;
        data.push_back(it);
      }
    }
  } else {
    data.resize(handles.size());
    for (size_t i = 0; i < handles.size(); ++i) data[i].handle = handles[i];
  }
  s = CheckExtents(r, &data);
  if (!s.ok()) return s;
  if (stats) stats->data_blocks = data.size();
  s = FirstBad(data);
  if (!s.ok()) return s;
  return after_data;
}

Status VerifySstFileChecksum(tcrc_ctx* ctx, const RandomAccessFile* file, const std::string& file_name,
                             const SstVerifyOptions& options, SstVerifyStats* stats) {
  DeviceBlockBatchChecker checker(ctx);
  return VerifySstFileChecksum(&checker, file, file_name, options, stats);
}

Status VerifySstFileChecksum(tcrc_ctx* ctx, const std::string& path, const SstVerifyOptions& options,
                             SstVerifyStats* stats) {
  std::unique_ptr<RandomAccessFile> file;
  Status s = NewPosixRandomAccessFile(path, &file);
  if (!s.ok()) return s;
  return VerifySstFileChecksum(ctx, file.get(), path, options, stats);
}

}  // namespace terarkdb_amd
