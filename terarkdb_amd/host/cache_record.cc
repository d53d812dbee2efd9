#include "cache_record.h"

#include <string.h>

#include <atomic>

namespace terarkdb_amd {

namespace {

Status EngineError() { return Status::IOError(std::string("checksum engine: ") + tcrc_last_error()); }

// crc32c of a record's first eight bytes with the crc field zeroed: the same for every record, so a
// stored record is checked as Extend(this, bytes 8 .. end) without touching its crc field.  Computed by
// the engine the first time it is needed.
Status HeaderPrefixCrc(tcrc_ctx* ctx, uint32_t* out) {
  static std::atomic<uint64_t> cached{0};  // bit 32 set once the low word is valid
  uint64_t c = cached.load(std::memory_order_acquire);
  if (!(c >> 32)) {
    CacheRecordHeader h = {kCacheRecordMagic, 0, 0, 0};
    uint32_t v = 0;
    if (tcrc_value(ctx, &h, 8, &v) != TCRC_OK) return EngineError();
    c = (1ull << 32) | v;
    cached.store(c, std::memory_order_release);
  }
  *out = static_cast<uint32_t>(c);
  return Status::OK();
}

}  // namespace

Status SerializeCacheRecords(tcrc_ctx* ctx, const std::vector<Slice>& keys, const std::vector<Slice>& vals,
                             std::string* out, std::vector<uint64_t>* offsets) {
  if (keys.size() != vals.size()) return Status::InvalidArgument("keys and values differ in number");
  const size_t n = keys.size();
  if (offsets) offsets->clear();
  if (n == 0) return Status::OK();
  const size_t start = out->size();
  size_t total = 0;
  for (size_t i = 0; i < n; ++i) {
    if (keys[i].size() > 0xffffffffull || vals[i].size() > 0xffffffffull ||
        sizeof(CacheRecordHeader) + keys[i].size() + vals[i].size() > (1ull << 30))
      return Status::InvalidArgument("cache record too large");
    total += sizeof(CacheRecordHeader) + keys[i].size() + vals[i].size();
  }
  out->resize(start + total);
  char* base = &(*out)[start];
  std::vector<tcrc_desc> spans(n);
  size_t pos = 0;
  for (size_t i = 0; i < n; ++i) {
    CacheRecordHeader h = {kCacheRecordMagic, 0, static_cast<uint32_t>(keys[i].size()),
                           static_cast<uint32_t>(vals[i].size())};
    memcpy(base + pos, &h, sizeof(h));
    if (keys[i].size()) memcpy(base + pos + sizeof(h), keys[i].data(), keys[i].size());
    if (vals[i].size()) memcpy(base + pos + sizeof(h) + keys[i].size(), vals[i].data(), vals[i].size());
    spans[i].offset = pos;  // the crc field is zero at this point, as in ComputeCRC's copy of the header
    spans[i].len = static_cast<uint32_t>(sizeof(h) + keys[i].size() + vals[i].size());
    spans[i].init = 0;
    if (offsets) offsets->push_back(start + pos);
    pos += spans[i].len;
  }
  std::vector<uint32_t> crc(n);
  if (tcrc_host_batch_stamp(ctx, base, total, spans.data(), n, nullptr, 0, crc.data()) != TCRC_OK) {
    out->resize(start);
    return EngineError();
  }
  for (size_t i = 0; i < n; ++i) memcpy(base + spans[i].offset + offsetof(CacheRecordHeader, crc), &crc[i], 4);
  return Status::OK();
}

Status DeserializeCacheRecords(tcrc_ctx* ctx, const Slice& image, const std::vector<std::pair<uint64_t, uint32_t>>& lbas,
                               std::vector<CacheRecordView>* records) {
  const size_t n = lbas.size();
  records->assign(n, CacheRecordView());
  std::vector<tcrc_desc> spans;
  std::vector<uint32_t> expected;
  std::vector<size_t> which;
  uint32_t prefix = 0;
  for (size_t i = 0; i < n; ++i) {
    const uint64_t off = lbas[i].first;
    const uint64_t size = lbas[i].second;
    if (off > image.size() || size > image.size() - off) return Status::InvalidArgument("record outside the image");
    if (size < sizeof(CacheRecordHeader)) continue;  // :164-166
    CacheRecordHeader h;
    memcpy(&h, image.data() + off, sizeof(h));
    // :170-173 (the sum is formed in 32 bits there; a wrapped sum cannot describe a record that fits)
    if (static_cast<uint64_t>(h.key_size) + h.val_size + sizeof(h) != size) continue;
    CacheRecordView& v = (*records)[i];
    v.key = Slice(image.data() + off + sizeof(h), h.key_size);
    v.val = Slice(v.key.data() + h.key_size, h.val_size);
    if (h.magic != kCacheRecordMagic) continue;  // :178,191
    if (size > (1ull << 30)) return Status::NotSupported("cache record above 1 GiB");
    if (spans.empty()) {
      Status s = HeaderPrefixCrc(ctx, &prefix);
      if (!s.ok()) return s;
    }
    tcrc_desc d;
    d.offset = off + 8;
    d.len = static_cast<uint32_t>(size - 8);
    d.init = prefix;
    spans.push_back(d);
    expected.push_back(tcrc_mask(h.crc));  // the engine compares masked values
    which.push_back(i);
  }
  if (spans.empty()) return Status::OK();
  std::vector<uint8_t> ok(spans.size());
  uint64_t counts[TCRC_CNT_WORDS];
  if (tcrc_host_batch_verify(ctx, image.data(), image.size(), spans.data(), spans.size(), nullptr, 0, expected.data(),
                             nullptr, ok.data(), counts) != TCRC_OK)
    return EngineError();
  for (size_t k = 0; k < spans.size(); ++k) (*records)[which[k]].ok = ok[k] != 0;
  return Status::OK();
}

}  // namespace terarkdb_amd
