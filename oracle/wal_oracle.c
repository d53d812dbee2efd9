/*
 * TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT.
 *
 * CPU restatement of the reference's write-ahead-log record framing, used only
 * as the checker in tests/ and as the timed CPU baseline.  Nothing under
 * terarkdb_amd/ links, loads or calls this file.
 *
 * What it restates (paths relative to the reference tree):
 *   db/log_format.h:19-44     record types, 32 KiB blocks, 7 / 11 byte headers
 *   db/log_writer.cc:23-34    per-type CRC seeds
 *   db/log_writer.cc:45-100   AddRecord: fragmentation and block trailers
 *   db/log_writer.cc:104-151  EmitPhysicalRecord: header layout and masked CRC
 *   db/log_reader.cc:54-198   ReadRecord: reassembly and recovery-mode policy
 *   db/log_reader.cc:202-264  UnmarkEOF
 *   db/log_reader.cc:276-312  ReadMore
 *   db/log_reader.cc:314-406  ReadPhysicalRecord: parsing, bounds, CRC check
 * plus the in-memory file fakes the reference's own test drives them with
 * (db/log_test.cc:49-126, util/testutil.h:201-252,370-400).
 *
 * Parity status: pinned.  tests/test_oracle_wal.py replays the scenarios of
 * db/log_test.cc and checks the byte strings and per-record CRCs captured from
 * a build of the reference (tests/golden/reference_captured.json).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

uint32_t oracle_crc32c_extend(uint32_t init, const void* data, size_t n);
uint32_t oracle_crc32c_extend_hw(uint32_t init, const void* data, size_t n);
uint32_t oracle_crc32c_mask(uint32_t crc);
uint32_t oracle_crc32c_unmask(uint32_t masked);

enum { kZeroType = 0, kFullType = 1, kFirstType = 2, kMiddleType = 3, kLastType = 4,
       kRecyclableFullType = 5, kRecyclableFirstType = 6, kRecyclableMiddleType = 7, kRecyclableLastType = 8,
       kMaxRecordType = 8 };
enum { kBlockSize = 32768, kHeaderSize = 7, kRecyclableHeaderSize = 11 };
enum { kEof = kMaxRecordType + 1, kBadRecord = kMaxRecordType + 2, kBadHeader = kMaxRecordType + 3,
       kOldRecord = kMaxRecordType + 4, kBadRecordLen = kMaxRecordType + 5, kBadRecordChecksum = kMaxRecordType + 6 };
/* include/rocksdb/options.h WALRecoveryMode */
enum { kTolerateCorruptedTailRecords = 0, kAbsoluteConsistency = 1, kPointInTimeRecovery = 2,
       kSkipAnyCorruptedRecords = 3 };

/* ---- growable byte string ------------------------------------------------- */
typedef struct {
  uint8_t* p;
  size_t n, cap;
} bytes_t;

static void bytes_reserve(bytes_t* b, size_t want) {
  if (want <= b->cap) return;
  size_t c = b->cap ? b->cap : 256;
  while (c < want) c *= 2;
  b->p = (uint8_t*)realloc(b->p, c);
  b->cap = c;
}
static void bytes_append(bytes_t* b, const void* s, size_t n) {
  bytes_reserve(b, b->n + n + 1);
  if (n) memcpy(b->p + b->n, s, n);
  b->n += n;
}
static void bytes_assign(bytes_t* b, const void* s, size_t n) {
  b->n = 0;
  bytes_append(b, s, n);
}

/* ---- sink: the file image a writer produces ------------------------------- */
typedef struct {
  bytes_t contents;
  size_t write_pos; /* == contents.n in append mode; < n while overwriting a recycled file */
} oracle_sink;

void* oracle_sink_new(void) { return calloc(1, sizeof(oracle_sink)); }
void oracle_sink_free(void* s) {
  if (!s) return;
  free(((oracle_sink*)s)->contents.p);
  free(s);
}
uint8_t* oracle_sink_data(void* s) { return ((oracle_sink*)s)->contents.p; }
size_t oracle_sink_size(void* s) { return ((oracle_sink*)s)->contents.n; }
void oracle_sink_drop(void* s, size_t bytes) { /* StringSink::Drop */
  oracle_sink* k = (oracle_sink*)s;
  k->contents.n -= bytes;
  if (k->write_pos > k->contents.n) k->write_pos = k->contents.n;
}
void oracle_sink_append_raw(void* s, const void* data, size_t n) {
  oracle_sink* k = (oracle_sink*)s;
  bytes_append(&k->contents, data, n);
  k->write_pos = k->contents.n;
}
void oracle_sink_rewind_for_overwrite(void* s) { ((oracle_sink*)s)->write_pos = 0; } /* OverwritingStringSink */
static void sink_write(oracle_sink* k, const void* s, size_t n) {
  if (k->write_pos + n > k->contents.n) {
    bytes_reserve(&k->contents, k->write_pos + n + 1);
    k->contents.n = k->write_pos + n;
  }
  if (n) memcpy(k->contents.p + k->write_pos, s, n);
  k->write_pos += n;
}

/* ---- writer ---------------------------------------------------------------- */
typedef struct {
  oracle_sink* dest;
  size_t block_offset;
  uint64_t log_number;
  int recycle;
  int use_hw;
  uint32_t type_crc[kMaxRecordType + 1];
  uint64_t n_physical, crc_bytes;
} oracle_wal_writer;

void* oracle_wal_writer_new(void* sink, uint64_t log_number, int recycle_log_files) {
  oracle_wal_writer* w = (oracle_wal_writer*)calloc(1, sizeof(*w));
  w->dest = (oracle_sink*)sink;
  w->log_number = log_number;
  w->recycle = recycle_log_files != 0;
  for (int i = 0; i <= kMaxRecordType; i++) { /* db/log_writer.cc:30-33 */
    char t = (char)i;
    w->type_crc[i] = oracle_crc32c_extend(0, &t, 1);
  }
  return w;
}
void oracle_wal_writer_free(void* w) { free(w); }
void oracle_wal_writer_use_hw(void* w, int on) { ((oracle_wal_writer*)w)->use_hw = on; }
uint32_t oracle_wal_writer_type_crc(void* w, int t) { return ((oracle_wal_writer*)w)->type_crc[t]; }
uint64_t oracle_wal_writer_physical_records(void* w) { return ((oracle_wal_writer*)w)->n_physical; }
uint64_t oracle_wal_writer_crc_bytes(void* w) { return ((oracle_wal_writer*)w)->crc_bytes; }

static void put_le32(uint8_t* p, uint32_t v) {
  p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
static uint32_t get_le32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* db/log_writer.cc:104-151 */
static void emit_physical(oracle_wal_writer* w, int t, const uint8_t* ptr, size_t n) {
  uint8_t buf[kRecyclableHeaderSize];
  size_t header_size;
  buf[4] = (uint8_t)(n & 0xff);
  buf[5] = (uint8_t)(n >> 8);
  buf[6] = (uint8_t)t;
  uint32_t crc = w->type_crc[t];
  if (t < kRecyclableFullType) {
    header_size = kHeaderSize;
  } else {
    header_size = kRecyclableHeaderSize;
    put_le32(buf + 7, (uint32_t)w->log_number);
    crc = oracle_crc32c_extend(crc, buf + 7, 4);
  }
  crc = w->use_hw ? oracle_crc32c_extend_hw(crc, ptr, n) : oracle_crc32c_extend(crc, ptr, n);
  crc = oracle_crc32c_mask(crc);
  put_le32(buf, crc);
  sink_write(w->dest, buf, header_size);
  sink_write(w->dest, ptr, n);
  w->block_offset += header_size + n;
  w->n_physical++;
  w->crc_bytes += header_size - 6 + n;
}

/* db/log_writer.cc:45-100 */
int oracle_wal_writer_add_record(void* wp, const void* data, size_t left) {
  oracle_wal_writer* w = (oracle_wal_writer*)wp;
  const uint8_t* ptr = (const uint8_t*)data;
  const int header_size = w->recycle ? kRecyclableHeaderSize : kHeaderSize;
  int begin = 1;
  do {
    const int64_t leftover = (int64_t)kBlockSize - (int64_t)w->block_offset;
    if (leftover < header_size) {
      if (leftover > 0) {
        static const uint8_t zeros[kRecyclableHeaderSize] = {0};
        sink_write(w->dest, zeros, (size_t)leftover);
      }
      w->block_offset = 0;
    }
    const size_t avail = kBlockSize - w->block_offset - header_size;
    const size_t fragment_length = (left < avail) ? left : avail;
    const int end = (left == fragment_length);
    int type;
    if (begin && end) type = w->recycle ? kRecyclableFullType : kFullType;
    else if (begin) type = w->recycle ? kRecyclableFirstType : kFirstType;
    else if (end) type = w->recycle ? kRecyclableLastType : kLastType;
    else type = w->recycle ? kRecyclableMiddleType : kMiddleType;
    emit_physical(w, type, ptr, fragment_length);
    ptr += fragment_length;
    left -= fragment_length;
    begin = 0;
  } while (left > 0);
  return 0;
}

/* ---- source: what the reader reads from (db/log_test.cc:49-114) ------------ */
typedef struct {
  oracle_sink* sink; /* unread bytes are sink contents [pos, size) */
  size_t pos;
  int force_error;
  size_t force_error_position;
  int force_eof;
  size_t force_eof_position;
  int returned_partial;
  int read_after_partial; /* the reference's test asserts this never happens */
} oracle_source;

void* oracle_source_new(void* sink) {
  oracle_source* s = (oracle_source*)calloc(1, sizeof(*s));
  s->sink = (oracle_sink*)sink;
  return s;
}
void oracle_source_free(void* s) { free(s); }
void oracle_source_force_error(void* s, size_t position) {
  ((oracle_source*)s)->force_error = 1;
  ((oracle_source*)s)->force_error_position = position;
}
void oracle_source_force_eof(void* s, size_t position) {
  ((oracle_source*)s)->force_eof = 1;
  ((oracle_source*)s)->force_eof_position = position;
}
void oracle_source_clear_partial(void* s) { ((oracle_source*)s)->returned_partial = 0; }
int oracle_source_read_after_partial(void* s) { return ((oracle_source*)s)->read_after_partial; }

/* returns 0 ok, 1 = Corruption("read error"); *got bytes copied to scratch */
static int source_read(oracle_source* s, size_t n, uint8_t* scratch, size_t* got) {
  if (s->returned_partial) s->read_after_partial = 1;
  size_t avail = s->sink->contents.n > s->pos ? s->sink->contents.n - s->pos : 0;
  const uint8_t* cur = s->sink->contents.p + s->pos;
  if (s->force_error) {
    if (s->force_error_position >= n) {
      s->force_error_position -= n;
    } else {
      /* the reference hands back the bytes before the error position (unchecked
       * against the remaining size, as in the test fake) */
      size_t k = s->force_error_position;
      if (k > avail) k = avail;
      if (k) memcpy(scratch, cur, k);
      *got = k;
      s->pos += k;
      s->force_error = 0;
      s->returned_partial = 1;
      return 1;
    }
  }
  if (avail < n) {
    n = avail;
    s->returned_partial = 1;
  }
  if (s->force_eof) {
    if (s->force_eof_position >= n) {
      s->force_eof_position -= n;
    } else {
      s->force_eof = 0;
      n = s->force_eof_position;
      s->returned_partial = 1;
    }
  }
  if (n) memcpy(scratch, cur, n);
  *got = n;
  s->pos += n;
  return 0;
}

/* ---- reader ----------------------------------------------------------------- */
typedef struct {
  oracle_source* file;
  int checksum;
  uint8_t* backing_store;
  const uint8_t* buf; /* buffer_ */
  size_t buf_n;
  int eof, read_error;
  size_t eof_offset;
  uint64_t last_record_offset, end_of_buffer_offset, log_number;
  int recycled, retry_after_eof;
  /* reporter (db/log_test.cc:116-126) */
  size_t dropped_bytes;
  bytes_t message;
  bytes_t scratch;
  uint64_t n_physical_ok, crc_bytes;
  int use_hw;
} oracle_wal_reader;

void* oracle_wal_reader_new(void* source, int checksum, uint64_t log_num, int retry_after_eof) {
  oracle_wal_reader* r = (oracle_wal_reader*)calloc(1, sizeof(*r));
  r->file = (oracle_source*)source;
  r->checksum = checksum;
  r->backing_store = (uint8_t*)malloc(kBlockSize);
  r->log_number = log_num;
  r->retry_after_eof = retry_after_eof;
  bytes_append(&r->message, "", 0);
  r->message.p[0] = 0;
  return r;
}
void oracle_wal_reader_free(void* rp) {
  oracle_wal_reader* r = (oracle_wal_reader*)rp;
  if (!r) return;
  free(r->backing_store);
  free(r->message.p);
  free(r->scratch.p);
  free(r);
}
void oracle_wal_reader_use_hw(void* r, int on) { ((oracle_wal_reader*)r)->use_hw = on; }
size_t oracle_wal_reader_dropped_bytes(void* r) { return ((oracle_wal_reader*)r)->dropped_bytes; }
const char* oracle_wal_reader_report(void* r) { return (const char*)((oracle_wal_reader*)r)->message.p; }
int oracle_wal_reader_is_eof(void* r) { return ((oracle_wal_reader*)r)->eof; }
uint64_t oracle_wal_reader_last_record_offset(void* r) { return ((oracle_wal_reader*)r)->last_record_offset; }
uint64_t oracle_wal_reader_physical_records(void* r) { return ((oracle_wal_reader*)r)->n_physical_ok; }
uint64_t oracle_wal_reader_crc_bytes(void* r) { return ((oracle_wal_reader*)r)->crc_bytes; }

static void report_drop(oracle_wal_reader* r, size_t bytes, const char* status_string) {
  r->dropped_bytes += bytes;
  bytes_append(&r->message, status_string, strlen(status_string));
  r->message.p[r->message.n] = 0;
}
static void report_corruption(oracle_wal_reader* r, size_t bytes, const char* reason) {
  char tmp[128];
  snprintf(tmp, sizeof(tmp), "Corruption: %s", reason);
  report_drop(r, bytes, tmp);
}

/* db/log_reader.cc:202-264 */
void oracle_wal_reader_unmark_eof(void* rp) {
  oracle_wal_reader* r = (oracle_wal_reader*)rp;
  if (r->read_error) return;
  r->eof = 0;
  if (!r->retry_after_eof && r->eof_offset == 0) return;
  size_t consumed_bytes = r->eof_offset - r->buf_n;
  size_t remaining = kBlockSize - r->eof_offset;
  if (r->buf != r->backing_store + consumed_bytes) {
    memmove(r->backing_store + consumed_bytes, r->buf, r->buf_n);
  }
  size_t added = 0;
  int err = source_read(r->file, remaining, r->backing_store + r->eof_offset, &added);
  r->end_of_buffer_offset += added;
  if (err) {
    if (added > 0) report_drop(r, added, "Corruption: read error");
    r->read_error = 1;
    return;
  }
  r->buf = r->backing_store + consumed_bytes;
  r->buf_n = r->eof_offset + added - consumed_bytes;
  if (added < remaining) {
    r->eof = 1;
    r->eof_offset += added;
  } else {
    r->eof_offset = 0;
  }
}

/* db/log_reader.cc:276-312 */
static int read_more(oracle_wal_reader* r, size_t* drop_size, int* error) {
  if (!r->eof && !r->read_error) {
    r->buf_n = 0;
    size_t got = 0;
    int err = source_read(r->file, kBlockSize, r->backing_store, &got);
    r->buf = r->backing_store;
    r->buf_n = got;
    r->end_of_buffer_offset += got;
    if (err) {
      r->buf_n = 0;
      report_drop(r, kBlockSize, "Corruption: read error");
      r->read_error = 1;
      *error = kEof;
      return 0;
    } else if (got < (size_t)kBlockSize) {
      r->eof = 1;
      r->eof_offset = got;
    }
    return 1;
  } else if (r->retry_after_eof && !r->read_error) {
    oracle_wal_reader_unmark_eof(r);
    return !r->read_error;
  } else {
    if (r->buf_n) {
      *drop_size = r->buf_n;
      r->buf_n = 0;
      *error = kBadHeader;
      return 0;
    }
    r->buf_n = 0;
    *error = kEof;
    return 0;
  }
}

/* db/log_reader.cc:314-406 */
static unsigned read_physical(oracle_wal_reader* r, const uint8_t** frag, size_t* frag_n, size_t* drop_size) {
  for (;;) {
    if (r->buf_n < (size_t)kHeaderSize) {
      int e = kEof;
      if (!read_more(r, drop_size, &e)) return (unsigned)e;
      continue;
    }
    const uint8_t* header = r->buf;
    const uint32_t a = header[4], b = header[5];
    const unsigned type = (unsigned)(int)(signed char)header[6]; /* char -> unsigned, as the reference */
    const uint32_t length = a | (b << 8);
    int header_size = kHeaderSize;
    if (type >= kRecyclableFullType && type <= kRecyclableLastType) {
      if (r->end_of_buffer_offset - r->buf_n == 0) r->recycled = 1;
      header_size = kRecyclableHeaderSize;
      if (r->buf_n < (size_t)kRecyclableHeaderSize) {
        int e = kEof;
        if (!read_more(r, drop_size, &e)) return (unsigned)e;
        continue;
      }
      const uint32_t log_num = get_le32(header + 7);
      if ((uint64_t)log_num != r->log_number) return kOldRecord; /* 64-bit compare, as the reference */
    }
    if ((size_t)header_size + length > r->buf_n) {
      if (!r->retry_after_eof) {
        *drop_size = r->buf_n;
        r->buf_n = 0;
        if (!r->eof) return kBadRecordLen;
        if (*drop_size) return kBadHeader;
      } else {
        int e = kEof;
        if (!read_more(r, drop_size, &e)) return (unsigned)e;
        continue;
      }
      return kEof;
    }
    if (type == kZeroType && length == 0) {
      r->buf_n = 0;
      return kBadRecord;
    }
    if (r->checksum) {
      uint32_t expected = oracle_crc32c_unmask(get_le32(header));
      size_t n = length + header_size - 6;
      uint32_t actual = r->use_hw ? oracle_crc32c_extend_hw(0, header + 6, n) : oracle_crc32c_extend(0, header + 6, n);
      if (actual != expected) {
        *drop_size = r->buf_n;
        r->buf_n = 0;
        return kBadRecordChecksum;
      }
      r->n_physical_ok++;
      r->crc_bytes += n;
    }
    r->buf += header_size + length;
    r->buf_n -= header_size + length;
    *frag = header + header_size;
    *frag_n = length;
    return type;
  }
}

/* db/log_reader.cc:54-198.  Returns 1 and the record, or 0 at end of input. */
int oracle_wal_reader_read_record(void* rp, int wal_recovery_mode, const uint8_t** rec, size_t* rec_n) {
  oracle_wal_reader* r = (oracle_wal_reader*)rp;
  bytes_t* scratch = &r->scratch;
  scratch->n = 0;
  *rec = NULL;
  *rec_n = 0;
  int in_fragmented_record = 0;
  uint64_t prospective_record_offset = 0;
  const uint8_t* frag = NULL;
  size_t frag_n = 0;
  for (;;) {
    uint64_t physical_record_offset = r->end_of_buffer_offset - r->buf_n;
    size_t drop_size = 0;
    frag = NULL;
    frag_n = 0;
    const unsigned record_type = read_physical(r, &frag, &frag_n, &drop_size);
    switch (record_type) {
      case kFullType:
      case kRecyclableFullType:
        if (in_fragmented_record && scratch->n) report_corruption(r, scratch->n, "partial record without end(1)");
        prospective_record_offset = physical_record_offset;
        scratch->n = 0;
        *rec = frag;
        *rec_n = frag_n;
        r->last_record_offset = prospective_record_offset;
        return 1;
      case kFirstType:
      case kRecyclableFirstType:
        if (in_fragmented_record && scratch->n) report_corruption(r, scratch->n, "partial record without end(2)");
        prospective_record_offset = physical_record_offset;
        bytes_assign(scratch, frag, frag_n);
        in_fragmented_record = 1;
        break;
      case kMiddleType:
      case kRecyclableMiddleType:
        if (!in_fragmented_record) report_corruption(r, frag_n, "missing start of fragmented record(1)");
        else bytes_append(scratch, frag, frag_n);
        break;
      case kLastType:
      case kRecyclableLastType:
        if (!in_fragmented_record) {
          report_corruption(r, frag_n, "missing start of fragmented record(2)");
        } else {
          bytes_append(scratch, frag, frag_n);
          *rec = scratch->p;
          *rec_n = scratch->n;
          r->last_record_offset = prospective_record_offset;
          return 1;
        }
        break;
      case kBadHeader:
        if (wal_recovery_mode == kAbsoluteConsistency) report_corruption(r, drop_size, "truncated header");
        /* fall through */
      case kEof:
        if (in_fragmented_record) {
          if (wal_recovery_mode == kAbsoluteConsistency)
            report_corruption(r, scratch->n, "error reading trailing data");
          scratch->n = 0;
        }
        return 0;
      case kOldRecord:
        if (wal_recovery_mode != kSkipAnyCorruptedRecords) {
          if (in_fragmented_record) {
            if (wal_recovery_mode == kAbsoluteConsistency)
              report_corruption(r, scratch->n, "error reading trailing data");
            scratch->n = 0;
          }
          return 0;
        }
        /* fall through */
      case kBadRecord:
        if (in_fragmented_record) {
          report_corruption(r, scratch->n, "error in middle of record");
          in_fragmented_record = 0;
          scratch->n = 0;
        }
        break;
      case kBadRecordLen:
      case kBadRecordChecksum:
        if (r->recycled && wal_recovery_mode == kTolerateCorruptedTailRecords) {
          scratch->n = 0;
          return 0;
        }
        if (record_type == kBadRecordLen) report_corruption(r, drop_size, "bad record length");
        else report_corruption(r, drop_size, "checksum mismatch");
        if (in_fragmented_record) {
          report_corruption(r, scratch->n, "error in middle of record");
          in_fragmented_record = 0;
          scratch->n = 0;
        }
        break;
      default: {
        char buf[40];
        snprintf(buf, sizeof(buf), "unknown record type %u", record_type);
        report_corruption(r, frag_n + (in_fragmented_record ? scratch->n : 0), buf);
        in_fragmented_record = 0;
        scratch->n = 0;
        break;
      }
    }
  }
}

/* ---- helpers shared by tests and the bench --------------------------------- */

/* Scan a complete in-memory log image the way ReadPhysicalRecord walks it
 * (checksum on, no retry) and list every physical record it accepts:
 * out4[4*i..] = {file offset of header, type, payload length, stored masked crc}.
 * Returns the number of records (stops listing at cap). */
size_t oracle_wal_list_physical(const uint8_t* img, size_t n, uint64_t log_number, uint64_t* out4, size_t cap) {
  size_t count = 0;
  for (size_t blk = 0; blk < n; blk += kBlockSize) {
    size_t end = blk + kBlockSize < n ? blk + kBlockSize : n;
    size_t p = blk;
    while (end - p >= (size_t)kHeaderSize) {
      const uint8_t* h = img + p;
      unsigned type = (unsigned)(int)(signed char)h[6];
      uint32_t len = h[4] | ((uint32_t)h[5] << 8);
      size_t hs = kHeaderSize;
      if (type >= kRecyclableFullType && type <= kRecyclableLastType) {
        hs = kRecyclableHeaderSize;
        if (end - p < hs) break;
        if ((uint64_t)get_le32(h + 7) != log_number) break;
      }
      if (hs + len > end - p) break;
      if (type == kZeroType && len == 0) break;
      uint32_t actual = oracle_crc32c_extend(0, h + 6, len + hs - 6);
      if (actual != oracle_crc32c_unmask(get_le32(h))) break;
      if (count < cap) {
        out4[4 * count + 0] = p;
        out4[4 * count + 1] = type;
        out4[4 * count + 2] = len;
        out4[4 * count + 3] = get_le32(h);
      }
      count++;
      p += hs + len;
    }
  }
  return count;
}

/* util/random.h:22-75 (the generator db/log_test.cc seeds with 301) */
typedef struct { uint32_t seed; } oracle_random;
void oracle_random_seed(oracle_random* r, uint32_t s) {
  r->seed = (s & 2147483647u) != 0 ? (s & 2147483647u) : 1;
}
uint32_t oracle_random_next(oracle_random* r) {
  uint64_t product = (uint64_t)r->seed * 16807u;
  r->seed = (uint32_t)((product >> 31) + (product & 2147483647u));
  if (r->seed > 2147483647u) r->seed -= 2147483647u;
  return r->seed;
}
uint32_t oracle_random_uniform(oracle_random* r, int n) { return oracle_random_next(r) % (uint32_t)n; }
uint32_t oracle_random_skewed(oracle_random* r, int max_log) {
  return oracle_random_uniform(r, 1 << oracle_random_uniform(r, max_log + 1));
}

/* Per-block view of ReadPhysicalRecord over a complete image (checksum on, no retry):
 * for block b, out[4b..] = {n_records, first_bad, end_offset, end_code} where n_records
 * counts the records that pass the structural checks of db/log_reader.cc:317-384 when
 * every checksum is taken as good, first_bad is the index of the first of them whose
 * checksum (db/log_reader.cc:388-390) fails (0xffffffff if none), end_offset is where
 * that walk stops inside the block and end_code says why:
 * 0 block exactly full, 1 trailer shorter than a header, 2 the same at end of file,
 * 3 length runs past the block, 4 the same in the last block, 5 zero type and length,
 * 6 recyclable record of another log number. */
void oracle_wal_scan_blocks(const uint8_t* img, size_t n, uint64_t log_number, uint32_t* out) {
  size_t b = 0;
  for (size_t blk = 0; blk < n; blk += kBlockSize, b++) {
    size_t end = blk + kBlockSize < n ? blk + kBlockSize : n;
    int last = end == n;
    size_t p = blk;
    uint32_t count = 0, first_bad = 0xffffffffu, code = 0;
    for (;;) {
      size_t remaining = end - p;
      if (remaining < (size_t)kHeaderSize) {
        code = remaining == 0 ? 0 : (last ? 2 : 1);
        break;
      }
      const uint8_t* h = img + p;
      unsigned type = (unsigned)(int)(signed char)h[6];
      uint32_t len = h[4] | ((uint32_t)h[5] << 8);
      size_t hs = kHeaderSize;
      if (type >= kRecyclableFullType && type <= kRecyclableLastType) {
        hs = kRecyclableHeaderSize;
        if (remaining < hs) { code = last ? 2 : 1; break; }
        if ((uint64_t)get_le32(h + 7) != log_number) { code = 6; break; }
      }
      if (hs + len > remaining) { code = last ? 4 : 3; break; }
      if (type == kZeroType && len == 0) { code = 5; break; }
      uint32_t actual = oracle_crc32c_extend(0, h + 6, len + hs - 6);
      if (actual != oracle_crc32c_unmask(get_le32(h)) && first_bad == 0xffffffffu) first_bad = count;
      count++;
      p += hs + len;
    }
    out[4 * b + 0] = count;
    out[4 * b + 1] = first_bad;
    out[4 * b + 2] = (uint32_t)(p - blk);
    out[4 * b + 3] = code;
  }
}

/* n logical records of rec_len bytes each, back to back in data: the loop a group commit runs over the
 * reference writer (db/db_impl_write.cc:834-939 calls AddRecord once per merged batch; this is the same
 * writer fed many records, for timing the CPU side of the group-commit comparison). */
int oracle_wal_writer_add_records_fixed(void* writer, const void* data, size_t rec_len, size_t n) {
  const uint8_t* p = (const uint8_t*)data;
  for (size_t i = 0; i < n; i++) {
    int rc = oracle_wal_writer_add_record(writer, p + i * rec_len, rec_len);
    if (rc != 0) return rc;
  }
  return 0;
}

/* The log of the CPU-reference configuration: n logical records, each a write batch of one
 * Put (db/write_batch.cc:10-34: fixed64 sequence, fixed32 count = 1, tag kTypeValue = 1,
 * varint32 key length + key, varint32 value length + value) with a 24-byte key (first 8
 * bytes the little-endian record index, rest 'k') and a 2000-byte value drawn from
 * Random(301).Uniform(256): 12 + 1 + 1 + 24 + 2 + 2000 = 2040 bytes per record.
 * Appends them to the writer; returns the number of logical records written. */
size_t oracle_wal_write_dbbench_records(void* writer, size_t n, uint32_t seed) {
  oracle_random rnd;
  oracle_random_seed(&rnd, seed);
  uint8_t rec[2040];
  for (size_t i = 0; i < n; i++) {
    uint8_t* p = rec;
    uint64_t seq = i + 1;
    memcpy(p, &seq, 8); p += 8;
    put_le32(p, 1); p += 4;
    *p++ = 1;            /* kTypeValue */
    *p++ = 24;           /* varint32(24) */
    uint64_t idx = i;
    memcpy(p, &idx, 8);
    memset(p + 8, 'k', 16);
    p += 24;
    *p++ = 0xd0;         /* varint32(2000) = d0 0f */
    *p++ = 0x0f;
    for (int k = 0; k < 2000; k++) p[k] = (uint8_t)oracle_random_uniform(&rnd, 256);
    oracle_wal_writer_add_record(writer, rec, sizeof(rec));
  }
  return n;
}

/* Reads every record of the reader to the end; returns the number of logical records and adds up
 * their sizes (the timed CPU read-back of the reference configuration). */
size_t oracle_wal_read_all(void* reader, int mode, uint64_t* bytes_out) {
  const uint8_t* rec;
  size_t n, count = 0;
  uint64_t bytes = 0;
  while (oracle_wal_reader_read_record(reader, mode, &rec, &n)) {
    count++;
    bytes += n;
  }
  if (bytes_out) *bytes_out = bytes;
  return count;
}
