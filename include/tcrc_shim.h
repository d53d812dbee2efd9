/*
 * C entry points over libtcrc.so's C++ classes that keep the reference's call surface:
 *   terarkdb_amd::log::Writer   ~ log::Writer  (db/log_writer.h:67-113)
 *   terarkdb_amd::log::Reader   ~ log::Reader  (db/log_reader.h:35-150)
 *   terarkdb_amd::WriteRawBlocks / CheckBlockChecksums ~ the checksum parts of
 *       BlockBasedTableBuilder::WriteRawBlock (table/block_based_table_builder.cc:580-646) and
 *       BlockFetcher::CheckBlockChecksum (table/block_fetcher.cc:36-71)
 * C++ callers include the headers under terarkdb_amd/host/ directly; these wrappers exist for
 * bindings (ctypes, cgo, JNI) and are what the test-suite drives.  Files are either POSIX
 * paths or in-memory files (the fakes of db/log_test.cc:49-114 and util/testutil.h:201-252).
 */
#ifndef TCRC_SHIM_H_
#define TCRC_SHIM_H_

#include <stddef.h>
#include <stdint.h>

#include "tcrc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Checksum seams as callbacks (used to exercise the host logic without a GPU): return 0 on success. */
typedef int (*tcrc_shim_stamp_fn)(void* user, const char* base, size_t base_bytes, const tcrc_desc* spans, size_t n,
                                  uint32_t* masked_out);
typedef int (*tcrc_shim_scan_fn)(void* user, const char* img, size_t n, uint64_t log_number, tcrc_wal_block* blocks);

/* in-memory file */
void* tcrc_shim_memfile_new(void);
void tcrc_shim_memfile_free(void* f);
size_t tcrc_shim_memfile_size(void* f);
char* tcrc_shim_memfile_data(void* f);
void tcrc_shim_memfile_drop(void* f, size_t bytes);
void tcrc_shim_memfile_append(void* f, const void* data, size_t n);

/* log::Writer(dest, log_number, recycle_log_files, manual_flush): AddRecord / WriteBuffer */
void* tcrc_shim_wal_writer_new(tcrc_ctx* ctx, void* memfile, uint64_t log_number, int recycle, int manual_flush,
                               int overwrite);
void* tcrc_shim_wal_writer_new_cb(tcrc_shim_stamp_fn fn, void* user, void* memfile, uint64_t log_number, int recycle,
                                  int manual_flush, int overwrite);
void* tcrc_shim_wal_writer_open(tcrc_ctx* ctx, const char* path, uint64_t log_number, int recycle, int manual_flush);
int tcrc_shim_wal_writer_add_record(void* w, const void* data, size_t n);
/* AddRecord for n records of record_len bytes laid out back to back at base */
int tcrc_shim_wal_writer_add_records_fixed(void* w, const void* base, size_t record_len, size_t n);
int tcrc_shim_wal_writer_write_buffer(void* w);
const char* tcrc_shim_wal_writer_status(void* w);
uint64_t tcrc_shim_wal_writer_physical_records(void* w);
uint64_t tcrc_shim_wal_writer_crc_bytes(void* w);
void tcrc_shim_wal_writer_free(void* w);

/* log::Reader(file, reporter, checksum, log_num, retry_after_eof): ReadRecord / UnmarkEOF / ... */
void* tcrc_shim_wal_reader_new(tcrc_ctx* ctx, void* memfile, int checksum, uint64_t log_number, int retry_after_eof);
void* tcrc_shim_wal_reader_new_cb(tcrc_shim_scan_fn fn, void* user, void* memfile, int checksum, uint64_t log_number,
                                  int retry_after_eof);
/* Readers over a real file (POSIX sequential reads).  With retry_after_eof the reader keeps polling the
 * file for the rest of a record that is still being written (db/log_reader.cc:294-296), as a tailing
 * reader does. */
void* tcrc_shim_wal_reader_open(tcrc_ctx* ctx, const char* path, int checksum, uint64_t log_number, int retry_after_eof);
void* tcrc_shim_wal_reader_open_cb(tcrc_shim_scan_fn fn, void* user, const char* path, int checksum,
                                   uint64_t log_number, int retry_after_eof);
/* Verifies img[0..n) (or, with img == NULL, the in-memory file's present contents) in one
 * pass and attaches the verdicts, so ReadRecord does not ask block by block. */
int tcrc_shim_wal_reader_attach_image(void* r, const void* img, size_t n);
/* mode = WALRecoveryMode value; returns 1 and the record, or 0 at end of input. */
int tcrc_shim_wal_reader_read_record(void* r, int mode, const char** rec, size_t* n);
size_t tcrc_shim_wal_reader_dropped_bytes(void* r);
const char* tcrc_shim_wal_reader_report(void* r);
int tcrc_shim_wal_reader_is_eof(void* r);
void tcrc_shim_wal_reader_unmark_eof(void* r);
uint64_t tcrc_shim_wal_reader_last_record_offset(void* r);
uint64_t tcrc_shim_wal_reader_verifier_calls(void* r);
void tcrc_shim_wal_reader_force_error(void* r, size_t position);
void tcrc_shim_wal_reader_force_eof(void* r, size_t position);
int tcrc_shim_wal_reader_read_after_partial(void* r);
void tcrc_shim_wal_reader_free(void* r);

/* SST block trailers.  checksum_type: 0 none, 1 CRC32c, 2/3 xxHash kinds (declined: returns -2). */
int tcrc_shim_block_trailers(tcrc_ctx* ctx, const void* base, const uint64_t* offsets, const uint64_t* sizes,
                             const uint8_t* types, size_t n, int checksum_type, uint8_t* trailers_out);
int tcrc_shim_write_raw_blocks(tcrc_ctx* ctx, void* memfile, const void* base, const uint64_t* offsets,
                               const uint64_t* sizes, const uint8_t* types, size_t n, int checksum_type,
                               uint64_t* handle_offsets_out);
/* returns 0 all good, 1 at least one mismatch (first_message holds the reference's Corruption text), <0 error */
int tcrc_shim_check_blocks(tcrc_ctx* ctx, const void* image, size_t image_size, uint64_t image_file_offset,
                           const uint64_t* handle_offsets, const uint64_t* handle_sizes, size_t n, int checksum_type,
                           const char* file_name, uint8_t* ok_out, char* first_message, size_t first_message_cap);

/* Whole-file checksum verification of a block-based table file: BlockBasedTable::VerifyChecksum
 * (table/block_based_table_reader.cc:2616-2695) as reached from VerifySstFileChecksum
 * (db/convenience.cc:35-64) and `sst_dump --command=verify` (tools/sst_dump_tool.cc:656-665).
 * extent_bytes / read_threads 0 = defaults; index_type_hint = BlockBasedTableOptions::index_type to assume
 * when the file does not say.  stats_out (may be NULL) receives 7 values: file size, meta blocks, index
 * blocks, data blocks, bytes covered, mismatching blocks, engine calls.  message receives Status::ToString().
 * Returns 0 OK, 1 Corruption, 2 NotSupported (declined: xxHash checksums, compressed index), -1 other. */
int tcrc_shim_sst_verify(tcrc_ctx* ctx, const char* path, size_t extent_bytes, int read_threads, int index_type_hint,
                         uint64_t* stats_out, char* message, size_t message_cap);
/* The same driver with the checksum arithmetic supplied by the caller (tests of the host logic). */
typedef int (*tcrc_shim_check_fn)(void* user, const void* image, size_t image_bytes, const tcrc_desc* spans, size_t n,
                                  uint32_t* computed_masked, uint8_t* ok);
int tcrc_shim_sst_verify_cb(tcrc_shim_check_fn fn, void* user, const char* path, size_t extent_bytes, int read_threads,
                            int index_type_hint, uint64_t* stats_out, char* message, size_t message_cap);

/* Table footer (table/format.cc:113-210): encode writes 48 bytes (legacy != 0: no checksum byte, version 0,
 * legacy magic number) or 53 bytes into out and returns the length; decode takes the tail of a file (at
 * least its last 48 bytes) and returns 0 or -1 with the Status text in message.  A legacy magic number
 * comes back up-converted with version 0 and checksum kCRC32c, as the reference reports it. */
int tcrc_shim_footer_encode(int checksum, uint32_t version, int legacy, uint64_t metaindex_offset,
                            uint64_t metaindex_size, uint64_t index_offset, uint64_t index_size, uint8_t* out);
int tcrc_shim_footer_decode(const void* tail, size_t n, int* checksum, uint32_t* version, uint64_t* magic,
                            uint64_t* handles4, char* message, size_t message_cap);

/* `ldb dump_wal` (tools/ldb_cmd.cc:1967-2037): checks every record of the log file on the GPU, then
 * writes the rows ("Sequence,Count,ByteSize,Physical Offset,Key(s)" ...) to out and the
 * "Corruption detected in log file ..." lines to err, both NUL-terminated and cut at their caps;
 * *out_len / *err_len (may be NULL) receive the untruncated lengths.  Returns 0, or -1 when the file
 * cannot be read or the engine fails (err then ends with the Status text). */
int tcrc_shim_dump_wal(tcrc_ctx* ctx, const char* path, int print_header, int print_values, char* out, size_t out_cap,
                       size_t* out_len, char* err, size_t err_cap, size_t* err_len);
int tcrc_shim_dump_wal_cb(tcrc_shim_scan_fn fn, void* user, const char* path, int print_header, int print_values,
                          char* out, size_t out_cap, size_t* out_len, char* err, size_t err_cap, size_t* err_len);

/* Records of the persistent block cache (utilities/persistent_cache/block_cache_tier_file.cc:55-192):
 * serialize writes n records [magic | crc | key size | value size | key | value] for the keys and values at
 * base + key_off[i] / val_off[i] into out (out_cap bytes; *out_len receives the size, also when out is too
 * small or NULL), the crc32c of all records computed in one engine submission (CacheRecord::ComputeCRC,
 * :108-118).  check gives ok_out[i] = what CacheRecord::Deserialize returns (:162-192) for the record of
 * sizes[i] bytes at image + offs[i].  Return 0, -1 on errors, -3 (serialize) when out_cap is too small. */
int tcrc_shim_cache_records_serialize(tcrc_ctx* ctx, const void* base, const uint64_t* key_off, const uint32_t* key_len,
                                      const uint64_t* val_off, const uint32_t* val_len, size_t n, void* out,
                                      size_t out_cap, size_t* out_len);
int tcrc_shim_cache_records_check(tcrc_ctx* ctx, const void* image, size_t image_bytes, const uint64_t* offs,
                                  const uint32_t* sizes, size_t n, uint8_t* ok_out);

/* Checks every physical record of one log file the way recovery's reader meets it (db/log_reader.cc:276-406
 * with checksum = true): the file is read in pieces of extent_bytes (0 = default 256 MiB) with read_threads
 * (0 = default) positional reads each, the read of one piece overlapping the engine's work on the previous one.
 * blocks_out (may be NULL) receives up to blocks_cap entries, one per 32 KiB block.  report_out (may be NULL)
 * receives 5 values: file bytes, physical records, records with a checksum mismatch, first 32 KiB block holding
 * one (~0 if none), TCRC_WAL_END_* of the file's last block.  Returns 0, or -1 with the Status text in message. */
int tcrc_shim_wal_verify_file(tcrc_ctx* ctx, const char* path, uint64_t log_number, size_t extent_bytes,
                              int read_threads, tcrc_wal_block* blocks_out, size_t blocks_cap, uint64_t* report_out,
                              char* message, size_t message_cap);

/* File-level DBImpl::VerifyChecksum (db/db_impl.cc:4657-4690) over a database directory: every NNNNNN.sst in
 * file-number order (stopping at the first corrupt one, whose name goes to bad_file and whose status to
 * message), then a checksum scan of NNNNNN.log and MANIFEST-NNNNNN.  Files are dealt to shard_index of
 * shard_count by size, largest first to the shard with the fewest bytes so far (one shard per GPU; every
 * caller computes the same deal from the same listing).  report_out (may be NULL) receives 8 values: table files
 * verified, table files declined, table blocks, table bytes covered, log files, log bytes, log records, log
 * records with a checksum mismatch.  Returns 0 OK, 1 Corruption, -1 other errors. */
int tcrc_shim_verify_db(tcrc_ctx* ctx, const char* dir, uint32_t shard_index, uint32_t shard_count, uint64_t* report_out,
                        char* message, size_t message_cap, char* bad_file, size_t bad_file_cap);

/* The same over a group of GPUs in one process (tcrc_group_init, tcrc.h): member m verifies shard m of the
 * directory's files on a host thread of its own; the report adds up, and of several failing tables the status
 * is that of the one with the smallest file number.  DBImpl::VerifyChecksum (db/db_impl.cc:4657-4690) spread
 * over the GPUs of one machine. */
int tcrc_shim_verify_db_group(tcrc_group* group, const char* dir, uint64_t* report_out, char* message, size_t message_cap,
                              char* bad_file, size_t bad_file_cap);

/* Whole-file crc32c as the backup engine computes it (utilities/backupable/backupable_db.cc:1414-1452
 * CalculateChecksum; :1195-1283 CopyOrCreateFile): size_limit 0 = whole file; buffer_size 0 = default.
 * Return 0, or -1 with the Status text in message (if not NULL). */
int tcrc_shim_file_checksum(tcrc_ctx* ctx, const char* path, uint64_t size_limit, size_t buffer_size,
                            uint32_t* checksum_out, uint64_t* size_out, char* message, size_t message_cap);
int tcrc_shim_copy_file(tcrc_ctx* ctx, const char* src, const char* dst, uint64_t size_limit, size_t buffer_size,
                        int sync, uint32_t* checksum_out, uint64_t* size_out, char* message, size_t message_cap);

#ifdef __cplusplus
}
#endif
#endif /* TCRC_SHIM_H_ */
