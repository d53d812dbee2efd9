// `ldb dump_wal` over the engine: every record of a log file is checksummed on the GPU in one
// pass, then the logical records are printed the way DumpWalFile does (tools/ldb_cmd.cc:1865-2037):
//   Sequence,Count,ByteSize,Physical Offset,Key(s)[ : value ]
// with one row per write batch and PUT(cf) / DELETE(cf) / ... entries with hex keys.
#pragma once
#include <stdint.h>

#include <string>

#include "files.h"
#include "status.h"
#include "wal_format.h"

namespace terarkdb_amd {

// The log number encoded in a file name "<dir>/NNNNNN.log", or 0 when the name does not parse
// (tools/ldb_cmd.cc:1995-2003, util/filename.cc:306-343).
uint64_t LogNumberFromFileName(const std::string& path);

// Appends the rows to *out and "Corruption detected in log file ..." lines to *err.  Returns a
// non-OK status only when the file cannot be read or the engine fails.
Status DumpWalFile(log::RecordVerifier* verifier, const std::string& wal_file, bool print_header, bool print_values,
                   std::string* out, std::string* err);

// One row body for a write batch (db/write_batch.cc:10-34,388-581; handler texts
// tools/ldb_cmd.cc:1873-1965): "seq,count,bytesize,offset,OP(cf) : 0xKEY ..." without the newline.
void WriteBatchRow(const Slice& rep, uint64_t physical_offset, bool print_values, std::string* row);

}  // namespace terarkdb_amd
