// rds_writer.h -- native writer of R's serialisation format (what saveRDS produces), so that
// ResultsPrinter::save_results (Estimation/ResultsPrinter.cpp:23-79, :442-452) needs neither R nor Rcpp.
//
// Format written: gzip stream of the XDR ("X\n") serialisation, version 2 -- the one R 3.x writes and every R since
// reads: header (version, writer R version, minimal reader version), then one item.  An item is a flags word
// (SEXPTYPE | object bit << 8 | attribute bit << 9 | tag bit << 10 | gp << 12) followed by its payload; attributes
// follow the payload as a tagged pairlist closed by NILVALUE_SXP; symbols are written once and referenced afterwards
// (REFSXP).  Only the node types the results need are implemented: NULL, symbols, pairlists (attributes only),
// character / integer / double / logical vectors, generic vectors (lists) and S4 objects (dgCMatrix).
// Host-only code (no HIP); integers are big-endian, doubles IEEE-754 big-endian.
//
// Round 6: the file is written as CONCATENATED GZIP MEMBERS (RFC 1952 2.2; R's gzfile / gzcon and zlib's gzread read them as one stream):
// a serial walk over the value cuts the serialisation into pieces of ~2 MB -- small items as literal bytes, the long vectors (the dgCMatrix
// slots i / x, saturation_info's reads / cbs / umis) as ranges of the vectors themselves -- and a pool of host threads byte-swaps and
// deflates the pieces side by side.  One gzopen stream on one thread was the slowest stage of BAM -> .rds.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace Rds {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
	enum Kind { Null, Integer, Real, String, List, S4 } kind = Null;
	std::vector<uint32_t> u32s;                                    // Integer / Real from 32-bit unsigned slots (no widened copy: the writer converts as it swaps)
	bool from_u32 = false;
	std::vector<uint64_t> packed;                                  // String from 2-bit packed base codes (include/dropest_amd.h: sentinel bit + 2 bits per base): decoded by the writer's threads
	bool from_packed = false;
	std::vector<int32_t> ints;
	std::vector<double> reals;
	std::vector<std::string> strings;
	std::vector<ValuePtr> items;                                   // List
	std::vector<std::pair<std::string, ValuePtr>> attributes;      // in order; S4: the slots + "class"
	bool is_object = false;                                        // has a class attribute (data.frame, S4)
};

ValuePtr null_value();
ValuePtr integers(std::vector<int32_t> v);
ValuePtr reals(std::vector<double> v);
ValuePtr strings(std::vector<std::string> v);
ValuePtr list(std::vector<ValuePtr> items);
ValuePtr named_list(std::vector<std::pair<std::string, ValuePtr>> items);
ValuePtr with_names(ValuePtr v, std::vector<std::string> names);
// as.data.frame of an integer matrix given by columns
ValuePtr data_frame(const std::vector<std::string> &col_names, const std::vector<std::string> &row_names,
                    std::vector<std::vector<int32_t>> columns);
// Matrix::dgCMatrix (CSC): p = column pointers, i = row indices (ascending inside a column), x = values
// (the three slot vectors are taken over, not copied: pass std::move(...) where the caller is done with them)
ValuePtr dgCMatrix(std::vector<uint32_t> colptr, std::vector<uint32_t> rowidx, std::vector<uint32_t> values,
                   const std::vector<std::string> &row_names, const std::vector<std::string> &col_names);
// a character vector of base strings given as packed codes (no escapes: bit 63 clear; 0 = the empty string) -- millions of barcodes / UMIs
// (saturation_info) become strings only inside the writer's pieces
ValuePtr strings_from_packed(std::vector<uint64_t> codes);
ValuePtr integers_from_u32(std::vector<uint32_t> v);      // values must be below 2^31
ValuePtr reals_from_u32(std::vector<uint32_t> v);

// saveRDS(value, path): gzip-compressed XDR serialisation.  Throws std::runtime_error on I/O errors.
// threads: host threads that swap + deflate the pieces (0: up to 16 of the machine's; 1: everything on the calling thread).
void save(const ValuePtr &value, const std::string &path, unsigned threads = 0);

// fn(piece) for piece = 0 .. n - 1 on up to `threads` host threads (0: up to 16 of the machine's); the first exception is rethrown
void parallel_pieces(size_t n, unsigned threads, const std::function<void(size_t)> &fn);

}  // namespace Rds
