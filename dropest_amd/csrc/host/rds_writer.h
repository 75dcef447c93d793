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
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace Rds {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
	enum Kind { Null, Integer, Real, String, List, S4 } kind = Null;
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
ValuePtr dgCMatrix(const std::vector<uint32_t> &colptr, const std::vector<uint32_t> &rowidx, const std::vector<uint32_t> &values,
                   const std::vector<std::string> &row_names, const std::vector<std::string> &col_names);

// saveRDS(value, path): gzip-compressed XDR serialisation.  Throws std::runtime_error on I/O errors.
void save(const ValuePtr &value, const std::string &path);

}  // namespace Rds
