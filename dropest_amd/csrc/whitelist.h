// whitelist.h -- host-side whitelist ("real barcodes") files, parsed as the reference parses them.
#pragma once

#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "k_merge.h"
#include "util.h"

namespace dropest {

// Estimation/Merge/BarcodesParsing/{BarcodesParser,InDropBarcodesParser,ConstLengthBarcodesParser}.cpp
struct Whitelist {
	int kind = 0;                                   // DROPEST_BARCODES_INDROP / _CONST
	std::vector<std::vector<std::string>> parts;
	std::vector<size_t> part_lengths;               // length of the first entry of each part
	size_t total_length = 0;
	bool loaded = false;

	static std::string reverse_complement(const std::string &s) {   // Tools/UtilFunctions.cpp:97-116
		std::string r(s);
		for (size_t i = 0; i < s.size(); ++i) {
			const char c = s[s.size() - 1 - i];
			char o;
			switch (c) {
				case 'A': o = 'T'; break; case 'T': o = 'A'; break; case 'G': o = 'C'; break; case 'C': o = 'G'; break;
				case 'N': o = 'N'; break;
				default: throw IoError(std::string("whitelist contains an unexpected letter: '") + c + "'");
			}
			r[i] = o;
		}
		return r;
	}
	// BarcodesParser::read_line (BarcodesParser.cpp:117-144)
	static bool read_line(std::istream &in, std::vector<std::string> &out, bool equal_len) {
		std::string line;
		if (!std::getline(in, line)) return false;
		std::istringstream ss(line);
		std::string tok;
		size_t len0 = 0;
		while (ss >> tok) {
			if (len0 == 0) len0 = tok.size();
			else if (equal_len && len0 != tok.size()) throw IoError("All barcodes in one line must have the same length");
			out.push_back(reverse_complement(tok));
		}
		return true;
	}
	void load(int k, const std::string &file) {
		kind = k;
		std::ifstream f(file);
		if (f.fail()) throw IoError("Can't open barcodes file: '" + file + "'");
		parts.clear();
		if (kind == 0) {   // InDropBarcodesParser::get_barcodes_list (:15-30): exactly two lines
			parts.resize(2);
			for (int i = 0; i < 2; ++i)
				if (!read_line(f, parts[size_t(i)], false) || parts[size_t(i)].empty())
					throw IoError("File with barcodes (" + file + ") has wrong format");
		} else {           // ConstLengthBarcodesParser::get_barcodes_list (:50-68): one part per line
			std::vector<std::string> part;
			while (read_line(f, part, true)) {
				if (part.empty()) throw IoError("File with barcodes (" + file + ") has wrong format");
				parts.push_back(part);
				part.clear();
			}
		}
		if (parts.empty()) throw IoError("ERROR: empty barcodes list");   // BarcodesParser::init (:89-103)
		part_lengths.clear(); total_length = 0;
		for (auto &p : parts) {
			if (p.empty()) throw IoError("ERROR: empty barcodes list");
			part_lengths.push_back(p[0].size()); total_length += p[0].size();
			for (auto &s : p) {
				if (s.size() > size_t(WL_MAX_LEN)) throw UnsupportedError("whitelist part longer than 31 bases");
				if (s.find('N') != std::string::npos) throw UnsupportedError("whitelist entries containing N are not supported");
			}
		}
		// (more than WL_MAX_PARTS parts: the neighbour search runs on the host, merge_host.h search_merge_candidates_host -- the
		// reference has no limit, ConstLengthBarcodesParser.cpp:50-68)
		for (auto &p : parts) if (p.size() > 65535) throw UnsupportedError("whitelist part with more than 65535 entries");
		loaded = true;
	}
	// BarcodesParser::split_barcode: InDropBarcodesParser.cpp:32-39 / ConstLengthBarcodesParser.cpp:33-48
	std::vector<std::string> split(const std::string &cb) const {
		std::vector<std::string> out;
		if (kind == 0) {
			const size_t l2 = part_lengths[1];
			if (cb.size() < l2) throw InvalidError("Barcode '" + cb + "' is shorter than the second whitelist part");
			out.push_back(cb.substr(0, cb.size() - l2)); out.push_back(cb.substr(cb.size() - l2));
		} else {
			if (cb.size() != total_length)
				throw InvalidError("Barcode '" + cb + "' has wrong length (" + std::to_string(total_length) + " expected)");
			size_t at = 0;
			for (size_t l : part_lengths) { out.push_back(cb.substr(at, l)); at += l; }
		}
		for (auto &x : out) if (x.size() > size_t(WL_MAX_LEN)) throw UnsupportedError("barcode part longer than 31 bases");
		return out;
	}
};

}  // namespace dropest
