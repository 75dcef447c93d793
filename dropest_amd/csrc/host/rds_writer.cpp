// rds_writer.cpp -- see rds_writer.h.  Restates the writer side of R's serialize.c (WriteItem) for the node types the
// dropEst results use; the layout was checked against files written by R itself (tests/test_rds.py parses the
// reference's data/*.rds with the same reader that checks this writer's output).
#include "rds_writer.h"

#include <zlib.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <unordered_map>

namespace Rds {

void parallel_pieces(size_t n, unsigned threads, const std::function<void(size_t)> &fn) {
	if (threads == 0) threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
	threads = unsigned(std::min<size_t>(threads, n));
	if (threads <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
	std::atomic<size_t> next{0};
	std::exception_ptr error;
	std::mutex m;
	auto work = [&] {
		for (;;) {
			const size_t i = next.fetch_add(1, std::memory_order_relaxed);
			if (i >= n) return;
			try { fn(i); }
			catch (...) { std::lock_guard<std::mutex> lk(m); if (!error) error = std::current_exception(); next.store(n); return; }
		}
	};
	std::vector<std::thread> pool;
	for (unsigned t = 1; t < threads; ++t) pool.emplace_back(work);
	work();
	for (auto &t : pool) t.join();
	if (error) std::rethrow_exception(error);
}

namespace {

enum : int { SYMSXP = 1, LISTSXP = 2, CHARSXP = 9, LGLSXP = 10, INTSXP = 13, REALSXP = 14, STRSXP = 16, VECSXP = 19, S4SXP = 25,
             REFSXP = 255, NILVALUE_SXP = 254 };
constexpr int IS_OBJECT_BIT = 1 << 8, HAS_ATTR_BIT = 1 << 9, HAS_TAG_BIT = 1 << 10;
constexpr int GP_ASCII = 1 << 6, GP_UTF8 = 1 << 3, GP_S4 = 1 << 4;
constexpr size_t PIECE_BYTES = size_t(2) << 20;      // serialised bytes per gzip member (about: strings are estimated)
constexpr size_t LONG_VECTOR = 4096;                 // vectors from here on are referenced, not copied into the literal bytes

inline void put32(unsigned char *b, uint32_t u) { b[0] = static_cast<unsigned char>(u >> 24); b[1] = static_cast<unsigned char>(u >> 16); b[2] = static_cast<unsigned char>(u >> 8); b[3] = static_cast<unsigned char>(u); }
inline void put64(unsigned char *b, uint64_t u) { put32(b, uint32_t(u >> 32)); put32(b + 4, uint32_t(u)); }
inline size_t charsxp_size(const std::string &s) { return 8 + s.size(); }
inline size_t packed_length(uint64_t code) { return code ? size_t(63 - __builtin_clzll(code)) / 2 : 0; }   // bases of a packed code (sentinel bit above them)
inline void charsxp_to(std::vector<unsigned char> &out, const std::string &s) {
	bool ascii = true;
	for (unsigned char c : s) ascii &= c < 128;
	if (s.size() > 0x7FFFFFFFull) throw std::runtime_error("rds: long vectors are not supported");
	const size_t at = out.size();
	out.resize(at + 8 + s.size());
	put32(&out[at], uint32_t(CHARSXP | ((ascii ? GP_ASCII : GP_UTF8) << 12)));
	put32(&out[at + 4], uint32_t(s.size()));
	std::memcpy(&out[at + 8], s.data(), s.size());
}

// One stretch of the serialisation: literal bytes [a, b) of the arena, or elements [a, b) of a vector the value owns.
struct Part {
	enum Kind { Literal, I32, U32AsI32, F64, U32AsF64, Strings, Packed } kind;
	const void *src;
	size_t a, b;
};

class Plan {
	std::vector<unsigned char> lit;                 // the literal bytes of the whole stream, one after the other
	std::vector<Part> parts;
	std::vector<size_t> piece_end;                  // parts[piece_end[k - 1] .. piece_end[k]) = piece k
	size_t piece_fill = 0;
	std::unordered_map<std::string, int> symbols;   // name -> 1-based reference index
	void close_piece() { if (piece_fill) { piece_end.push_back(parts.size()); piece_fill = 0; } }
	void add_bytes(size_t n) { piece_fill += n; if (piece_fill >= PIECE_BYTES) close_piece(); }
	void raw(const void *p, size_t n) {
		const unsigned char *c = static_cast<const unsigned char *>(p);
		const size_t at = lit.size();
		lit.insert(lit.end(), c, c + n);
		if (!parts.empty() && parts.back().kind == Part::Literal && parts.back().b == at && (piece_end.empty() || piece_end.back() != parts.size())) parts.back().b = at + n;
		else parts.push_back(Part{Part::Literal, nullptr, at, at + n});
		add_bytes(n);
	}
	// elements [0, n) of a long vector, in stretches that fill the pieces evenly
	void range(Part::Kind kind, const void *src, size_t n, size_t elem_bytes) {
		for (size_t a = 0; a < n;) {
			const size_t room = PIECE_BYTES > piece_fill ? PIECE_BYTES - piece_fill : 0;
			const size_t take = std::min(n - a, std::max<size_t>(1, room / elem_bytes));
			parts.push_back(Part{kind, src, a, a + take});
			a += take;
			add_bytes(take * elem_bytes);
		}
	}
	void i32(int32_t v) { unsigned char b[4]; put32(b, uint32_t(v)); raw(b, 4); }
	void f64(double d) { uint64_t u; std::memcpy(&u, &d, 8); unsigned char b[8]; put64(b, u); raw(b, 8); }
	void length(size_t n) {
		if (n > 0x7FFFFFFFull) throw std::runtime_error("rds: long vectors are not supported");
		i32(int32_t(n));
	}
	void charsxp(const std::string &s) { std::vector<unsigned char> b; charsxp_to(b, s); raw(b.data(), b.size()); }
	void symbol(const std::string &name) {
		auto it = symbols.find(name);
		if (it != symbols.end()) { i32((it->second << 8) | REFSXP); return; }
		symbols.emplace(name, int(symbols.size()) + 1);
		i32(SYMSXP);
		charsxp(name);
	}
	void attributes(const std::vector<std::pair<std::string, ValuePtr>> &attrs) {
		for (auto const &a : attrs) {
			i32(LISTSXP | HAS_TAG_BIT);
			symbol(a.first);
			item(*a.second);
		}
		i32(NILVALUE_SXP);
	}
public:
	void header() {
		raw("X\n", 2);
		i32(2);                                  // serialisation version
		i32((3 << 16) | (4 << 8) | 0);           // R_VERSION of the "writer": 3.4.0
		i32((2 << 16) | (3 << 8) | 0);           // minimal R version that can read it: 2.3.0
	}
	void item(const Value &v) {
		const int extra = (v.is_object ? IS_OBJECT_BIT : 0) | (v.attributes.empty() ? 0 : HAS_ATTR_BIT);
		switch (v.kind) {
			case Value::Null: i32(NILVALUE_SXP); return;
			case Value::Integer:
				i32(INTSXP | extra);
				if (v.from_u32) { length(v.u32s.size()); range(Part::U32AsI32, v.u32s.data(), v.u32s.size(), 4); }
				else if (v.ints.size() >= LONG_VECTOR) { length(v.ints.size()); range(Part::I32, v.ints.data(), v.ints.size(), 4); }
				else { length(v.ints.size()); for (int32_t x : v.ints) i32(x); }
				break;
			case Value::Real:
				i32(REALSXP | extra);
				if (v.from_u32) { length(v.u32s.size()); range(Part::U32AsF64, v.u32s.data(), v.u32s.size(), 8); }
				else if (v.reals.size() >= LONG_VECTOR) { length(v.reals.size()); range(Part::F64, v.reals.data(), v.reals.size(), 8); }
				else { length(v.reals.size()); for (double x : v.reals) f64(x); }
				break;
			case Value::String:
				if (v.from_packed) {
					i32(STRSXP | extra); length(v.packed.size());
					size_t bytes = 0;      // (sized from every 64th code)
					for (size_t k = 0; k < v.packed.size(); k += 64) bytes += 8 + packed_length(v.packed[k]);
					range(Part::Packed, v.packed.data(), v.packed.size(), std::max<size_t>(8, bytes / std::max<size_t>(1, (v.packed.size() + 63) / 64)));
					break;
				}
				i32(STRSXP | extra); length(v.strings.size());
				if (v.strings.size() >= LONG_VECTOR) {
					// (sized from the vector's own mean: the pieces need not be equal, only bounded)
					size_t bytes = 0;
					for (size_t k = 0; k < v.strings.size(); k += 64) bytes += charsxp_size(v.strings[k]);
					range(Part::Strings, v.strings.data(), v.strings.size(), std::max<size_t>(8, bytes / ((v.strings.size() + 63) / 64)));
				} else for (auto const &s : v.strings) charsxp(s);
				break;
			case Value::List:
				i32(VECSXP | extra); length(v.items.size());
				for (auto const &p : v.items) item(*p);
				break;
			case Value::S4:
				i32(S4SXP | IS_OBJECT_BIT | HAS_ATTR_BIT | (GP_S4 << 12));
				break;
		}
		if (!v.attributes.empty() || v.kind == Value::S4) attributes(v.attributes);
	}
	size_t pieces() { close_piece(); return piece_end.size(); }
	// the bytes of piece k
	void materialise(size_t k, std::vector<unsigned char> &out) const {
		out.clear();
		for (size_t q = k ? piece_end[k - 1] : 0; q < piece_end[k]; ++q) {
			const Part &p = parts[q];
			const size_t n = p.b - p.a, at = out.size();
			switch (p.kind) {
				case Part::Literal: out.insert(out.end(), lit.begin() + long(p.a), lit.begin() + long(p.b)); break;
				case Part::I32: case Part::U32AsI32: {
					const uint32_t *s = static_cast<const uint32_t *>(p.src) + p.a;
					out.resize(at + n * 4);
					unsigned char *d = &out[at];
					for (size_t i = 0; i < n; ++i) { const uint32_t u = __builtin_bswap32(s[i]); std::memcpy(d + 4 * i, &u, 4); }
					if (p.kind == Part::U32AsI32) for (size_t i = 0; i < n; ++i) if (s[i] > 0x7FFFFFFFu) throw std::runtime_error("rds: an integer slot beyond 2^31 - 1");
					break;
				}
				case Part::F64: {
					const uint64_t *s = static_cast<const uint64_t *>(p.src) + p.a;
					out.resize(at + n * 8);
					unsigned char *d = &out[at];
					for (size_t i = 0; i < n; ++i) { const uint64_t u = __builtin_bswap64(s[i]); std::memcpy(d + 8 * i, &u, 8); }
					break;
				}
				case Part::U32AsF64: {
					const uint32_t *s = static_cast<const uint32_t *>(p.src) + p.a;
					out.resize(at + n * 8);
					unsigned char *d = &out[at];
					for (size_t i = 0; i < n; ++i) { const double x = double(s[i]); uint64_t u; std::memcpy(&u, &x, 8); u = __builtin_bswap64(u); std::memcpy(d + 8 * i, &u, 8); }
					break;
				}
				case Part::Packed: {
					const uint64_t *c = static_cast<const uint64_t *>(p.src);
					for (size_t i = p.a; i < p.b; ++i) {
						if (c[i] >> 63) throw std::runtime_error("rds: an escaped code in a packed string vector");
						const size_t len = packed_length(c[i]), w = out.size();
						out.resize(w + 8 + len);
						put32(&out[w], uint32_t(CHARSXP | (GP_ASCII << 12)));
						put32(&out[w + 4], uint32_t(len));
						for (size_t b = 0; b < len; ++b) out[w + 8 + b] = static_cast<unsigned char>("ACGT"[(c[i] >> (2 * (len - 1 - b))) & 3]);
					}
					break;
				}
				case Part::Strings: {
					const std::string *s = static_cast<const std::string *>(p.src);
					for (size_t i = p.a; i < p.b; ++i) charsxp_to(out, s[i]);
					break;
				}
			}
		}
	}
};

// one gzip member (RFC 1952) around `in`
void gzip_member(const std::vector<unsigned char> &in, std::vector<unsigned char> &out, int level) {
	z_stream z{};
	if (deflateInit2(&z, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("rds: deflateInit2 failed");
	out.resize(deflateBound(&z, uLong(in.size())) + 32);
	z.next_in = const_cast<Bytef *>(in.data()); z.avail_in = uInt(in.size());
	z.next_out = out.data(); z.avail_out = uInt(out.size());
	const int rc = deflate(&z, Z_FINISH);
	const size_t n = z.total_out;
	deflateEnd(&z);
	if (rc != Z_STREAM_END) throw std::runtime_error("rds: deflate failed");
	out.resize(n);
}

ValuePtr make(Value::Kind k) { auto p = std::make_shared<Value>(); p->kind = k; return p; }

}  // namespace

ValuePtr null_value() { return make(Value::Null); }
ValuePtr integers(std::vector<int32_t> v) { auto p = make(Value::Integer); p->ints = std::move(v); return p; }
ValuePtr reals(std::vector<double> v) { auto p = make(Value::Real); p->reals = std::move(v); return p; }
ValuePtr integers_from_u32(std::vector<uint32_t> v) { auto p = make(Value::Integer); p->u32s = std::move(v); p->from_u32 = true; return p; }
ValuePtr reals_from_u32(std::vector<uint32_t> v) { auto p = make(Value::Real); p->u32s = std::move(v); p->from_u32 = true; return p; }
ValuePtr strings_from_packed(std::vector<uint64_t> codes) { auto p = make(Value::String); p->packed = std::move(codes); p->from_packed = true; return p; }
ValuePtr strings(std::vector<std::string> v) { auto p = make(Value::String); p->strings = std::move(v); return p; }
ValuePtr list(std::vector<ValuePtr> items) { auto p = make(Value::List); p->items = std::move(items); return p; }

ValuePtr with_names(ValuePtr v, std::vector<std::string> names) {
	v->attributes.emplace_back("names", strings(std::move(names)));
	return v;
}

ValuePtr named_list(std::vector<std::pair<std::string, ValuePtr>> items) {
	std::vector<std::string> names;
	std::vector<ValuePtr> vals;
	for (auto &kv : items) { names.push_back(kv.first); vals.push_back(kv.second); }
	return with_names(list(std::move(vals)), std::move(names));
}

ValuePtr data_frame(const std::vector<std::string> &col_names, const std::vector<std::string> &row_names,
                    std::vector<std::vector<int32_t>> columns) {
	std::vector<ValuePtr> cols;
	for (auto &c : columns) cols.push_back(integers(std::move(c)));
	ValuePtr df = list(std::move(cols));
	df->attributes.emplace_back("names", strings(col_names));
	df->attributes.emplace_back("row.names", strings(row_names));
	df->attributes.emplace_back("class", strings({"data.frame"}));
	df->is_object = true;
	return df;
}

ValuePtr dgCMatrix(std::vector<uint32_t> colptr, std::vector<uint32_t> rowidx, std::vector<uint32_t> values,
                   const std::vector<std::string> &row_names, const std::vector<std::string> &col_names) {
	auto m = make(Value::S4);
	m->is_object = true;
	if (colptr.empty()) colptr.push_back(0);
	ValuePtr cls = strings({"dgCMatrix"});
	cls->attributes.emplace_back("package", strings({"Matrix"}));
	m->attributes.emplace_back("i", integers_from_u32(std::move(rowidx)));
	m->attributes.emplace_back("p", integers_from_u32(std::move(colptr)));
	m->attributes.emplace_back("Dim", integers({int32_t(row_names.size()), int32_t(col_names.size())}));
	m->attributes.emplace_back("Dimnames", list({strings(row_names), strings(col_names)}));
	m->attributes.emplace_back("x", reals_from_u32(std::move(values)));
	m->attributes.emplace_back("factors", list({}));
	m->attributes.emplace_back("class", cls);
	return m;
}

void save(const ValuePtr &value, const std::string &path, unsigned threads) {
	Plan plan;
	plan.header();
	plan.item(*value);
	const size_t n = plan.pieces();
	std::vector<std::vector<unsigned char>> member(n);
	parallel_pieces(n, threads, [&](size_t k) {
		std::vector<unsigned char> bytes;
		plan.materialise(k, bytes);
		gzip_member(bytes, member[k], 4);        // (level 4, as the single stream of earlier rounds: "wb4")
	});
	FILE *f = std::fopen(path.c_str(), "wb");
	if (!f) throw std::runtime_error("Can't open file: " + path);
	bool ok = true;
	for (auto const &m : member) ok = ok && (m.empty() || std::fwrite(m.data(), 1, m.size(), f) == m.size());
	ok = (std::fclose(f) == 0) && ok;
	if (!ok) throw std::runtime_error("rds: write failed");
}

}  // namespace Rds
