// rds_writer.cpp -- see rds_writer.h.  Restates the writer side of R's serialize.c (WriteItem) for the node types the
// dropEst results use; the layout was checked against files written by R itself (tests/test_rds.py parses the
// reference's data/*.rds with the same reader that checks this writer's output).
#include "rds_writer.h"

#include <zlib.h>

#include <cstring>
#include <stdexcept>
#include <unordered_map>

namespace Rds {

namespace {

enum : int { SYMSXP = 1, LISTSXP = 2, CHARSXP = 9, LGLSXP = 10, INTSXP = 13, REALSXP = 14, STRSXP = 16, VECSXP = 19, S4SXP = 25,
             REFSXP = 255, NILVALUE_SXP = 254 };
constexpr int IS_OBJECT_BIT = 1 << 8, HAS_ATTR_BIT = 1 << 9, HAS_TAG_BIT = 1 << 10;
constexpr int GP_ASCII = 1 << 6, GP_UTF8 = 1 << 3, GP_S4 = 1 << 4;

class Out {
	gzFile f;
	std::vector<unsigned char> buf;
	std::unordered_map<std::string, int> symbols;   // name -> 1-based reference index
public:
	explicit Out(const std::string &path) : f(gzopen(path.c_str(), "wb4")) {
		if (!f) throw std::runtime_error("Can't open file: " + path);
		buf.reserve(1 << 20);
	}
	~Out() { if (f) gzclose(f); }
	void flush() {
		if (!buf.empty() && gzwrite(f, buf.data(), unsigned(buf.size())) != int(buf.size())) throw std::runtime_error("rds: write failed");
		buf.clear();
	}
	void close() {
		flush();
		const int rc = gzclose(f);
		f = nullptr;
		if (rc != Z_OK) throw std::runtime_error("rds: close failed");
	}
	void raw(const void *p, size_t n) {
		const unsigned char *c = static_cast<const unsigned char *>(p);
		buf.insert(buf.end(), c, c + n);
		if (buf.size() >= (1u << 20)) flush();
	}
	void i32(int32_t v) {
		const uint32_t u = uint32_t(v);
		const unsigned char b[4] = {static_cast<unsigned char>(u >> 24), static_cast<unsigned char>(u >> 16),
		                            static_cast<unsigned char>(u >> 8), static_cast<unsigned char>(u)};
		raw(b, 4);
	}
	void f64(double d) {
		uint64_t u;
		std::memcpy(&u, &d, 8);
		unsigned char b[8];
		for (int k = 0; k < 8; ++k) b[k] = static_cast<unsigned char>(u >> (56 - 8 * k));
		raw(b, 8);
	}
	void length(size_t n) {
		if (n > 0x7FFFFFFFull) throw std::runtime_error("rds: long vectors are not supported");
		i32(int32_t(n));
	}
	void charsxp(const std::string &s) {
		bool ascii = true;
		for (unsigned char c : s) ascii &= c < 128;
		i32(CHARSXP | ((ascii ? GP_ASCII : GP_UTF8) << 12));
		length(s.size());
		raw(s.data(), s.size());
	}
	void symbol(const std::string &name) {
		auto it = symbols.find(name);
		if (it != symbols.end()) { i32((it->second << 8) | REFSXP); return; }
		symbols.emplace(name, int(symbols.size()) + 1);
		i32(SYMSXP);
		charsxp(name);
	}
	void attributes(const std::vector<std::pair<std::string, ValuePtr>> &attrs);
	void item(const Value &v);
};

void Out::attributes(const std::vector<std::pair<std::string, ValuePtr>> &attrs) {
	for (auto const &a : attrs) {
		i32(LISTSXP | HAS_TAG_BIT);
		symbol(a.first);
		item(*a.second);
	}
	i32(NILVALUE_SXP);
}

void Out::item(const Value &v) {
	const int extra = (v.is_object ? IS_OBJECT_BIT : 0) | (v.attributes.empty() ? 0 : HAS_ATTR_BIT);
	switch (v.kind) {
		case Value::Null: i32(NILVALUE_SXP); return;
		case Value::Integer:
			i32(INTSXP | extra); length(v.ints.size());
			for (int32_t x : v.ints) i32(x);
			break;
		case Value::Real:
			i32(REALSXP | extra); length(v.reals.size());
			for (double x : v.reals) f64(x);
			break;
		case Value::String:
			i32(STRSXP | extra); length(v.strings.size());
			for (auto const &s : v.strings) charsxp(s);
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

ValuePtr make(Value::Kind k) { auto p = std::make_shared<Value>(); p->kind = k; return p; }

}  // namespace

ValuePtr null_value() { return make(Value::Null); }
ValuePtr integers(std::vector<int32_t> v) { auto p = make(Value::Integer); p->ints = std::move(v); return p; }
ValuePtr reals(std::vector<double> v) { auto p = make(Value::Real); p->reals = std::move(v); return p; }
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

ValuePtr dgCMatrix(const std::vector<uint32_t> &colptr, const std::vector<uint32_t> &rowidx, const std::vector<uint32_t> &values,
                   const std::vector<std::string> &row_names, const std::vector<std::string> &col_names) {
	auto m = make(Value::S4);
	m->is_object = true;
	std::vector<int32_t> p(colptr.begin(), colptr.end()), i(rowidx.begin(), rowidx.end());
	if (p.empty()) p.push_back(0);
	std::vector<double> x(values.begin(), values.end());
	ValuePtr cls = strings({"dgCMatrix"});
	cls->attributes.emplace_back("package", strings({"Matrix"}));
	m->attributes.emplace_back("i", integers(std::move(i)));
	m->attributes.emplace_back("p", integers(std::move(p)));
	m->attributes.emplace_back("Dim", integers({int32_t(row_names.size()), int32_t(col_names.size())}));
	m->attributes.emplace_back("Dimnames", list({strings(row_names), strings(col_names)}));
	m->attributes.emplace_back("x", reals(std::move(x)));
	m->attributes.emplace_back("factors", list({}));
	m->attributes.emplace_back("class", cls);
	return m;
}

void save(const ValuePtr &value, const std::string &path) {
	Out out(path);
	out.raw("X\n", 2);
	out.i32(2);                                  // serialisation version
	out.i32((3 << 16) | (4 << 8) | 0);           // R_VERSION of the "writer": 3.4.0
	out.i32((2 << 16) | (3 << 8) | 0);           // minimal R version that can read it: 2.3.0
	out.item(*value);
	out.close();
}

}  // namespace Rds
