// bam_ingest.cpp -- see bam_ingest.h.
#include "bam_ingest.h"
#include "fast_inflate.h"
#include "../../../include/dropest_bgzf.h"
#include "../../../include/dropest_annotation.h"
#include <atomic>

#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <future>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <unordered_set>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>

static inline void bam_cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
	__builtin_ia32_pause();
#elif defined(__aarch64__)
	__asm__ __volatile__("yield");
#endif
}

namespace Estimation {
namespace BamProcessing {

namespace {

inline uint16_t le16(const uint8_t *p) { return uint16_t(p[0] | (p[1] << 8)); }
inline uint32_t le32(const uint8_t *p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }

struct RawBlock { const uint8_t *cdata = nullptr; size_t clen = 0; uint32_t isize = 0, crc = 0; };

// one BGZF block (SAMv1 §4.1): gzip member, FEXTRA with subfield 'B','C' = total block size - 1.  The file is MAPPED: a block
// is described in place (header walk only), the inflating workers read the compressed bytes straight from the page cache --
// a serial fread of every block into a vector of its own capped the reader at ~2 GB/s of BAM whatever the thread count.
bool read_block(const uint8_t *map, size_t size, size_t &at, RawBlock &b, const std::string &path) {
	if (at >= size) return false;
	if (size - at < 12) throw std::runtime_error("Not a BGZF/BAM file: " + path);
	const uint8_t *h = map + at;
	if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) throw std::runtime_error("Not a BGZF/BAM file: " + path);
	const uint16_t xlen = le16(h + 10);
	if (size - at < size_t(12) + xlen) throw std::runtime_error("Truncated BGZF header: " + path);
	const uint8_t *extra = h + 12;
	int bsize = -1;
	for (size_t o = 0; o + 4 <= xlen;) {
		const uint16_t slen = le16(extra + o + 2);
		if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = le16(extra + o + 4);
		o += 4u + slen;
	}
	if (bsize < 0) throw std::runtime_error("BGZF block without BC subfield: " + path);
	const long clen = long(bsize) - long(xlen) - 19;
	if (clen < 0) throw std::runtime_error("Corrupt BGZF block size: " + path);
	if (size - at < size_t(12) + xlen + size_t(clen) + 8) throw std::runtime_error("Truncated BGZF block: " + path);
	b.cdata = extra + xlen; b.clen = size_t(clen);
	const uint8_t *tail = b.cdata + clen;
	b.crc = le32(tail); b.isize = le32(tail + 4);
	at += size_t(12) + xlen + size_t(clen) + 8;
	return true;
}

void inflate_block(const RawBlock &b, uint8_t *out) {
	if (b.isize == 0) return;
	// the block decoder of fast_inflate.h first (about three times zlib 1.2.11's rate); a block it refuses -- and any block when
	// DROPEST_BAM_ZLIB is set -- goes through zlib below.  Either way CRC-32 and ISIZE of the block are checked.
	static const bool force_zlib = getenv("DROPEST_BAM_ZLIB") != nullptr;
	if (!force_zlib && fastinflate::inflate_raw(b.cdata, b.clen, out, b.isize)) {
		if (fastinflate::crc32(out, b.isize) != b.crc) throw std::runtime_error("Corrupt BGZF block (CRC)");
		return;
	}
	// one inflate state per thread, reset per block (inflateInit2 allocates its 40 KB window every time)
	struct State { z_stream zs; bool ready = false; ~State() { if (ready) inflateEnd(&zs); } };
	static thread_local State st;
	z_stream &zs = st.zs;
	if (!st.ready) {
		std::memset(&zs, 0, sizeof(zs));
		if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("zlib: inflateInit2 failed");
		st.ready = true;
	} else if (inflateReset2(&zs, -15) != Z_OK) throw std::runtime_error("zlib: inflateReset2 failed");
	zs.next_in = const_cast<Bytef *>(b.cdata); zs.avail_in = uInt(b.clen);
	zs.next_out = out; zs.avail_out = b.isize;
	const int rc = inflate(&zs, Z_FINISH);
	if (rc != Z_STREAM_END || zs.total_out != b.isize) throw std::runtime_error("Corrupt BGZF block (inflate)");
	if (crc32(crc32(0L, Z_NULL, 0), out, b.isize) != b.crc) throw std::runtime_error("Corrupt BGZF block (CRC)");
}

// Worker threads that live as long as the reader / controller: a window of records comes every few milliseconds at the rates
// this path is built for, and creating 64 threads per window costs more than parsing it.  run(fn) executes fn(worker) on every
// worker and returns when all are done; an exception of a worker is rethrown on the caller's thread.
class WorkerPool {
	std::vector<std::thread> threads;
	std::mutex m;
	std::condition_variable cv_go, cv_done;
	std::function<void(unsigned)> job;
	uint64_t generation = 0;
	unsigned pending = 0;
	bool stop = false;
	std::vector<std::string> errors;
public:
	explicit WorkerPool(unsigned n) : errors(n) {
		for (unsigned t = 0; t < n; ++t)
			threads.emplace_back([this, t] {
				uint64_t seen = 0;
				for (;;) {
					std::function<void(unsigned)> fn;
					{
						std::unique_lock<std::mutex> lk(m);
						cv_go.wait(lk, [&] { return stop || generation != seen; });
						if (stop) return;
						seen = generation; fn = job;
					}
					try { fn(t); } catch (const std::exception &e) { errors[t] = e.what(); } catch (...) { errors[t] = "unknown error"; }
					{ std::lock_guard<std::mutex> lk(m); if (--pending == 0) cv_done.notify_all(); }
				}
			});
	}
	~WorkerPool() {
		{ std::lock_guard<std::mutex> lk(m); stop = true; }
		cv_go.notify_all();
		for (auto &t : threads) t.join();
	}
	unsigned size() const { return unsigned(threads.size()); }
	void run(const std::function<void(unsigned)> &fn) {
		{
			std::lock_guard<std::mutex> lk(m);
			job = fn; pending = unsigned(threads.size()); ++generation;
			for (auto &e : errors) e.clear();
		}
		cv_go.notify_all();
		std::unique_lock<std::mutex> lk(m);
		cv_done.wait(lk, [&] { return pending == 0; });
		for (auto const &e : errors) if (!e.empty()) throw std::runtime_error(e);
	}
};

}  // namespace

struct BamReader::Impl {
	std::string path;
	const uint8_t *map = nullptr;                      // the whole file, mapped read-only
	size_t map_size = 0, map_at = 0;
	unsigned threads = 1;
	static constexpr size_t BATCH_BLOCKS = 512;        // <= 32 MB of BAM per batch (a window of ~2e5 records; the workers are persistent: a dispatch costs microseconds)
	std::unique_ptr<WorkerPool> pool;                  // inflate workers (only touched by the one batch loader running at a time)
	static constexpr size_t HEADROOM = 1 << 20;
	// Decompressed windows live in a few recycled buffers (a fresh 32 MB allocation per batch would spend more time in
	// page faults than the inflate takes).  A batch leaves HEADROOM bytes free in front: the unconsumed tail of the
	// previous window (a partial record) is copied there, so a new batch is adopted without moving it.
	struct Buf {
		std::unique_ptr<uint8_t[]> p; size_t cap = 0, size = 0;
		void need(size_t n) { if (n > cap) { p.reset(new uint8_t[n]); cap = n; } }
		// Record boundaries found by the loader (walk_records): starts of the COMPLETE records that begin in this batch (buffer offsets),
		// first_start = the first of them or tail_start, tail_start = where the unfinished last record begins (= size if none).
		std::vector<uint32_t> rec_off;
		bool walked = false, consumed = false;
		size_t first_start = 0, tail_start = 0;
	};
	Buf cur;                                           // window being parsed: bytes [pos, cur.size)
	std::vector<Buf> spare;                            // recycled buffers (under qm)
	size_t pos = 0;
	bool file_done = false;
	// Batches are loaded by ONE background thread, strictly one after the other (block discovery, the walk chain and the cut record carry
	// state from batch to batch), up to DEPTH ahead of the parser: with a single batch in flight every slow batch of either side -- the
	// hosts are shared -- stalled the other (215 ms of waiting in a 950 ms ingest whose loader and parser each needed ~700).
	static constexpr size_t DEPTH = 3;
	std::thread loader;
	std::mutex qm;
	std::condition_variable q_cv;
	std::deque<Buf> ready;
	bool loader_started = false, loader_finished = false, loader_stop = false;
	std::exception_ptr loader_error;
	void loader_main() {
		try {
			for (;;) {
				Buf b;
				{
					std::unique_lock<std::mutex> lk(qm);
					q_cv.wait(lk, [&] { return loader_stop || ready.size() < DEPTH; });
					if (loader_stop) break;
					b = take_spare();
				}
				Buf out = load_batch(std::move(b));
				const bool last = file_done;
				{ std::lock_guard<std::mutex> lk(qm); ready.push_back(std::move(out)); if (last) loader_finished = true; }
				q_cv.notify_all();
				if (last) break;
			}
		} catch (...) {
			{ std::lock_guard<std::mutex> lk(qm); loader_error = std::current_exception(); loader_finished = true; }
			q_cv.notify_all();
		}
	}
	bool next_batch(Buf &out) {   // false: the stream has ended (an error of the loader is rethrown here)
		if (!loader_started) { loader_started = true; loader = std::thread([this] { loader_main(); }); }
		std::unique_lock<std::mutex> lk(qm);
		q_cv.wait(lk, [&] { return !ready.empty() || loader_finished; });
		if (ready.empty()) { if (loader_error) std::rethrow_exception(loader_error); return false; }
		out = std::move(ready.front());
		ready.pop_front();
		lk.unlock();
		q_cv.notify_all();
		return true;
	}
	void stop_loader() {
		if (!loader_started) return;
		{ std::lock_guard<std::mutex> lk(qm); loader_stop = true; }
		q_cv.notify_all();
		if (loader.joinable()) loader.join();
	}
	std::vector<std::string> refs;
	std::string text;
	const uint8_t *bytes() const { return cur.p.get(); }

	// The walk over the records (every record's length says where the next one starts: a dependent chain) used to run on the caller's
	// thread over data the inflate workers had just written on other cores -- 4 ms per 32 MB window, the floor of the whole ingest
	// (profiles/NOTES_r03.md).  Now every worker, after inflating block i, waits for the position block i - 1's walker ended at, walks
	// ITS block while it is hot in its own cache and hands the position on; the loader of a batch knows where the last record of the
	// batch before was cut.  walk_state: what the next batch needs to know about that cut.
	bool loader_walks = getenv("DROPEST_BAM_CALLER_WALKS") == nullptr;
	struct WalkState { bool valid = false; size_t tail_len = 0; uint8_t tail[4] = {0, 0, 0, 0}; uint32_t tail_record = 0; } walk_state;
	std::vector<std::vector<uint32_t>> block_starts;
	double load_ms = 0, discover_ms = 0; size_t n_batches = 0, n_walked = 0;   // diagnostics (DROPEST_BAM_TRACE)
	size_t first_batch_compressed = 0;                           // bytes of the file the first batch covered (read-count estimate)
	Buf load_batch(Buf out) {
		const auto t_begin = std::chrono::steady_clock::now();
		std::vector<RawBlock> blocks;
		blocks.reserve(BATCH_BLOCKS);
		while (blocks.size() < BATCH_BLOCKS) {
			RawBlock b;
			if (!read_block(map, map_size, map_at, b, path)) { file_done = true; break; }
			blocks.push_back(std::move(b));
		}
		if (!n_batches) first_batch_compressed = map_at;
		discover_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
		std::vector<size_t> off(blocks.size() + 1, HEADROOM);
		for (size_t i = 0; i < blocks.size(); ++i) off[i + 1] = off[i] + blocks[i].isize;
		out.need(off.back());
		out.size = off.back();
		uint8_t *base = out.p.get();
		// (inflate + the walk chain stop scaling at about 16 workers -- the chain is serial -- and more of them only take cores from
		// the parsers, which do scale: 64 inflate workers measured a third of the rate of 16)
		if (!pool) pool.reset(new WorkerPool(std::max(1u, std::min(threads, 16u))));
		const unsigned nt = pool->size();
		std::atomic<size_t> next_block{0};            // blocks differ in cost: taken one at a time, in file order
		out.walked = out.consumed = false; out.rec_off.clear();
		// where this batch's first record starts (buffer offset), if the batch before told us; batch 0 holds the header: walked below
		const size_t NB = blocks.size();
		bool chain = loader_walks && n_batches > 0 && walk_state.valid && NB > 0;
		size_t p0 = HEADROOM;
		bool size_from_new = false;                   // the cut went through the length field: its missing bytes open this batch
		if (chain) {
			if (walk_state.tail_len >= 4) p0 = HEADROOM + 4 + size_t(walk_state.tail_record) - walk_state.tail_len;
			else if (walk_state.tail_len > 0) size_from_new = true;
		}
		std::unique_ptr<std::atomic<uint64_t>[]> carry;   // carry[i] = 1 + position the walker of block i starts at (0: not known yet)
		std::atomic<bool> abort_walk{false}, corrupt{false};
		if (chain) {
			carry.reset(new std::atomic<uint64_t>[NB + 1]);
			for (size_t i = 0; i <= NB; ++i) carry[i].store(0, std::memory_order_relaxed);
			if (!size_from_new) carry[0].store(uint64_t(p0) + 1, std::memory_order_release);
			if (block_starts.size() < NB) block_starts.resize(NB);
		}
		try {
			pool->run([&](unsigned) {
				for (size_t i; (i = next_block.fetch_add(1)) < NB;) {
					try { inflate_block(blocks[i], base + off[i]); } catch (...) { abort_walk = true; throw; }
					if (!chain) continue;
					if (i == 0 && size_from_new) {
						const size_t missing = 4 - walk_state.tail_len;
						if (off[1] - off[0] < missing) { abort_walk = true; continue; }   // (a block shorter than the rest of a length field: the caller walks this batch)
						uint8_t sz[4];
						std::memcpy(sz, walk_state.tail, walk_state.tail_len);
						std::memcpy(sz + walk_state.tail_len, base + HEADROOM, missing);
						carry[0].store(uint64_t(HEADROOM + 4 + size_t(le32(sz)) - walk_state.tail_len) + 1, std::memory_order_release);
					}
					uint64_t c;
					for (unsigned spins = 0; (c = carry[i].load(std::memory_order_acquire)) == 0; ++spins) {
						if (abort_walk.load(std::memory_order_relaxed)) break;
						if (spins < 2000) bam_cpu_relax(); else std::this_thread::yield();   // (the walker before us may have lost its core)
					}
					if (c == 0) continue;
					size_t P = size_t(c - 1);
					std::vector<uint32_t> &st = block_starts[i];
					st.clear();
					const size_t end_i = off[i + 1];
					while (P + 4 <= end_i) {
						const uint32_t bs = le32(base + P);
						if (bs < 32 || P > 0xFFFFFFF0ull) { corrupt = true; abort_walk = true; break; }
						st.push_back(uint32_t(P));
						P += 4 + size_t(bs);
					}
					carry[i + 1].store(uint64_t(P) + 1, std::memory_order_release);
				}
			});
		} catch (const std::exception &e) { throw std::runtime_error(std::string(e.what()) + ": " + path); }
		(void)nt;
		if (corrupt) throw std::runtime_error("Corrupt BAM record: " + path);
		if (chain && !abort_walk) {
			size_t total = 0;
			for (size_t i = 0; i < NB; ++i) total += block_starts[i].size();
			out.rec_off.resize(total);
			size_t at = 0;
			for (size_t i = 0; i < NB; ++i) { if (!block_starts[i].empty()) std::memcpy(out.rec_off.data() + at, block_starts[i].data(), block_starts[i].size() * 4); at += block_starts[i].size(); }
			finish_walk(out, size_t(carry[NB].load() - 1));
		} else if (loader_walks && n_batches == 0 && NB > 0) walk_first_batch(out);
		else walk_state.valid = false;                 // (the caller walks; a later batch cannot know where its records start)
		load_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); ++n_batches;
		return out;
	}
	// After the starts are known: drop the ones whose record does not end inside the batch (they are the tail), remember the cut.
	void finish_walk(Buf &out, size_t next_start) {
		const uint8_t *base = out.p.get();
		size_t tail_start = next_start;              // the position after the last started record (its length field may be cut: < 4 bytes left)
		while (!out.rec_off.empty()) {
			const size_t s = out.rec_off.back();
			if (s + 4 + size_t(le32(base + s)) <= out.size) break;
			tail_start = s;
			out.rec_off.pop_back();
		}
		if (tail_start > out.size) {                 // a record that runs past this whole batch: the caller's path handles such giants
			out.rec_off.clear(); out.walked = false; walk_state.valid = false;
			return;
		}
		out.walked = true; out.tail_start = tail_start; ++n_walked;
		out.first_start = !out.rec_off.empty() ? size_t(out.rec_off.front()) : tail_start;
		walk_state.valid = true;
		walk_state.tail_len = out.size - tail_start;
		if (walk_state.tail_len >= 4) walk_state.tail_record = le32(base + tail_start);
		else std::memcpy(walk_state.tail, base + tail_start, walk_state.tail_len);
	}
	// Batch 0 starts with the header (magic, text, reference names): parsed here only to find the first record, then one serial walk.
	void walk_first_batch(Buf &out) {
		walk_state.valid = false;
		const uint8_t *base = out.p.get();
		size_t o = HEADROOM;
		auto have = [&](size_t n) { return o + n <= out.size; };
		if (!have(12) || std::memcmp(base + o, "BAM\1", 4) != 0) return;
		const size_t l_text = le32(base + o + 4);
		if (!have(12 + l_text)) return;
		const uint32_t n_ref = le32(base + o + 8 + l_text);
		o += 12 + l_text;
		for (uint32_t r = 0; r < n_ref; ++r) {
			if (!have(4)) return;
			const size_t l_name = le32(base + o);
			if (!have(8 + l_name)) return;
			o += 8 + l_name;
		}
		size_t P = o;
		while (P + 4 <= out.size) {
			const uint32_t bs = le32(base + P);
			if (bs < 32 || P > 0xFFFFFFF0ull) throw std::runtime_error("Corrupt BAM record: " + path);
			out.rec_off.push_back(uint32_t(P));
			P += 4 + size_t(bs);
		}
		finish_walk(out, P);
		if (out.walked && out.rec_off.empty()) out.first_start = out.tail_start;
	}
	Buf take_spare() {
		if (spare.empty()) return Buf();
		Buf b = std::move(spare.back());
		spare.pop_back();
		return b;
	}
	// makes at least `need` bytes available at bytes()[pos..]; false if the stream ends first
	bool ensure(size_t need) {
		while (cur.size - pos < need) {
			Buf next;
			if (!next_batch(next)) return false;
			if (next.size <= HEADROOM) {   // an empty batch (the end of the file fell on a batch boundary): the next call ends the stream
				if (next.cap) { std::lock_guard<std::mutex> lk(qm); if (spare.size() < DEPTH + 2) spare.push_back(std::move(next)); }
				continue;
			}
			const size_t tail = cur.size - pos;
			if (tail <= HEADROOM && next.size >= HEADROOM) {
				if (tail) std::memcpy(next.p.get() + HEADROOM - tail, cur.p.get() + pos, tail);
				std::swap(cur, next);
				pos = HEADROOM - tail;
			} else {   // a tail longer than the headroom (one enormous record): concatenate into a larger buffer
				Buf big;
				const size_t payload = next.size > HEADROOM ? next.size - HEADROOM : 0;
				big.need(tail + payload);
				if (tail) std::memcpy(big.p.get(), cur.p.get() + pos, tail);
				if (payload) std::memcpy(big.p.get() + tail, next.p.get() + HEADROOM, payload);
				big.size = tail + payload;
				std::swap(cur, big);
				pos = 0;
				if (big.cap) { std::lock_guard<std::mutex> lk(qm); spare.push_back(std::move(big)); }
			}
			if (next.cap) { std::lock_guard<std::mutex> lk(qm); if (spare.size() < DEPTH + 2) spare.push_back(std::move(next)); }   // the old window, recycled
		}
		return true;
	}
};

BamReader::BamReader(const std::string &path, unsigned threads) : impl(new Impl()) {
	impl->path = path;
	impl->threads = threads ? threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
	{
		const int fd = open(path.c_str(), O_RDONLY);
		struct stat sb;
		if (fd < 0 || fstat(fd, &sb) != 0) { if (fd >= 0) close(fd); delete impl; throw std::runtime_error("Can't open BAM file: " + path); }
		impl->map_size = size_t(sb.st_size);
		void *m = impl->map_size ? mmap(nullptr, impl->map_size, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
		close(fd);
		if (impl->map_size && m == MAP_FAILED) { delete impl; throw std::runtime_error("Can't open BAM file: " + path); }
		impl->map = static_cast<const uint8_t *>(m);
		if (m) (void)madvise(m, impl->map_size, MADV_SEQUENTIAL);
	}
	try {
		Impl &m = *impl;
		if (!m.ensure(12) || std::memcmp(m.bytes() + m.pos, "BAM\1", 4) != 0) throw std::runtime_error("Can't open BAM file: " + path);
		const uint32_t l_text = le32(m.bytes() + m.pos + 4);
		if (!m.ensure(12 + size_t(l_text))) throw std::runtime_error("Truncated BAM header: " + path);
		m.text.assign(reinterpret_cast<const char *>(m.bytes() + m.pos + 8), l_text);
		const uint32_t n_ref = le32(m.bytes() + m.pos + 8 + l_text);
		m.pos += 12 + size_t(l_text);
		for (uint32_t r = 0; r < n_ref; ++r) {
			if (!m.ensure(4)) throw std::runtime_error("Truncated BAM header: " + path);
			const uint32_t l_name = le32(m.bytes() + m.pos);
			if (!m.ensure(8 + size_t(l_name))) throw std::runtime_error("Truncated BAM header: " + path);
			m.refs.emplace_back(reinterpret_cast<const char *>(m.bytes() + m.pos + 4), l_name ? l_name - 1 : 0);
			m.pos += 8 + size_t(l_name);
		}
	} catch (...) {
		impl->stop_loader();
		if (impl->map) munmap(const_cast<uint8_t *>(impl->map), impl->map_size);
		delete impl; throw;
	}
}

BamReader::~BamReader() {
	impl->stop_loader();
	if (getenv("DROPEST_BAM_TRACE")) std::fprintf(stderr, "[bam] %zu batches (%zu with the record boundaries from the loader), load %.1f ms (block discovery %.1f ms), %u inflate threads\n", impl->n_batches, impl->n_walked, impl->load_ms, impl->discover_ms, impl->threads);
	if (impl->map) munmap(const_cast<uint8_t *>(impl->map), impl->map_size);
	delete impl;
}

const std::vector<std::string> &BamReader::reference_names() const { return impl->refs; }
double BamReader::file_over_first_batch() const { return impl->first_batch_compressed ? double(impl->map_size) / double(impl->first_batch_compressed) : 1.0; }
const std::string &BamReader::header_text() const { return impl->text; }

void BamReader::parse_record(const uint8_t *at, BamRecord &rec) {
	const uint32_t block_size = le32(at);
	const uint8_t *p = at + 4;
	rec.ref_id = int32_t(le32(p));
	rec.position = int32_t(le32(p + 4));
	const uint32_t l_read_name = p[8];
	const uint32_t n_cigar = le16(p + 12);
	rec.flag = le16(p + 14);
	const uint32_t l_seq = le32(p + 16);
	const size_t fixed = 32, name_end = fixed + l_read_name;
	const size_t aux = name_end + size_t(n_cigar) * 4 + (size_t(l_seq) + 1) / 2 + l_seq;
	if (block_size < 32 || aux > block_size) throw std::runtime_error("Corrupt BAM record");
	int64_t ref_len = 0;                                                 // CIGAR ops: MIDNSHP=X -> 0..8 (SAMv1 §4.2)
	for (uint32_t k = 0; k < n_cigar; ++k) {
		const uint32_t op = le32(p + name_end + size_t(k) * 4);
		const uint32_t kind = op & 0xF;
		if (kind == 0 || kind == 2 || kind == 3 || kind == 7 || kind == 8) ref_len += op >> 4;
	}
	rec.end_position = int32_t(int64_t(rec.position) + ref_len);
	rec.name_view = std::string_view(reinterpret_cast<const char *>(p + fixed), l_read_name ? l_read_name - 1 : 0);
	rec.tags = p + aux; rec.tags_size = block_size - aux;
}

bool BamReader::next(BamRecord &rec) {
	Impl &m = *impl;
	if (!m.ensure(4)) return false;
	const uint32_t block_size = le32(m.bytes() + m.pos);
	if (block_size < 32) throw std::runtime_error("Corrupt BAM record: " + m.path);
	if (!m.ensure(4 + size_t(block_size))) throw std::runtime_error("Truncated BAM record: " + m.path);
	parse_record(m.bytes() + m.pos, rec);
	rec.name.assign(rec.name_view);
	m.pos += 4 + size_t(block_size);
	return true;
}

bool BamReader::next_window(const uint8_t *&data, std::vector<uint32_t> &offsets) {
	Impl &m = *impl;
	offsets.clear();
	if (m.cur.walked && m.cur.consumed) {   // the window handed out last time was the whole batch: what is left is the cut record
		const size_t tail = m.cur.size - m.pos;
		if (!m.ensure(tail + 1)) { if (tail) throw std::runtime_error("Truncated BAM record: " + m.path); return false; }
	} else if (!m.ensure(4)) return false;
	if (m.cur.walked && !m.cur.consumed) {
		// records of this batch as its loader found them; in front of them the record the last batch was cut in (its head was copied
		// to just before this batch's first byte, its end is where this batch's first own record starts)
		data = m.bytes() + m.pos;
		if (m.pos < m.cur.first_start) {
			const uint32_t bs = le32(data);
			if (bs < 32 || m.pos + 4 + size_t(bs) != m.cur.first_start) throw std::runtime_error("Corrupt BAM record: " + m.path);
			offsets.push_back(0);
		}
		offsets.reserve(offsets.size() + m.cur.rec_off.size());
		for (uint32_t s : m.cur.rec_off) offsets.push_back(uint32_t(s - m.pos));
		m.pos = m.cur.tail_start;
		m.cur.consumed = true;
		if (!offsets.empty()) return true;
		return next_window(data, offsets);       // (a batch without a single complete record: go on with the next one)
	}
	// (after a consumed batch only tail + 1 bytes are known to be there: the length field itself may be cut by the end of the file)
	if (!m.ensure(4)) { if (m.cur.size > m.pos) throw std::runtime_error("Truncated BAM record: " + m.path); return false; }
	const uint32_t first_size = le32(m.bytes() + m.pos);
	if (first_size < 32) throw std::runtime_error("Corrupt BAM record: " + m.path);
	if (!m.ensure(4 + size_t(first_size))) throw std::runtime_error("Truncated BAM record: " + m.path);
	data = m.bytes() + m.pos;
	const size_t avail = m.cur.size - m.pos;
	size_t o = 0;
	while (o + 4 <= avail) {
		const uint32_t bs = le32(data + o);
		if (bs < 32) throw std::runtime_error("Corrupt BAM record: " + m.path);
		if (o + 4 + size_t(bs) > avail || o > 0xFFFFFFF0ull) break;
		offsets.push_back(uint32_t(o));
		o += 4 + size_t(bs);
	}
	m.pos += o;
	return true;
}

bool BamRecord::get_string_tag(const std::string &tag, std::string_view &value, char *type_out) const {
	if (tag.size() != 2) return false;
	size_t o = 0;
	while (o + 3 <= tags_size) {
		const char t0 = char(tags[o]), t1 = char(tags[o + 1]), type = char(tags[o + 2]);
		o += 3;
		size_t len = 0;
		bool text = false;
		switch (type) {
			case 'A': case 'c': case 'C': len = 1; break;
			case 's': case 'S': len = 2; break;
			case 'i': case 'I': case 'f': len = 4; break;
			case 'Z': case 'H': { const void *e = std::memchr(tags + o, 0, tags_size - o); if (!e) return false; len = size_t(static_cast<const uint8_t *>(e) - (tags + o)) + 1; text = true; break; }
			case 'B': {
				if (o + 5 > tags_size) return false;
				const char sub = char(tags[o]);
				const uint32_t cnt = le32(tags + o + 1);
				const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
				len = 5 + size_t(cnt) * w;
				break;
			}
			default: return false;   // unknown type: cannot skip safely
		}
		if (o + len > tags_size) return false;
		if (t0 == tag[0] && t1 == tag[1]) {
			if (type_out) *type_out = type;
			if (text) { value = std::string_view(reinterpret_cast<const char *>(tags + o), len - 1); return true; }
			if (type == 'A') { value = std::string_view(reinterpret_cast<const char *>(tags + o), 1); return true; }
			// numeric tag where a string is asked for: treated as absent; said once, the counters tell the rest
			static std::atomic<bool> warned{false};
			if (!warned.exchange(true))
				std::fprintf(stderr, "WARNING: BAM tag %c%c has the numeric type '%c' where a string (Z) is expected; records with it are handled as if the tag were absent\n", t0, t1, type);
			return false;
		}
		o += len;
	}
	return false;
}

void BamRecord::get_string_tags(const uint16_t *wanted, int n_wanted, std::string_view *values, bool *found) const {
	for (int k = 0; k < n_wanted; ++k) found[k] = false;
	bool closed[16] = {false};   // a tag met with a numeric type counts as absent, and later records of the same name are not looked at
	size_t o = 0;
	while (o + 3 <= tags_size) {
		const uint16_t name = uint16_t(tags[o] | (tags[o + 1] << 8));
		const char type = char(tags[o + 2]);
		o += 3;
		size_t len = 0;
		bool text = false;
		switch (type) {
			case 'A': case 'c': case 'C': len = 1; break;
			case 's': case 'S': len = 2; break;
			case 'i': case 'I': case 'f': len = 4; break;
			case 'Z': case 'H': { const void *e = std::memchr(tags + o, 0, tags_size - o); if (!e) return; len = size_t(static_cast<const uint8_t *>(e) - (tags + o)) + 1; text = true; break; }
			case 'B': {
				if (o + 5 > tags_size) return;
				const char sub = char(tags[o]);
				const uint32_t cnt = le32(tags + o + 1);
				const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
				len = 5 + size_t(cnt) * w;
				break;
			}
			default: return;   // unknown type: cannot skip safely
		}
		if (o + len > tags_size) return;
		for (int k = 0; k < n_wanted && k < 16; ++k) {
			if (wanted[k] != name || !wanted[k] || found[k] || closed[k]) continue;
			if (text) { values[k] = std::string_view(reinterpret_cast<const char *>(tags + o), len - 1); found[k] = true; }
			else if (type == 'A') { values[k] = std::string_view(reinterpret_cast<const char *>(tags + o), 1); found[k] = true; }
			else {
				closed[k] = true;
				static std::atomic<bool> warned{false};
				if (!warned.exchange(true))
					std::fprintf(stderr, "WARNING: BAM tag %c%c has the numeric type '%c' where a string (Z) is expected; records with it are handled as if the tag were absent\n", char(name & 0xFF), char(name >> 8), type);
			}
		}
		o += len;
	}
}

bool BamRecord::get_string_tag(const std::string &tag, std::string &value, char *type_out) const {
	std::string_view v;
	if (!get_string_tag(tag, v, type_out)) return false;
	value.assign(v);
	return true;
}

namespace {
std::mutex &decoder_cache_mutex() { static std::mutex m; return m; }
std::unordered_map<int, dropest_bam_decoder *> &decoder_cache() { static auto *c = new std::unordered_map<int, dropest_bam_decoder *>(); return *c; }   // (never destroyed: the HIP runtime may be gone by then)
}  // namespace

namespace {
// The decoders of the files just read go back on a helper thread (several GB of device buffers and up to 512 MB of pinned staging: ~40 ms of
// hipFree / hipHostFree that the container's passes need not wait for).  Joined before the next file list, by release_device_decoders(), and
// at exit before the HIP runtime goes down (std::atexit handlers run in reverse order of registration, and the runtime registered first).
std::thread *&release_thread() { static std::thread *t = nullptr; return t; }
void join_release_thread() {
	std::thread *t = nullptr;
	{ std::lock_guard<std::mutex> lk(decoder_cache_mutex()); t = release_thread(); release_thread() = nullptr; }
	if (t) { if (t->joinable()) t->join(); delete t; }
}
void release_decoders_in_background() {
	join_release_thread();
	std::lock_guard<std::mutex> lk(decoder_cache_mutex());
	if (decoder_cache().empty()) return;
	std::vector<dropest_bam_decoder *> gone;
	for (auto &kv : decoder_cache()) gone.push_back(kv.second);
	decoder_cache().clear();
	static const bool at_exit = (std::atexit(join_release_thread), true);
	(void)at_exit;
	release_thread() = new std::thread([gone] { for (dropest_bam_decoder *d : gone) dropest_bam_decoder_destroy(d); });
}
}  // namespace

void BamController::release_device_decoders() {
	join_release_thread();
	std::lock_guard<std::mutex> lk(decoder_cache_mutex());
	for (auto &kv : decoder_cache()) dropest_bam_decoder_destroy(kv.second);
	decoder_cache().clear();
}

BamController::BamController(const BamTags &tags, bool filled_bam, const std::string &read_param_filenames, const std::string &gtf_path,
                             bool gene_in_chromosome_name, int min_barcode_phred, unsigned threads)
	: _tags(tags), _filled_bam(filled_bam), _gene_in_chromosome_name(gene_in_chromosome_name), _min_barcode_phred(min_barcode_phred),
	  _threads(threads) {
	if (!gtf_path.empty()) _genes = Tools::GeneAnnotation::RefGenesContainer(gtf_path);
	if (!filled_bam && !read_param_filenames.empty()) load_read_params(read_param_filenames);   // BamController::get_parser (:117-130)
	if (!_tags.read_type.empty() && _tags.intronic_read_value.empty())
		throw std::runtime_error("You have to specify tag values to be able to parse info about read types (see conf_desc.xml \"Estimation/BamTags/Type/\")");
}

// ReadMapParamsParser::init (ReadMapParamsParser.cpp:50-108): gzip text files, one "name cb umi cb_quality umi_quality" row per
// read (ReadParameters::parse_from_string, ReadParameters.cpp:58-78); malformed rows and repeated names are reported and skipped
void BamController::load_read_params(const std::string &filenames) {
	_params_from_files = true;
	std::istringstream names(filenames);
	std::string name;
	const int quality_offset = 33;
	while (names >> name) {
		gzFile f = gzopen(name.c_str(), "rb");
		if (!f) throw std::runtime_error("Can't open file with read parameters'" + name + "'");
		std::string row;
		char buf[1 << 16];
		auto handle_row = [&]() {
			if (row.empty()) return;
			std::string parts[5];
			size_t start = 0;
			bool ok = true;
			for (int i = 0; i < 4 && ok; ++i) {
				const size_t end = row.find(' ', start);
				if (end == std::string::npos) { ok = false; break; }
				parts[i] = row.substr(start, end - start);
				start = end + 1;
			}
			if (!ok) return;                                            // "can't parse read parameters from string": skipped
			parts[4] = row.substr(start);
			if (!parts[0].empty() && parts[0][0] == '@') parts[0] = parts[0].substr(1);
			if (parts[1].empty() || parts[2].empty()) return;           // the ReadParameters constructor throws: skipped
			bool pass = true;                                           // ReadParameters::check_quality (:118-136)
			if (_min_barcode_phred > quality_offset) {
				for (char q : parts[3]) pass &= q >= char(_min_barcode_phred);
				for (char q : parts[4]) pass &= q >= char(_min_barcode_phred);
			}
			_read_params.emplace(parts[0], ReadParams{parts[1], parts[2], parts[4], pass});   // a repeated name keeps the first row
		};
		while (gzgets(f, buf, int(sizeof(buf)))) {
			row += buf;
			if (!row.empty() && row.back() == '\n') { row.pop_back(); if (!row.empty() && row.back() == '\r') row.pop_back(); handle_row(); row.clear(); }
		}
		handle_row();
		gzclose(f);
	}
}

void BamController::parse_bam_files(const std::vector<std::string> &bam_files, CellsDataContainer &container) {
	const int quality_offset = 33;                                       // Tools::ReadParameters::quality_offset
	enum : uint8_t { OK = 0, SKIP, CANT_PARSE_NO_COUNT, CANT_PARSE, LOW_QUALITY };
	struct Parsed {
		CellsDataContainer::ParsedRead r; uint8_t status; std::string gene; /* -g: the annotation's answer (owned) */
		std::string_view name;                  // -r: the read name, looked up in file order by the caller's thread
		std::string p_cb, p_umi, p_quality;     // -r: the served parameters (owned: the map entry is erased)
	};
	const unsigned nthreads = _threads ? _threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
	const auto t_parse = std::chrono::steady_clock::now();
	struct SayAll { std::chrono::steady_clock::time_point t; ~SayAll() { if (getenv("DROPEST_BAM_TRACE")) std::fprintf(stderr, "[bam] parse_bam_files: %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count()); } } say_all{t_parse};
	join_release_thread();
	for (auto const &bam_name : bam_files) {
		std::unique_ptr<BamReader> reader_p;       // the host reader: opened only when it is the one that reads (its loader thread inflates ahead from the start)
		std::vector<std::string> refs;
		const uint8_t *data = nullptr;
		std::vector<uint32_t> offsets;
		std::vector<Parsed> parsed;
		auto tag16 = [](const std::string &t) { return t.size() == 2 ? uint16_t(uint8_t(t[0]) | (uint8_t(t[1]) << 8)) : uint16_t(0); };
		const uint16_t wanted_tags[6] = {tag16(_tags.cell_barcode), tag16(_tags.umi), tag16(_tags.cell_barcode_quality), tag16(_tags.umi_quality),
		                                 tag16(_tags.gene), tag16(_tags.read_type)};
		// one record -> Parsed; runs on the worker threads (reads `data`, writes only its own slot)
		auto parse_one = [&](const uint8_t *at, Parsed &out) {
			BamRecord al;
			BamReader::parse_record(at, al);
			CellsDataContainer::ParsedRead &r = out.r;
			r = CellsDataContainer::ParsedRead();
			if (!al.is_mapped() || !al.is_primary()) { out.status = SKIP; return; }            // BamController.cpp:87-88
			if (al.ref_id < 0 || size_t(al.ref_id) >= refs.size()) { out.status = CANT_PARSE_NO_COUNT; return; }   // :90-104
			r.ref_id = al.ref_id;
			const std::string &chr_name = refs[size_t(al.ref_id)];
			bool pass_quality = true;
			// every tag the record may be asked for, in one walk over its aux data
			enum { T_CB, T_UMI, T_CBQ, T_UMIQ, T_GENE, T_TYPE, T_N };
			std::string_view tv[T_N]; bool tf[T_N];
			al.get_string_tags(wanted_tags, T_N, tv, tf);
			if (_filled_bam) {                                            // FilledBamParamsParser.cpp:12-40
				std::string_view cbq, umiq;
				if (!tf[T_CB] || !tf[T_UMI]) { out.status = CANT_PARSE; return; }
				r.cb = tv[T_CB]; r.umi = tv[T_UMI];
				if (tf[T_CBQ]) cbq = tv[T_CBQ];
				if (tf[T_UMIQ]) umiq = tv[T_UMIQ];
				if (r.cb.empty() || r.umi.empty()) { out.status = CANT_PARSE; return; }       // ReadParameters ctor throws -> false
				if (_min_barcode_phred > quality_offset) {                 // ReadParameters::check_quality (:118-136)
					for (char q : cbq) pass_quality &= q >= char(_min_barcode_phred);
					for (char q : umiq) pass_quality &= q >= char(_min_barcode_phred);
				}
				r.umi_quality_length = uint32_t(umiq.size());
				r.umi_quality = umiq;
			} else if (_params_from_files) {                              // ReadMapParamsParser: resolved below, in file order
				out.name = al.name_view;
			} else {                                                      // ReadParamsParser.cpp:20-33: "id!CB#UMI"
				const std::string_view name = al.name_view;
				const size_t up = name.rfind('#');
				const size_t cp = up == std::string_view::npos ? std::string_view::npos : name.rfind('!', up);
				if (up == std::string_view::npos || cp == std::string_view::npos) { out.status = CANT_PARSE; return; }
				r.cb = name.substr(cp + 1, up - cp - 1); r.umi = name.substr(up + 1);
				if (r.cb.empty() || r.umi.empty()) { out.status = CANT_PARSE; return; }
				// parse_encoded_id builds ReadParameters(cb, umi, "", "") = min_phred_score 0: always passes (ReadParameters.cpp:42-56)
			}
			if (!pass_quality) { out.status = LOW_QUALITY; return; }
			// get_gene (ReadParamsParser.cpp:36-65) + parse_read_type (:67-90)
			UMI::Mark mark;
			if (_gene_in_chromosome_name) {
				r.gene = chr_name;
				if (!chr_name.empty()) mark.add(UMI::Mark::HAS_EXONS);
			} else if (!_genes.is_empty()) {                              // get_gene_from_reference (:92-151)
				int bits;
				try { bits = _genes.gene_of_alignment(chr_name, size_t(al.position), size_t(al.end_position), out.gene); }
				catch (const Tools::GeneAnnotation::RefGenesContainer::ChrNotFoundException &) { out.status = CANT_PARSE; return; }   // BamController.cpp:153-161
				if (bits & 1) mark.add(UMI::Mark::HAS_NOT_ANNOTATED);
				if (bits & 2) mark.add(UMI::Mark::HAS_EXONS);
				if (bits & 4) mark.add(UMI::Mark::HAS_INTRONS);
				r.gene = out.gene;
			} else if (!tf[T_GENE]) {
				r.gene = std::string_view();
				mark.add(UMI::Mark::HAS_NOT_ANNOTATED);
			} else {
				r.gene = tv[T_GENE];
				std::string_view read_type;
				if (tf[T_TYPE]) read_type = tv[T_TYPE];
				if (_tags.read_type.empty() || !tf[T_TYPE]) mark.add(UMI::Mark::HAS_EXONS);
				else if (read_type == _tags.intronic_read_value) mark.add(UMI::Mark::HAS_INTRONS);
				else if (!_tags.intergenic_read_value.empty() && read_type == _tags.intergenic_read_value) mark.add(UMI::Mark::HAS_NOT_ANNOTATED);
				else mark.add(UMI::Mark::HAS_EXONS);
			}
			r.mark = uint8_t(mark.bits());
			if (_params_from_files) r.cb_code = r.umi_code = 0;
			else {
				if (!CellsDataContainer::pack_code(r.cb, r.cb_code)) r.cb_code = 0;
				if (!CellsDataContainer::pack_code(r.umi, r.umi_code)) r.umi_code = 0;
			}
			r.gene_hash = CellsDataContainer::hash_name(r.gene);
			if (!r.gene.empty()) r.gene_id = container.lookup_gene(r.gene_hash, r.gene);   // dictionary is read-only while the workers run
			out.status = OK;
		};
		using clk = std::chrono::steady_clock;
		auto since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
		// ---- fast path: the workers write packed records, the caller's thread only resolves what is NEW ----
		// A record needs the caller's thread when it brings something the dictionaries have not seen: a gene name that is not in the
		// gene dictionary yet, a chromosome no counted read touched before, a barcode / UMI that does not pack into a 2-bit code (N).
		// After the first windows that is a handful of records per million.  Everything else -- parsing, packing, gene look-up,
		// the chromosome index, the compaction of the accepted records -- runs on the workers; the container receives whole arrays
		// (CellsDataContainer::add_records_packed).  Order of every dictionary operation = file order, as add_record read by read.
		struct Need { uint32_t idx; uint8_t what; int32_t ref; std::string_view cb, umi, gene; uint64_t gene_hash; std::string gene_owned; /* -g: the annotation's answer lives in the worker's scratch record */ };
		enum : uint8_t { NEED_CB = 1, NEED_UMI = 2, NEED_GENE = 4, NEED_CHR = 8 };
		WorkerPool workers(nthreads);
		const unsigned NT = workers.size();
		std::vector<std::vector<Need>> needs(NT);
		// every worker writes the records it accepts densely from the start of ITS range of the window (o_*[n t / NT ...]): the container
		// then takes one call per worker, in worker order = file order -- no second pass that closes the gaps (it was 170 ms of a 930 ms ingest)
		std::vector<uint64_t> o_cb, o_umi;
		std::vector<uint32_t> o_gene, o_aux;
		// UMI quality strings (UQ tags): while every gene-bearing read brings one of the same length, the window still goes in bulk -- every worker
		// keeps where the strings of its records are and, once the length is known for the whole window, copies them into one row per read
		// (CellsDataContainer::add_records_packed with rows); reads of different lengths, or a length other than the container's so far, leave the
		// window to the record-by-record path (UMI::add_read's length check, UMI.cpp:26-28, is per molecule there)
		constexpr uint32_t NO_QL = 0xFFFFFFFFu;
		struct Tally { size_t total = 0, cant = 0, low = 0, ok = 0; uint32_t ql = NO_QL; bool mixed = false; char pad[64]; };
		std::vector<Tally> tally(NT);
		std::vector<CellsDataContainer::PackedRun> runs;
		std::vector<const char *> o_qp;
		size_t est_reads_host = 0;
		std::vector<uint8_t> o_qual;
		std::vector<const uint8_t *> q_rows;
		double fw_ms[3] = {0, 0, 0};   // DROPEST_BAM_TRACE: parse + pack on the workers, new dictionary entries on this thread, container
		auto fast_window = [&](size_t n) -> bool {
			auto t_phase = clk::now();
			auto phase = [&](int k) { fw_ms[k] += since(t_phase); t_phase = clk::now(); };
			if (o_cb.size() < n) { o_cb.resize(n); o_umi.resize(n); o_gene.resize(n); o_aux.resize(n); o_qp.resize(n); }   // (never shrunk: no refill per window)
			workers.run([&](unsigned t) {
				Parsed tmp;
				std::vector<Need> &mine = needs[t];
				mine.clear();
				Tally tl;
				size_t at = n * t / NT;
				for (size_t i = n * t / NT; i < n * (t + 1) / NT; ++i) {
					parse_one(data + offsets[i], tmp);
					switch (tmp.status) {
						case SKIP: continue;
						case CANT_PARSE_NO_COUNT: ++tl.cant; continue;
						case CANT_PARSE: ++tl.total; ++tl.cant; continue;
						case LOW_QUALITY: ++tl.total; ++tl.low; continue;
						default: break;
					}
					++tl.total; ++tl.ok;
					const CellsDataContainer::ParsedRead &r = tmp.r;
					const bool has_gene = !r.gene.empty();
					o_qp[at] = has_gene ? r.umi_quality.data() : nullptr;
					if (has_gene) { if (tl.ql == NO_QL) tl.ql = r.umi_quality_length; else if (tl.ql != r.umi_quality_length) tl.mixed = true; }
					uint8_t what = 0;
					o_cb[at] = r.cb_code; if (!r.cb_code) what |= NEED_CB;
					if (has_gene) {
						o_umi[at] = r.umi_code; if (!r.umi_code) what |= NEED_UMI;
						if (r.gene_id >= 0) o_gene[at] = uint32_t(r.gene_id); else what |= NEED_GENE;
					} else { o_umi[at] = 1; o_gene[at] = DROPEST_NO_GENE; }
					const bool touches = !has_gene || (r.mark & (UMI::Mark::HAS_EXONS | UMI::Mark::HAS_INTRONS));
					uint32_t aux = uint32_t(r.mark) << 16;
					if (touches) {
						const int chr = container.chromosome_of_ref(r.ref_id);   // (the dictionaries are read-only while the workers run)
						if (chr >= 0) aux |= uint32_t(chr); else what |= NEED_CHR;
					}
					o_aux[at] = aux;
					if (what) mine.push_back(Need{uint32_t(at), what, r.ref_id, r.cb, r.umi, r.gene, r.gene_hash, (what & NEED_GENE) && !_genes.is_empty() ? std::string(r.gene) : std::string()});
					++at;
				}
				tally[t] = tl;
			});
			phase(0);
			uint32_t ql = NO_QL;
			for (unsigned t = 0; t < NT; ++t) {
				if (tally[t].mixed) return false;
				if (tally[t].ql == NO_QL) continue;
				if (ql == NO_QL) ql = tally[t].ql; else if (ql != tally[t].ql) return false;
			}
			const bool with_quality = ql != NO_QL && ql > 0;
			if (with_quality ? !container.bulk_ingest_possible_with_quality(ql) : !container.bulk_ingest_possible()) return false;
			// in file order: what the dictionaries have not seen (per record: barcode, then UMI, gene, chromosome -- the order of add_record)
			for (unsigned t = 0; t < NT; ++t)
				for (const Need &nd : needs[t]) {
					if (nd.what & NEED_CB) o_cb[nd.idx] = container.intern_barcode(std::string(nd.cb));
					if (nd.what & NEED_UMI) o_umi[nd.idx] = container.intern_umi(std::string(nd.umi));
					if (nd.what & NEED_GENE) o_gene[nd.idx] = container.intern_gene(nd.gene_owned.empty() ? nd.gene : std::string_view(nd.gene_owned), nd.gene_hash);
					if (nd.what & NEED_CHR) { container.intern_chromosome_of_ref(nd.ref); o_aux[nd.idx] |= uint32_t(container.chromosome_of_ref(nd.ref)); }
				}
			phase(1);
			runs.clear();
			for (unsigned t = 0; t < NT; ++t) {
				const size_t b0 = n * t / NT;
				if (tally[t].ok) runs.push_back(CellsDataContainer::PackedRun{o_cb.data() + b0, o_umi.data() + b0, o_gene.data() + b0, o_aux.data() + b0, tally[t].ok});
			}
			if (with_quality) {    // one row of ql bytes per accepted read, beside the columns (zeros for reads without a gene)
				container.reserve_quality_rows(est_reads_host, ql);
				if (o_qual.size() < n * size_t(ql)) o_qual.resize(n * size_t(ql));
				workers.run([&](unsigned t) {
					const size_t b0 = n * t / NT;
					for (size_t k = 0; k < tally[t].ok; ++k) {
						uint8_t *row = o_qual.data() + (b0 + k) * size_t(ql);
						if (o_qp[b0 + k]) std::memcpy(row, o_qp[b0 + k], ql); else std::memset(row, 0, ql);
					}
				});
				q_rows.clear();
				for (unsigned t = 0; t < NT; ++t) if (tally[t].ok) q_rows.push_back(o_qual.data() + (n * t / NT) * size_t(ql));
				container.add_records_packed(runs, q_rows, ql);
			} else container.add_records_packed(runs);
			phase(2);
			for (unsigned t = 0; t < NT; ++t) {
				_counters.total_reads += tally[t].total; _counters.cant_parse += tally[t].cant; _counters.low_quality += tally[t].low; _counters.saved += tally[t].ok;
			}
			return true;
		};
		// ---- device path (include/dropest_bgzf.h; DROPEST_BAM_DEVICE=1 or BamController::set_device_decode) ------------------------
		// The blocks are inflated on the GPU (csrc/k_inflate.h: a wave per block), the chain of records is found and every record's tags
		// are walked there (csrc/k_bamparse.h: a lane per record); 39 bytes per record come back instead of the ~300 of the record.  The
		// dictionaries stay here: the workers turn gene hashes and reference ids into indices, and the records that bring something NEW
		// (a gene name, a chromosome, a string with N) are fetched as bytes and go through parse_one + the intern_* members in file order,
		// exactly like the Needs of fast_window.  Windows grow from 1 MB of compressed bytes (the first ones meet most gene names).
		// Every block's CRC-32 is checked on the device (the wave that inflated it reads it back).  Falls back to the host reader (returns false before anything was added) when
		// the configuration needs what the kernels do not do: -r parameter files, gene = chromosome name, sharded containers.
		auto device_file = [&]() -> bool {
			if (_params_from_files || _gene_in_chromosome_name || !container.bulk_ingest_possible_at_all()) return false;
			const auto t_enter = clk::now();
			double ms_header = 0, ms_create = 0;
			if (_tags.intronic_read_value.size() > 24 || _tags.intergenic_read_value.size() > 24) return false;
			struct Map {
				const uint8_t *p = nullptr; size_t n = 0; int fd = -1;
				~Map() {
					const auto t = std::chrono::steady_clock::now();
					if (p) munmap(const_cast<uint8_t *>(p), n);
					if (fd >= 0) close(fd);
					if (getenv("DROPEST_BAM_TRACE")) std::fprintf(stderr, "[bam] device path: file unmapped %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count());
				}
			} map;
			map.fd = open(bam_name.c_str(), O_RDONLY);
			if (map.fd < 0) return false;
			struct stat sb;
			if (fstat(map.fd, &sb) || sb.st_size <= 0) return false;
			map.n = size_t(sb.st_size);
			void *mp = mmap(nullptr, map.n, PROT_READ, MAP_PRIVATE, map.fd, 0);
			if (mp == MAP_FAILED) return false;
			map.p = static_cast<const uint8_t *>(mp);
			// (the mapping is read at every block's header only, a cache line per ~20 KB: without this every fault maps its 16 neighbours too, and
			// half a million page-table entries of a 683 MB file took 38 ms to unmap)
			(void)madvise(mp, map.n, MADV_RANDOM);
			// where the records begin: magic, l_text, text, n_ref, (l_name, name, l_ref) x n_ref -- blocks inflated here until that much is seen
			size_t at = 0, c0 = 0; uint32_t u0 = 0;
			{
				std::vector<uint8_t> hdr;
				std::vector<std::pair<size_t, size_t>> starts;   // (file offset of the block, inflated bytes before it)
				auto have = [&](size_t nbytes) {
					while (hdr.size() < nbytes) {
						RawBlock b;
						const size_t before = at;
						if (!read_block(map.p, map.n, at, b, bam_name)) throw std::runtime_error("Truncated BAM header: " + bam_name);
						starts.emplace_back(before, hdr.size());
						hdr.resize(hdr.size() + b.isize);
						inflate_block(b, hdr.data() + hdr.size() - b.isize);
					}
				};
				have(12);
				if (std::memcmp(hdr.data(), "BAM\1", 4) != 0) throw std::runtime_error("Can't open BAM file: " + bam_name);
				size_t h = 8 + size_t(le32(hdr.data() + 4));
				have(h + 4);
				const uint32_t n_ref = le32(hdr.data() + h);
				h += 4;
				refs.clear();
				for (uint32_t r = 0; r < n_ref; ++r) {
					have(h + 4);
					const size_t ln = le32(hdr.data() + h);
					have(h + 4 + ln + 4);
					refs.emplace_back(reinterpret_cast<const char *>(hdr.data() + h + 4), ln ? ln - 1 : 0);
					h += 4 + ln + 4;
				}
				container.set_reference_names(refs);
				c0 = at; u0 = 0;                                 // the header ends with a block: the records open the next one
				for (size_t k = starts.size(); k-- > 0;)
					if (starts[k].second <= h && h < (k + 1 < starts.size() ? starts[k + 1].second : hdr.size())) { c0 = starts[k].first; u0 = uint32_t(h - starts[k].second); break; }
			}
			ms_header = since(t_enter);
			dropest_bam_parse_cfg cfg{};
			for (int k = 0; k < 6; ++k) cfg.tag[k] = wanted_tags[k];
			cfg.filled_bam = _filled_bam ? 1 : 0; cfg.min_phred = _min_barcode_phred; cfg.has_read_type = _tags.read_type.empty() ? 0 : 1;
			cfg.n_refs = int32_t(refs.size());
			cfg.intronic_len = uint32_t(_tags.intronic_read_value.size()); cfg.intergenic_len = uint32_t(_tags.intergenic_read_value.size());
			std::memcpy(cfg.intronic, _tags.intronic_read_value.data(), cfg.intronic_len);
			std::memcpy(cfg.intergenic, _tags.intergenic_read_value.data(), cfg.intergenic_len);
			// One decoder per device is kept between the files of one parse_bam_files call (streams, device buffers of ~15 x the staging size --
			// 2 to 7 GB for windows of 128 to 256 MB -- and 2 x 32 to 2 x 256 MB of pinned staging: ~30 ms to set up, ~40 ms to give back); after the
			// last file they are given back on a helper thread (release_decoders_in_background) unless DROPEST_BAM_KEEP_DECODERS=1 keeps them
			// for the next call of a long-lived process.
			dropest_bam_decoder *dec = nullptr;
			{
				std::lock_guard<std::mutex> lk(decoder_cache_mutex());
				auto it = decoder_cache().find(container.device());
				if (it != decoder_cache().end()) { dec = it->second; decoder_cache().erase(it); }
			}
			if (dec && dropest_bam_decoder_reset(dec, &cfg)) { dropest_bam_decoder_destroy(dec); dec = nullptr; }
			if (!dec && dropest_bam_decoder_create(container.device(), &cfg, &dec)) return false;       // (no GPU for this: the host reader does it)
			// its kernels on the container's stream for this file: that one exists and has run kernels; one of the decoder's own is ~16 ms to make
			if (container.handle() && dropest_bam_decoder_use_stream(dec, dropest_stream(container.handle()))) { dropest_bam_decoder_destroy(dec); return false; }
			ms_create = since(t_enter) - ms_header;
			struct Keep {    // back into the cache when the file went through, destroyed when it did not (whatever state an exception left)
				dropest_bam_decoder *d; int device; bool ok = false;
				~Keep() {
					const auto t_k = std::chrono::steady_clock::now();
					(void)dropest_bam_decoder_use_stream(d, nullptr);      // (the container's stream is the container's)
					if (getenv("DROPEST_BAM_TRACE")) std::fprintf(stderr, "[bam] device path: the lent stream given back (and waited for) %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_k).count());
					if (!ok) { dropest_bam_decoder_destroy(d); return; }
					std::lock_guard<std::mutex> lk(decoder_cache_mutex());
					auto &slot = decoder_cache()[device];
					if (slot) dropest_bam_decoder_destroy(slot);
					slot = d;
				}
			} keep_dec{dec, container.device()};
			// -g: the annotation's flat tables on the device (annotation_api.hip); the decoder asks it about the two ends of every alignment
			struct FreeAnn { dropest_annotation *a = nullptr; ~FreeAnn() { if (a) dropest_annotation_destroy(a); } } ann;
			Tools::GeneAnnotation::RefGenesContainer::Flat flat;
			std::vector<uint64_t> ann_gene_hash;
			std::vector<int32_t> id_of_ann_gene;
			if (!_genes.is_empty()) {
				try { flat = _genes.flatten(); } catch (const std::exception &) { return false; }   // (positions beyond 32 bits: the host reader does it)
				dropest_flat_annotation fa{};
				fa.n_chr = uint32_t(flat.chr_names.size()); fa.n_seg = uint32_t(flat.seg_start.size()); fa.n_tr = uint32_t(flat.tr_gene.size()); fa.n_genes = uint32_t(flat.gene_names.size());
				fa.use_introns_from_gtf = flat.use_introns_from_gtf ? 1 : 0;
				fa.chr_seg_begin = flat.chr_seg_begin.data(); fa.seg_start = flat.seg_start.data(); fa.seg_end = flat.seg_end.data(); fa.seg_tr_begin = flat.seg_tr_begin.data();
				fa.seg_tr = flat.seg_tr.data(); fa.tr_gene = flat.tr_gene.data(); fa.tr_exon_begin = flat.tr_exon_begin.data(); fa.tr_intron_begin = flat.tr_intron_begin.data();
				fa.exon_start = flat.exon_start.data(); fa.exon_end = flat.exon_end.data(); fa.intron_start = flat.intron_start.data(); fa.intron_end = flat.intron_end.data();
				if (dropest_annotation_create(container.device(), &fa, &ann.a)) return false;
				std::unordered_map<std::string, int32_t> chr_index;
				for (size_t k = 0; k < flat.chr_names.size(); ++k) chr_index.emplace(flat.chr_names[k], int32_t(k));
				std::vector<int32_t> ann_chr_of_ref(refs.size(), -1);
				for (size_t r = 0; r < refs.size(); ++r) { auto it = chr_index.find(refs[r]); if (it != chr_index.end()) ann_chr_of_ref[r] = it->second; }
				if (dropest_bam_decoder_set_annotation(dec, ann.a, ann_chr_of_ref.data(), uint32_t(refs.size()))) throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
				ann_gene_hash.resize(flat.gene_names.size());
				for (size_t g = 0; g < flat.gene_names.size(); ++g) ann_gene_hash[g] = CellsDataContainer::hash_name(flat.gene_names[g]);
				id_of_ann_gene.assign(flat.gene_names.size(), -1);
			}
			auto host_inflate = [](const uint8_t *in, uint32_t in_len, uint8_t *out_bytes, uint32_t out_len, void *) -> int {
				RawBlock b; b.cdata = in; b.clen = in_len; b.isize = out_len; b.crc = le32(in + in_len);
				try { inflate_block(b, out_bytes); } catch (...) { return 1; }
				return 0;
			};
			std::vector<uint32_t> idx_all, p_pos, p_gene, p_aux; std::vector<uint64_t> need_off, p_cb, p_umi; std::vector<uint8_t> need_bytes;
			std::vector<uint64_t> dict_hash; std::vector<uint32_t> dict_id; std::vector<int32_t> dict_chr;
			std::vector<uint32_t> name_off; std::vector<uint8_t> name_pool;
			double dev_ms[4] = {0, 0, 0, 0}, host_ms[3] = {0, 0, 0};
			const auto t_file = clk::now();
			// (DROPEST_BAM_RAMP="first MB,factor": the first window and how fast the windows grow to window_max)
			// (1 MB first: the file's genes and chromosomes are mostly met there, and every one of them is a record the host parses -- a first window of
			// 4 / 8 / 16 MB: 65 -> 76 / 79 / 87 ms on the 3.2 x file; then x 8: 1, 8, 64 MB ... measured against x 4 and x 16 with the round's final decoder:
			// 3.2 x file 65-70 -> 59-67 ms, 10.8 x file 85-87 -> 73-75)
			size_t ramp_first = 1, ramp_factor = 8;
			if (const char *e = getenv("DROPEST_BAM_RAMP")) { int a = 0, b = 0; if (std::sscanf(e, "%d,%d", &a, &b) == 2 && a >= 1 && b >= 2) { ramp_first = size_t(a); ramp_factor = size_t(b); } }
			size_t window_bytes = ramp_first << 20, n_windows = 0, repaired = 0, refused = 0, n_needs = 0;
			// windows: a machine's worth of blocks (one wave per block) in a file of a few hundred MB, two in a long one (measured on 0.4 and 1.6 GB:
			// NOTES_r05 §11) -- the pinned staging buffers of larger windows cost more to allocate than their fuller kernels give back
			const bool long_file = map.n > (size_t(1) << 30);
			size_t window_max = size_t(getenv("DROPEST_BAM_DEVICE_WINDOW_MB") ? std::max(1, atoi(getenv("DROPEST_BAM_DEVICE_WINDOW_MB"))) : (long_file ? 80 : 48)) << 20;
			// (DROPEST_BAM_WINDOW_BLOCKS: blocks per window instead of the device's wave slots, twice that in a long file)
			const size_t window_blocks = getenv("DROPEST_BAM_WINDOW_BLOCKS") ? size_t(std::max(64, atoi(getenv("DROPEST_BAM_WINDOW_BLOCKS")))) : std::max<size_t>(1024, dropest_bam_decoder_wave_slots(dec)) * (long_file ? 2 : 1);
			size_t inflated_seen = 0, compressed_seen = 0;      // over the first 64 blocks behind the header: how far this file inflates
			if (!getenv("DROPEST_BAM_DEVICE_WINDOW_MB")) {
				// ... of blocks, not of bytes: a file that deflates 3 x (real bases and qualities) has blocks of ~20 KB where the 10 x synthetic ones have 6 KB,
				// and a window of 48 MB of them would fill a third of the wave slots -- each window takes the time of ONE block however many it holds.
				// The blocks behind the header say how large they are; up to 128 / 256 MB of staging (9.5 ms of pinning per 48 MB: NOTES_r05 section 13).
				size_t at = c0, n_seen = 0, bytes_seen = 0;
				while (n_seen < 64 && at + 18 <= map.n && map.p[at] == 31 && map.p[at + 1] == 139) {
					const size_t xlen = le16(map.p + at + 10);
					size_t bsize = 0;
					for (size_t x = 0; x + 4 <= xlen && at + 12 + x + 6 <= map.n;) { const uint8_t *sf = map.p + at + 12 + x; const size_t sl = le16(sf + 2); if (sf[0] == 'B' && sf[1] == 'C' && sl == 2) bsize = size_t(le16(sf + 4)) + 1; x += 4 + sl; }
					if (!bsize || at + bsize > map.n) break;
					inflated_seen += le32(map.p + at + bsize - 4);
					at += bsize; bytes_seen += bsize; ++n_seen;
				}
				compressed_seen = bytes_seen;
				if (n_seen >= 8) {
					const size_t want = window_blocks * (bytes_seen / n_seen + 1);
					window_max = std::min(std::max(window_max, want), size_t(getenv("DROPEST_BAM_WINDOW_BLOCKS") ? 512 : long_file ? 256 : 128) << 20);
					window_max = std::min(window_max, std::max<size_t>(map.n - c0, size_t(1) << 20));      // (no more than the file)
				}
			}
			// The compressed bytes reach the device in pieces: helper threads (up to eight for a long window) read the next window from the file into
			// pinned pieces of 2 MB (pread: page cache -> pinned memory) and sends each on its way (dropest_bam_decoder_upload_piece) while the device
			// and this thread work on the window before.  The window itself is the mapped file's bytes: the block table is made from them, the host
			// fall-back for a refused block reads them.  (Rounds 4-6a read whole windows into two pinned buffers of the window's size: 2 x 128 MB were
			// 35-65 ms to allocate, a fifth to a third of a 16 M read file's time -- DROPEST_BAM_WHOLE_WINDOW_STAGING=1 takes that road.)
			struct Staged { const uint8_t *p = nullptr; size_t used = 0; bool final = false; std::string error; };
			const size_t stage_cap = window_max + (size_t(1) << 17);
			const bool upload_ahead = !getenv("DROPEST_BAM_NO_UPLOAD_AHEAD");
			const bool in_pieces = upload_ahead && !getenv("DROPEST_BAM_WHOLE_WINDOW_STAGING");
			uint8_t *stage_p[2] = {nullptr, nullptr};
			const size_t PIECE = size_t(std::min(16, std::max(1, getenv("DROPEST_BAM_PIECE_MB") ? atoi(getenv("DROPEST_BAM_PIECE_MB")) : 2))) << 20;
			// (the copy out of the page cache runs at ~4-5 GB/s per thread; a piece being sent, one being filled, per reader)
			const uint32_t READERS = uint32_t(std::min(8, std::max(1, getenv("DROPEST_BAM_READERS") ? atoi(getenv("DROPEST_BAM_READERS")) : 8))), N_PIECES = 2 * READERS;
			uint8_t *piece_p[16] = {};
			if (in_pieces) {
				for (int k = 0; k < 2; ++k)      // (first: the upload stream is made beside the pinned allocation, and a device allocation would wait for it)
					if (dropest_bam_decoder_reserve(dec, k, stage_cap, compressed_seen ? uint64_t(double(stage_cap) * double(inflated_seen) / double(compressed_seen) * 1.25) : 0)) throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
				if (getenv("DROPEST_BAM_TRACE")) std::fprintf(stderr, "[bam] device path: the device buffers of two windows of %zu MB %.1f ms\n", stage_cap >> 20, since(t_file));
				if (dropest_bam_decoder_pieces(dec, N_PIECES, PIECE, piece_p)) throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
				if (getenv("DROPEST_BAM_TRACE")) std::fprintf(stderr, "[bam] device path: ... and %u pinned pieces of %zu MB %.1f ms\n", N_PIECES, PIECE >> 20, since(t_file));
			} else
				for (int k = 0; k < 2; ++k)      // (pinning 2 x 80 MB and reserving a window's device buffers: 35-50 ms; side by side on two threads they take as long)
					if (dropest_bam_decoder_staging(dec, k, stage_cap, &stage_p[k])) throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
			const double ms_setup = since(t_file);
			double ms_wait_read = 0, ms_window_calls = 0;
			size_t file_at = c0;
			const size_t block_cap = getenv("DROPEST_BAM_DEVICE_WINDOW_MB") ? size_t(-1) : window_blocks;   // (one wave per block: a full machine's worth per window)
			// whole blocks of p[0 .. got): up to `want` bytes of them (at least one), and no more than the device inflates at once
			auto whole_blocks = [&](const uint8_t *p, size_t got, size_t want, Staged &st) -> size_t {
				size_t o = 0, n_blocks = 0;
				while (o + 18 <= got && (o < want || o == 0) && n_blocks < block_cap) {
					const uint8_t *h = p + o;
					if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { st.error = "Not a BGZF/BAM file"; return 0; }
					const size_t xlen = le16(h + 10);
					if (o + 12 + xlen > got) break;
					size_t bsize = 0;
					for (size_t x = 0; x + 4 <= xlen;) { const uint8_t *sf = h + 12 + x; const size_t sl = le16(sf + 2); if (sf[0] == 'B' && sf[1] == 'C' && sl == 2 && x + 6 <= xlen) bsize = size_t(le16(sf + 4)) + 1; x += 4 + sl; }
					if (!bsize) { st.error = "BGZF block without BC subfield"; return 0; }
					if (o + bsize > got) break;
					o += bsize; ++n_blocks;
				}
				if (!o && got) st.error = "Truncated BGZF block";
				return o;
			};
			// The same from the file itself, one pread per block (the eight bytes that end a block and the header of the next), with the block table the
			// inflate kernel takes as a by-product -- the mapped file is then not touched at all (walking it there mapped half a million pages of a
			// 683 MB file, 38 ms to unmap when the file was done)
			struct Blocks {
				std::vector<uint64_t> in_off; std::vector<uint32_t> in_len, out_len, crc;
				void clear() { in_off.clear(); in_len.clear(); out_len.clear(); crc.clear(); }
			} blocks_of[2];
			// blocks from offset `from` of the window (a block's first byte) until one begins at or behind `until` (and at least one), none beyond `avail`,
			// at most max_blocks; their table appended to B (offsets relative to the window); returns where it stopped
			auto walk_blocks = [&](size_t from, size_t until, size_t avail, size_t max_blocks, Blocks &B, std::string &error) -> size_t {
				uint8_t hb[8 + 64];
				std::vector<uint8_t> big;
				auto get = [&](uint8_t *to, size_t off, size_t n) -> bool {
					size_t g = 0;
					while (g < n) { const ssize_t r = pread(map.fd, to + g, n - g, off_t(file_at + off + g)); if (r <= 0) return false; g += size_t(r); }
					return true;
				};
				size_t o = from, n_blocks = 0, h_len = std::min<size_t>(avail - from, 64);
				if (!get(hb + 8, from, h_len)) { error = "Can't read BAM file"; return from; }
				while (o + 18 <= avail && (o < until || o == from) && n_blocks < max_blocks) {
					const uint8_t *h = hb + 8;
					if (h_len < 18) break;
					if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { error = "Not a BGZF/BAM file"; return from; }
					const size_t xlen = le16(h + 10);
					if (o + 12 + xlen > avail) break;
					if (12 + xlen > h_len) {      // (an extra field beyond the 52 bytes read ahead: never from htslib)
						big.resize(12 + xlen);
						if (!get(big.data(), o, 12 + xlen)) { error = "Can't read BAM file"; return from; }
						h = big.data();
					}
					size_t bsize = 0;
					for (size_t x = 0; x + 4 <= xlen;) { const uint8_t *sf = h + 12 + x; const size_t sl = le16(sf + 2); if (sf[0] == 'B' && sf[1] == 'C' && sl == 2 && x + 6 <= xlen) bsize = size_t(le16(sf + 4)) + 1; x += 4 + sl; }
					if (!bsize || bsize < 12 + xlen + 8) { error = "BGZF block without BC subfield"; return from; }
					if (o + bsize > avail) break;
					const size_t t_off = o + bsize - 8, t_len = std::min<size_t>(avail - t_off, sizeof(hb));
					if (!get(hb, t_off, t_len)) { error = "Can't read BAM file"; return from; }
					const uint32_t isize = le32(hb + 4);
					if (isize > 65536u) { error = "BGZF block with ISIZE beyond 64 KB"; return from; }
					B.in_off.push_back(o + 12 + xlen); B.in_len.push_back(uint32_t(bsize - 12 - xlen - 8)); B.out_len.push_back(isize); B.crc.push_back(le32(hb));
					o += bsize; ++n_blocks;
					h_len = t_len - 8;
				}
				return o;
			};
			// The first block that begins at or behind `from`, found by its first sixteen bytes as htslib and bgzip write them (gzip magic, FEXTRA, XLEN 6,
			// the BC subfield of two bytes) and by the block behind it beginning the same way; size_t(-1): none in the next 64 KB + (not such a file)
			auto find_block = [&](size_t from, size_t avail) -> size_t {
				static const uint8_t magic[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0};
				if (from + 64 >= avail) return size_t(-1);
				std::vector<uint8_t> buf(std::min<size_t>(avail - from, (size_t(1) << 16) + 32));
				size_t g = 0;
				while (g < buf.size()) { const ssize_t r = pread(map.fd, buf.data() + g, buf.size() - g, off_t(file_at + from + g)); if (r <= 0) return size_t(-1); g += size_t(r); }
				for (size_t i = 0; i + 18 <= buf.size(); ++i) {
					if (buf[i] != 31 || std::memcmp(buf.data() + i, magic, 16)) continue;
					const size_t bsize = size_t(le16(buf.data() + i + 16)) + 1, next = from + i + bsize;
					if (bsize < 26 || next + 16 > avail) continue;
					uint8_t nb[16];
					size_t h = 0;
					while (h < 16) { const ssize_t r = pread(map.fd, nb + h, 16 - h, off_t(file_at + next + h)); if (r <= 0) return size_t(-1); h += size_t(r); }
					if (!std::memcmp(nb, magic, 16)) return from + i;
				}
				return size_t(-1);
			};
			// The whole blocks of a window and their table.  A walk from header to header is one pread per block (~0.7 us): 270 000 blocks of a
			// 1.6 GB file of 6 KB blocks are 0.2 s on one thread -- more than the device needs for the file -- so a long window is cut into four
			// stretches, each begun at a block found by its magic bytes; the stretches must meet (a stretch ends where the next one began), else
			// one walk from the start decides.
			std::atomic<size_t> stretched_windows{0};
			auto whole_blocks_of_file = [&](size_t avail, size_t want, Blocks &B, Staged &st) -> size_t {
				B.clear();
				constexpr size_t STRETCHES = 4;
				// (no further than the blocks the device takes at once are expected to reach: beyond that the stretches' blocks would be thrown away)
				const size_t avg_block = compressed_seen >= 64 ? std::max<size_t>(compressed_seen / 64, 64) : size_t(1) << 14;
				const size_t reach = block_cap == size_t(-1) ? want : std::min<size_t>(want, size_t(double(block_cap) * double(avg_block) * 0.9));
				const size_t span = std::min(reach, avail);
				static const bool one_walker = getenv("DROPEST_BAM_ONE_WALKER") != nullptr;
				// (tests: DROPEST_BAM_TEST_STRETCH="bytes,blocks" lowers the two thresholds so that small files take the four stretches)
				static const std::pair<size_t, size_t> least = [] {
					size_t a = size_t(8) << 20, b = 8192;
					if (const char *e = getenv("DROPEST_BAM_TEST_STRETCH")) { long x = 0, y = 0; if (std::sscanf(e, "%ld,%ld", &x, &y) == 2 && x > 0 && y > 0) { a = size_t(x); b = size_t(y); } }
					return std::make_pair(a, b);
				}();
				if (span >= least.first && span / avg_block >= least.second && !one_walker) {      // (measured: files of 6 KB blocks 213-250 -> 196-200 ms for 64 M reads; windows of fewer, larger blocks the same or a little slower)
					Blocks part[STRETCHES];
					size_t begin[STRETCHES], end[STRETCHES];
					std::string err[STRETCHES];
					std::vector<std::future<void>> th;
					bool found = true;
					begin[0] = 0;
					for (size_t k = 1; k < STRETCHES && found; ++k) { begin[k] = find_block(span / STRETCHES * k, avail); found = begin[k] != size_t(-1); }
					if (found) {
						auto stretch = [&](size_t k) { end[k] = walk_blocks(begin[k], k + 1 < STRETCHES ? span / STRETCHES * (k + 1) : reach, avail, size_t(-1), part[k], err[k]); };
						for (size_t k = 1; k < STRETCHES; ++k) th.push_back(std::async(std::launch::async, stretch, k));
						stretch(0);
						for (auto &f : th) f.get();
						bool meet = true;
						for (size_t k = 0; k < STRETCHES; ++k) meet = meet && err[k].empty() && (k + 1 == STRETCHES || end[k] == begin[k + 1]);
						size_t total = 0;
						for (size_t k = 0; k < STRETCHES; ++k) total += part[k].in_off.size();
						if (meet && total && total <= block_cap) {
							for (size_t k = 0; k < STRETCHES; ++k) {
								B.in_off.insert(B.in_off.end(), part[k].in_off.begin(), part[k].in_off.end()); B.in_len.insert(B.in_len.end(), part[k].in_len.begin(), part[k].in_len.end());
								B.out_len.insert(B.out_len.end(), part[k].out_len.begin(), part[k].out_len.end()); B.crc.insert(B.crc.end(), part[k].crc.begin(), part[k].crc.end());
							}
							++stretched_windows;
							return end[STRETCHES - 1];
						}
					}
					B.clear();
				}
				const size_t o = walk_blocks(0, want, avail, block_cap, B, st.error);
				if (!st.error.empty()) return 0;
				if (!o && avail) st.error = "Truncated BGZF block";
				return o;
			};
			auto read_window = [&](int which, size_t want) {
				Staged st;
				const size_t ask = std::min(std::min(want + (size_t(1) << 16) + 64, stage_cap), map.n - file_at);
				if (in_pieces) {
					// the window's bytes to the device piece by piece (helper threads: everything up to `want` needs no block boundary), its extent and
					// block table from the file meanwhile (this thread), the last blocks' bytes beyond `want` at the end
					st.p = map.p + file_at;
					Blocks &B = blocks_of[which];
					const auto t_rw = clk::now();
					struct SayRw { decltype(t_rw) t; size_t &used; ~SayRw() { if (getenv("DROPEST_BAM_TRACE_READER")) std::fprintf(stderr, "[bam] reader: a window of %.1f MB read and sent in %.2f ms\n", double(used) / 1048576.0, std::chrono::duration<double, std::milli>(clk::now() - t).count()); } } say_rw{t_rw, st.used};
					std::atomic<bool> failed{dropest_bam_decoder_upload_begin(dec, which) != 0};
					const size_t covered = std::min(want, ask);
					const size_t n_pieces = (covered + PIECE - 1) / PIECE;
					const uint32_t n_readers = uint32_t(std::min<size_t>(READERS, n_pieces));
					const bool trace_rw = getenv("DROPEST_BAM_TRACE_READER") != nullptr && file_at == c0;
					auto lap_rw = [&](const char *what) { if (trace_rw) std::fprintf(stderr, "[bam] reader, first window: %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(clk::now() - t_rw).count()); };
					lap_rw("upload_begin done");
					auto send = [&](uint32_t slot, size_t from, size_t len) {
						if (dropest_bam_decoder_piece_wait(dec, slot)) { failed = true; return; }
						lap_rw("piece_wait done");
						size_t g = 0;
						while (g < len) {
							const ssize_t got = pread(map.fd, piece_p[slot] + g, len - g, off_t(file_at + from + g));
							if (got <= 0) { failed = true; return; }
							g += size_t(got);
						}
						lap_rw("pread done");
						if (dropest_bam_decoder_upload_piece(dec, which, slot, from, len)) failed = true;
						lap_rw("upload_piece done");
					};
					auto reader = [&](uint32_t r) {
						for (size_t k = r, turn = 0; k < n_pieces && !failed.load(std::memory_order_relaxed); k += n_readers, ++turn)
							send(2u * r + uint32_t(turn & 1u), k * PIECE, std::min(PIECE, covered - k * PIECE));
					};
					std::vector<std::future<void>> helpers;
					for (uint32_t r = 0; r < n_readers; ++r) helpers.push_back(std::async(std::launch::async, reader, r));
					const size_t o = whole_blocks_of_file(ask, want, B, st);
					lap_rw("block walk done");
					for (auto &f : helpers) f.get();
					lap_rw("helpers joined");
					if (!st.error.empty()) return st;
					for (size_t from = covered; from < o && !failed.load(); from += PIECE) send(0, from, std::min(PIECE, o - from));      // (at most 64 KB)
					st.used = o; file_at += o; st.final = file_at >= map.n;
					// (a piece that could not be read or sent: nothing is declared, and the window call copies the mapped bytes itself)
					const dropest_bgzf_blocks table{uint64_t(B.in_off.size()), B.in_off.data(), B.in_len.data(), B.out_len.data(), B.crc.data()};
					if (!failed.load()) (void)dropest_bam_decoder_upload_done(dec, which, st.p, o, &table);
					lap_rw("upload_done done");
					return st;
				}
				uint8_t *const buf = stage_p[which];
				st.p = buf;
				// (the copy out of the page cache runs at ~4-5 GB/s per thread: a long window is read in four pieces side by side, so that the reader
				// stays ahead of the device -- which takes ~11 ms per 64 MB window)
				auto read_piece = [&](size_t from, size_t len) -> long {
					size_t g = 0;
					while (g < len) {
						const ssize_t r = pread(map.fd, buf + from + g, len - g, off_t(file_at + from + g));
						if (r < 0) return -1;
						if (r == 0) break;
						g += size_t(r);
					}
					return long(g);
				};
				size_t got = 0;
				const size_t n_pieces = ask > (size_t(8) << 20) ? 4 : 1, piece = (ask / n_pieces + 4095) & ~size_t(4095);
				std::vector<std::future<long>> rest;
				for (size_t k = 1; k < n_pieces; ++k) if (k * piece < ask) rest.push_back(std::async(std::launch::async, read_piece, k * piece, std::min(piece, ask - k * piece)));
				const long g0 = read_piece(0, std::min(piece, ask));
				bool whole = g0 == long(std::min(piece, ask)), failed = g0 < 0;
				if (g0 > 0) got = size_t(g0);
				for (size_t k = 0; k < rest.size(); ++k) {
					const long g = rest[k].get();
					if (g < 0) failed = true;
					else if (whole) { got += size_t(g); whole = size_t(g) == std::min(piece, ask - (k + 1) * piece); }
				}
				if (failed) { st.error = "Can't read BAM file"; return st; }
				const size_t o = whole_blocks(buf, got, want, st);
				if (!st.error.empty()) return st;
				st.used = o; file_at += o; st.final = file_at >= map.n;
				if (upload_ahead) (void)dropest_bam_decoder_upload(dec, which, o);   // on its way while the window before it is in the kernels (if this fails, the window call copies)
				return st;
			};
			bool first = true, dict_dirty = true;
			size_t est_reads = 0;
			int which = 0;
			std::future<Staged> next = std::async(std::launch::async, read_window, which, window_bytes);
			// Round 6, DROPEST_BAM_PIPELINE=1 (off by default): the inflate of window k + 1 is given to the device BEFORE this thread waits for window k
			// (dropest_bam_decoder_window_inflate / _chain).  A window's inflate lasts as long as its slowest block (7-11 ms on a file that deflates
			// 3 x) and one window after the other leaves wave slots empty -- 86 ms of inflate in situ for a kernel that needs 42
			// (profiles/r06a_bam_3x_file_device_trace.txt) -- but the next window's blocks then hold every slot while this window's small kernels
			// and copies want some: inside the window calls 109-121 -> 81-95 ms, the dictionaries' copies 1 -> 31 ms, two more streams to create
			// (8 ms each): ingest 195 -> 233 ms on the same box.  Kept as an order the library supports and the suite runs; not the default.
			const bool pipeline = upload_ahead && getenv("DROPEST_BAM_PIPELINE") && !getenv("DROPEST_BAM_NO_PIPELINE");
			Staged stg_ahead; int slot_cur = 0, slot_ahead = 0; bool have_ahead = false;
			for (;;) {
				auto t_wait = clk::now();
				Staged stg;
				if (have_ahead) { stg = std::move(stg_ahead); slot_cur = slot_ahead; have_ahead = false; }
				else {
					stg = next.get();
					ms_wait_read += since(t_wait);
					if (!stg.error.empty()) throw std::runtime_error(stg.error + ": " + bam_name);
					window_bytes = std::min(window_bytes * ramp_factor, window_max);
					which ^= 1;
					if (!stg.final) next = std::async(std::launch::async, read_window, which, window_bytes);
					if (pipeline && dropest_bam_decoder_window_inflate(dec, stg.p, stg.used, &slot_cur)) {
						if (next.valid()) next.wait();
						throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
					}
				}
				const bool final = stg.final;
				const size_t used = stg.used;
				struct Drain { std::future<Staged> &f; ~Drain() { if (f.valid()) f.wait(); } } drain{next};   // (an exception below must not leave the reader running on freed buffers)
				if (pipeline && !final) {      // the window after this one: read by now (its buffer is the other one), its blocks to the device
					t_wait = clk::now();
					stg_ahead = next.get();
					ms_wait_read += since(t_wait);
					if (!stg_ahead.error.empty()) throw std::runtime_error(stg_ahead.error + ": " + bam_name);
					auto t_a = clk::now();
					if (dropest_bam_decoder_window_inflate(dec, stg_ahead.p, stg_ahead.used, &slot_ahead)) throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
					ms_window_calls += since(t_a);
					have_ahead = true;
				}
				auto t_phase = clk::now();
				if (dict_dirty) {      // the dictionaries as they stand (other files, earlier windows, add_record calls) go to the device
					container.dictionary_snapshot(dict_hash, dict_id, dict_chr);
					if (dropest_bam_decoder_set_dictionaries(dec, dict_hash.data(), dict_id.data(), uint32_t(dict_hash.size()), dict_chr.data(), uint32_t(dict_chr.size())))
						throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
					{   // ... and the genes' names: a hash the device finds is confirmed byte by byte (two names with one FNV-1a value must not share an index)
						const auto &names = container.gene_indexer().values();
						name_off.resize(names.size() + 1);
						size_t bytes = 0;
						for (size_t g = 0; g < names.size(); ++g) { name_off[g] = uint32_t(bytes); bytes += names[g].size(); }
						name_off[names.size()] = uint32_t(bytes);
						name_pool.resize(bytes + 1);
						for (size_t g = 0; g < names.size(); ++g) std::memcpy(name_pool.data() + name_off[g], names[g].data(), names[g].size());
						if (bytes > 0xFFFFFFF0ull || dropest_bam_decoder_set_gene_names(dec, name_off.data(), name_pool.data(), uint32_t(names.size())))
							throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
					}
					if (ann.a) {   // the annotation's genes that the dictionary holds by now
						for (size_t g = 0; g < flat.gene_names.size(); ++g)
							if (id_of_ann_gene[g] < 0) id_of_ann_gene[g] = int32_t(container.lookup_gene(ann_gene_hash[g], flat.gene_names[g]));
						if (dropest_bam_decoder_set_annotation_genes(dec, id_of_ann_gene.data(), uint32_t(id_of_ann_gene.size())))
							throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
					}
					dict_dirty = false;
				}
				host_ms[0] += since(t_phase);
				dropest_bam_window w{};
				auto t_call = clk::now();
				if (!pipeline) {
					if (dropest_bam_decoder_window(dec, stg.p, used, first ? u0 : 0u, final ? 1 : 0, host_inflate, nullptr, &w))
						throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
				} else {
					if (dropest_bam_decoder_window_chain(dec, slot_cur, first ? u0 : 0u, final ? 1 : 0, host_inflate, nullptr))
						throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
					// this window's compressed bytes are done with: the reader may fill their buffer with the window after the next
					if (have_ahead && !stg_ahead.final) {
						window_bytes = std::min(window_bytes * ramp_factor, window_max);
						which ^= 1;
						next = std::async(std::launch::async, read_window, which, window_bytes);
					}
					if (dropest_bam_decoder_window_finish(dec, slot_cur, &w)) throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
				}
				ms_window_calls += since(t_call);
				if (first) {
					est_reads = size_t(double(w.n_records) * double(map.n - c0) / double(std::max<size_t>(used, 1)) * double(bam_files.size()) * 1.05);
					const auto t_e = clk::now();
					container.expect_reads(est_reads);
					if (getenv("DROPEST_BAM_TRACE")) std::fprintf(stderr, "[bam] device path: the container told to expect %zu reads %.1f ms\n", est_reads, since(t_e));
				}
				first = false;
				++n_windows; repaired += w.guesses_repaired; refused += w.refused_blocks;
				dev_ms[0] += w.ms_copy; dev_ms[1] += w.ms_inflate; dev_ms[2] += w.ms_boundaries; dev_ms[3] += w.ms_parse;
				if (getenv("DROPEST_BAM_TRACE_READER")) std::fprintf(stderr, "[bam] window %zu: %.1f MB, %u blocks, %llu records: copy in %.2f, inflate %.2f, chain %.2f, fields %.2f ms; the call %.2f ms\n", n_windows, double(used) / 1048576.0, w.n_blocks, (unsigned long long)w.n_records, w.ms_copy, w.ms_inflate, w.ms_boundaries, w.ms_parse, since(t_call));
				const size_t n = size_t(w.n_records);
				if (!n) { if (final) break; continue; }
				t_phase = clk::now();
				// UMI quality strings: one length over the gene-bearing reads of the window (and the container's so far) -> the device hands over one row
				// per accepted read beside the columns; anything else -> the window's records come back as bytes (below)
				const bool has_q = w.any_gene && w.quality_len_max > 0;
				const uint32_t q_len = w.quality_len_max;
				const bool q_bulk = has_q && w.quality_len_min == w.quality_len_max && container.bulk_ingest_possible_with_quality(q_len);
				if (has_q ? !q_bulk : !container.bulk_ingest_possible()) {
					// UMI quality strings (the container keeps them on the host): the records of this window come back as bytes and take the host
					// reader's bulk path over them (fast_window: one quality row per read while the strings have one length), or, where that
					// refuses, go record by record
					idx_all.resize(n); need_off.resize(n);
					for (size_t i = 0; i < n; ++i) idx_all[i] = uint32_t(i);
					need_bytes.resize(size_t(w.window_bytes) + 16);
					if (dropest_bam_decoder_fetch_records(dec, idx_all.data(), uint32_t(n), need_bytes.data(), need_bytes.size(), need_off.data()))
						throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
					data = need_bytes.data();
					offsets.resize(n);
					for (size_t i = 0; i < n; ++i) offsets[i] = uint32_t(need_off[i]);
					if (w.window_bytes < (uint64_t(1) << 32) && !getenv("DROPEST_BAM_RECORD_BY_RECORD") && container.bulk_ingest_possible_at_all() && fast_window(n)) {
						dict_dirty = true;
						host_ms[1] += since(t_phase);
						if (final) break;
						continue;
					}
					Parsed tmp;
					for (size_t i = 0; i < n; ++i) {
						parse_one(need_bytes.data() + need_off[i], tmp);
						switch (tmp.status) {
							case SKIP: break;
							case CANT_PARSE_NO_COUNT: ++_counters.cant_parse; break;
							case CANT_PARSE: ++_counters.total_reads; ++_counters.cant_parse; break;
							case LOW_QUALITY: ++_counters.total_reads; ++_counters.low_quality; break;
							default: ++_counters.total_reads; container.add_record(tmp.r); ++_counters.saved;
						}
					}
					dict_dirty = true;
					host_ms[1] += since(t_phase);
					if (final) break;
					continue;
				}
				// what the dictionaries have not seen, in file order (per record: barcode, UMI, gene, chromosome -- the order of add_record)
				if (w.n_need) {
					const size_t m = w.n_need;
					n_needs += m;
					size_t bytes = 0;
					for (size_t k = 0; k < m; ++k) bytes += w.need_size[k];
					need_bytes.resize(bytes + 16); need_off.resize(m);
					p_pos.resize(m); p_cb.resize(m); p_umi.resize(m); p_gene.resize(m); p_aux.resize(m);
					if (dropest_bam_decoder_fetch_records(dec, w.need_rec, uint32_t(m), need_bytes.data(), need_bytes.size(), need_off.data()))
						throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
					Parsed tmp;
					for (size_t k = 0; k < m; ++k) {
						parse_one(need_bytes.data() + need_off[k], tmp);
						if (tmp.status != OK) throw std::runtime_error("internal: the device and the host read a BAM record differently: " + bam_name);
						const CellsDataContainer::ParsedRead &r = tmp.r;
						const bool has_gene = !r.gene.empty();
						p_pos[k] = w.need_pos[k];
						p_cb[k] = r.cb_code ? r.cb_code : container.intern_barcode(std::string(r.cb));
						if (has_gene) {
							p_umi[k] = r.umi_code ? r.umi_code : container.intern_umi(std::string(r.umi));
							p_gene[k] = container.intern_gene(r.gene, r.gene_hash);
						} else { p_umi[k] = 1; p_gene[k] = DROPEST_NO_GENE; }
						uint32_t aux = uint32_t(r.mark) << 16;
						if (!has_gene || (r.mark & (UMI::Mark::HAS_EXONS | UMI::Mark::HAS_INTRONS))) { container.intern_chromosome_of_ref(r.ref_id); aux |= uint32_t(container.chromosome_of_ref(r.ref_id)); }
						p_aux[k] = aux;
					}
					if (dropest_bam_decoder_patch(dec, p_pos.data(), p_cb.data(), p_umi.data(), p_gene.data(), p_aux.data(), uint32_t(m)))
						throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
					dict_dirty = true;
				}
				host_ms[1] += since(t_phase); t_phase = clk::now();
				bool any_gene = w.any_gene != 0;
				for (size_t k = 0; k < size_t(w.n_need) && !any_gene; ++k) any_gene = p_gene[k] != DROPEST_NO_GENE;
				if (q_bulk) {
					container.reserve_quality_rows(est_reads, q_len);
					const uint8_t *rows = nullptr;
					if (dropest_bam_decoder_quality_rows(dec, q_len, &rows)) throw std::runtime_error(std::string(dropest_bgzf_last_error()) + ": " + bam_name);
					container.add_records_packed_device(w.d_cb, w.d_umi, w.d_gene, w.d_aux, size_t(w.n_accepted), any_gene, rows, q_len);
				} else
					container.add_records_packed_device(w.d_cb, w.d_umi, w.d_gene, w.d_aux, size_t(w.n_accepted), any_gene);
				host_ms[2] += since(t_phase);
				_counters.cant_parse += size_t(w.counts[DROPEST_BAM_CANT_PARSE_NO_COUNT] + w.counts[DROPEST_BAM_CANT_PARSE]);
				_counters.low_quality += size_t(w.counts[DROPEST_BAM_LOW_QUALITY]);
				_counters.saved += size_t(w.counts[DROPEST_BAM_OK]);
				_counters.total_reads += size_t(w.counts[DROPEST_BAM_OK] + w.counts[DROPEST_BAM_CANT_PARSE] + w.counts[DROPEST_BAM_LOW_QUALITY]);
				if (final) break;
			}
			if (getenv("DROPEST_BAM_TRACE"))
				std::fprintf(stderr, "[bam] device path: %zu windows, %.1f ms behind the header; copy in %.1f ms, inflate %.1f ms (%zu blocks left to the host), record chain %.1f ms (%zu guesses "
				             "repaired), fields + dense columns %.1f ms; host: dictionaries to the device %.1f ms, %zu records with something new %.1f ms, container %.1f ms\n",
				             n_windows, since(t_file), dev_ms[0], dev_ms[1], refused, dev_ms[2], repaired, dev_ms[3], host_ms[0], n_needs, host_ms[1], host_ms[2]),
				std::fprintf(stderr, "[bam] device path: file mapped and header read %.1f ms, decoder created %.1f ms, annotation + the rest before the windows %.1f ms\n", ms_header, ms_create, std::chrono::duration<double, std::milli>(t_file - t_enter).count() - ms_header - ms_create),
				std::fprintf(stderr, "[bam] device path: pinned staging buffers %.1f ms, waiting for the file reader %.1f ms, inside the window calls %.1f ms\n", ms_setup, ms_wait_read, ms_window_calls);
			keep_dec.ok = true;
			if (getenv("DROPEST_BAM_TRACE")) std::fprintf(stderr, "[bam] device path: %.1f ms from the file's name to its last window (block tables from four stretches: %zu windows)\n", since(t_enter), stretched_windows.load());
			return true;
		};
		static const bool env_device = getenv("DROPEST_BAM_DEVICE") != nullptr && atoi(getenv("DROPEST_BAM_DEVICE")) != 0;
		if (_device_decode || env_device) {
			const auto t_df = std::chrono::steady_clock::now();
			const bool done = device_file();
			if (getenv("DROPEST_BAM_TRACE")) std::fprintf(stderr, "[bam] device path: %.1f ms in all for this file (taken: %d)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_df).count(), int(done));
			if (done) continue;
		}
		reader_p.reset(new BamReader(bam_name, _threads));
		BamReader &reader = *reader_p;
		refs = reader.reference_names();
		container.set_reference_names(refs);
		static const bool force_slow = getenv("DROPEST_BAM_RECORD_BY_RECORD") != nullptr;   // tests: the two paths agree
		bool first_window = true;
		for (;;) {
			auto t_wait = clk::now();
			if (!reader.next_window(data, offsets)) break;
			_counters.wait_ms += since(t_wait);
			if (first_window && &bam_name == &bam_files.front()) {
				// a container over several GPUs deals contiguous ranges of the stream to its shards: tell it how long the stream
				// will roughly be (records of the first window x file bytes / bytes the window covered, x the number of files)
				est_reads_host = size_t(double(offsets.size()) * reader.file_over_first_batch() * double(bam_files.size()) * 1.05);
				container.expect_reads(est_reads_host);
			}
			first_window = false;
			if (!_params_from_files && !force_slow && NT <= 256 && container.bulk_ingest_possible_at_all()) {
				auto t_fast = clk::now();
				const bool done = fast_window(offsets.size());
				_counters.parse_ms += since(t_fast);
				if (done) continue;
			}
			auto t_parse = clk::now();
			const size_t n = offsets.size();
			parsed.resize(n);
			const unsigned nt = unsigned(std::min<size_t>(nthreads, std::max<size_t>(1, n / 2048)));
			std::vector<std::thread> pool;
			std::vector<std::string> errors(nt);
			for (unsigned t = 0; t < nt; ++t)
				pool.emplace_back([&, t] {
					try { for (size_t i = n * t / nt; i < n * (t + 1) / nt; ++i) parse_one(data + offsets[i], parsed[i]); }
					catch (const std::exception &e) { errors[t] = e.what(); }
				});
			for (auto &th : pool) th.join();
			for (auto const &e : errors) if (!e.empty()) throw std::runtime_error(e + ": " + bam_name);
			_counters.parse_ms += since(t_parse);
			auto t_add = clk::now();
			// in file order: the counters and the container (first-seen ids are defined by this order)
			for (size_t i = 0; i < n; ++i) {
				if (_params_from_files && parsed[i].status != SKIP && parsed[i].status != CANT_PARSE_NO_COUNT) {
					// ReadMapParamsParser::get_read_params (:22-48) comes BEFORE the gene look-up (BamController.cpp:140-152):
					// an unknown name is "can't parse", a served name is erased, its quality verdict precedes the gene's
					Parsed &pr = parsed[i];
					std::string_view nm = pr.name;
					if (!nm.empty() && nm[0] == '@') nm.remove_prefix(1);
					auto it = _read_params.find(std::string(nm));
					if (it == _read_params.end()) pr.status = CANT_PARSE;
					else {
						const bool pass = it->second.pass_quality;
						pr.p_cb = std::move(it->second.cb); pr.p_umi = std::move(it->second.umi); pr.p_quality = std::move(it->second.umi_quality);
						_read_params.erase(it);
						if (!pass) pr.status = LOW_QUALITY;
						else if (pr.status == OK) {
							pr.r.cb = pr.p_cb; pr.r.umi = pr.p_umi; pr.r.umi_quality = pr.p_quality; pr.r.umi_quality_length = uint32_t(pr.p_quality.size());
							if (!CellsDataContainer::pack_code(pr.r.cb, pr.r.cb_code)) pr.r.cb_code = 0;
							if (!CellsDataContainer::pack_code(pr.r.umi, pr.r.umi_code)) pr.r.umi_code = 0;
						}
					}
				}
				switch (parsed[i].status) {
					case SKIP: break;
					case CANT_PARSE_NO_COUNT: ++_counters.cant_parse; break;             // reads with unknown chromosome are not counted (:107)
					case CANT_PARSE: ++_counters.total_reads; ++_counters.cant_parse; break;
					case LOW_QUALITY: ++_counters.total_reads; ++_counters.low_quality; break;
					default: ++_counters.total_reads; container.add_record(parsed[i].r); ++_counters.saved;
				}
			}
			_counters.add_ms += since(t_add);
		}
		if (getenv("DROPEST_BAM_TRACE")) std::fprintf(stderr, "[bam] bulk windows: parse + pack on the workers %.1f ms, new dictionary entries %.1f ms, container %.1f ms\n", fw_ms[0], fw_ms[1], fw_ms[2]);
	}
	// the device decoders' memory goes back before the container sorts (the cache was never released)
	if (!getenv("DROPEST_BAM_KEEP_DECODERS")) release_decoders_in_background();
}

}  // namespace BamProcessing
}  // namespace Estimation

// test hooks (tests/test_fast_inflate.py): the block decoder and the CRC against zlib's
extern "C" int dropest_test_fast_inflate(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_len) {
	return fastinflate::inflate_raw(in, size_t(in_len), out, size_t(out_len)) ? 1 : 0;
}
extern "C" uint32_t dropest_test_crc32(const uint8_t *p, uint64_t n) { return fastinflate::crc32(p, size_t(n)); }
