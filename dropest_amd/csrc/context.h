// context.h -- the device pipeline behind dropest_ctx (host orchestration of the HIP kernels).
#pragma once

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/dropest_amd.h"
#include "k_cbhash.h"
#include "k_collisions.h"
#include "k_merge.h"
#include "k_umi_directional.h"
#include "whitelist.h"
#include "k_misc.h"
#include "k_radix.h"
#include "k_ssort.h"
#include "k_segreduce.h"
#include "k_mergepath.h"
#include "matrix_decode.h"
#include "util.h"

namespace dropest {

typedef unsigned long long u64;
typedef uint32_t u32;

struct ReadChunk {
	DevBuf<u64> cb, umi;
	DevBuf<u32> gene, aux;
	const u64 *p_cb = nullptr, *p_umi = nullptr;
	const u32 *p_gene = nullptr, *p_aux = nullptr;
	uint64_t n = 0;
};

struct KernelStat { u32 launches = 0; double ms = 0, bytes = 0; };

// the four columns of `count` reads appended to the store's arrays in one launch (four device-to-device hipMemcpyAsync calls cost the host
// ~0.3 ms each: 11-15 ms over the nine windows of a BAM file)
__global__ __launch_bounds__(256) void store_append_kernel(const u64 *__restrict__ s_cb, const u64 *__restrict__ s_umi, const u32 *__restrict__ s_gene, const u32 *__restrict__ s_aux,
                                                           u64 *__restrict__ d_cb, u64 *__restrict__ d_umi, u32 *__restrict__ d_gene, u32 *__restrict__ d_aux, size_t count) {
	for (size_t i = size_t(blockIdx.x) * 256u + threadIdx.x; i < count; i += size_t(gridDim.x) * 256u) { d_cb[i] = s_cb[i]; d_umi[i] = s_umi[i]; d_gene[i] = s_gene[i]; d_aux[i] = s_aux[i]; }
}

// Pushed reads (CellsDataContainer::add_record, batched): ONE set of device arrays that grows geometrically -- no
// allocation per batch, nothing to concatenate later -- fed over PCIe on its own stream.  Host arrays that are already
// pinned are copied from in place; pageable ones go through two pinned staging buffers, the host memcpy of batch k + 1
// running under the transfer of batch k.
struct ReadStore {
	DevBuf<u64> cb, umi;
	DevBuf<u32> gene, aux;
	size_t n = 0;
	hipStream_t copy = nullptr;
	PinnedBuf<unsigned char> stage[2];
	hipEvent_t done[2] = {nullptr, nullptr};
	int cur = 0;
	~ReadStore() { for (auto e : done) if (e) (void)hipEventDestroy(e); if (copy) (void)hipStreamDestroy(copy); }
	size_t capacity() const { return cb.n; }
	void reserve(size_t want) {
		if (want <= capacity() && cb.p) return;
		size_t cap = std::max<size_t>(want, std::max<size_t>(capacity() * 2, size_t(1) << 20));
		// (the copy stream is made when something has to travel on it: a stream is 8-16 ms to create with its first use, and a store that is
		// reserved empty and then filled from device arrays -- the BAM decoder's windows -- never needs one)
		if (n && !copy) HIP_CHECK(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
		if (copy) HIP_CHECK(stream_wait(copy));
		DevBuf<u64> ncb, numi; DevBuf<u32> ngene, naux;
		ncb.alloc(cap); numi.alloc(cap); ngene.alloc(cap); naux.alloc(cap);
		ncb.mark_persistent(); numi.mark_persistent(); ngene.mark_persistent(); naux.mark_persistent();
		if (n) {
			HIP_CHECK(hipMemcpyAsync(ncb.p, cb.p, n * 8, hipMemcpyDeviceToDevice, copy));
			HIP_CHECK(hipMemcpyAsync(numi.p, umi.p, n * 8, hipMemcpyDeviceToDevice, copy));
			HIP_CHECK(hipMemcpyAsync(ngene.p, gene.p, n * 4, hipMemcpyDeviceToDevice, copy));
			HIP_CHECK(hipMemcpyAsync(naux.p, aux.p, n * 4, hipMemcpyDeviceToDevice, copy));
			HIP_CHECK(stream_wait(copy));
		}
		cb = std::move(ncb); umi = std::move(numi); gene = std::move(ngene); aux = std::move(naux);
	}
	static bool is_pinned(const void *p) {
		hipPointerAttribute_t a{};
		if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
		return a.type == hipMemoryTypeHost;
	}
	void push(const uint64_t *h_cb, const uint64_t *h_umi, const uint32_t *h_gene, const uint32_t *h_aux, size_t count) {
		if (!count) return;
		reserve(n + count);
		if (!copy) HIP_CHECK(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
		if (is_pinned(h_cb) && is_pinned(h_umi) && is_pinned(h_gene) && is_pinned(h_aux)) {
			HIP_CHECK(hipMemcpyAsync(cb.p + n, h_cb, count * 8, hipMemcpyHostToDevice, copy));
			HIP_CHECK(hipMemcpyAsync(umi.p + n, h_umi, count * 8, hipMemcpyHostToDevice, copy));
			HIP_CHECK(hipMemcpyAsync(gene.p + n, h_gene, count * 4, hipMemcpyHostToDevice, copy));
			HIP_CHECK(hipMemcpyAsync(aux.p + n, h_aux, count * 4, hipMemcpyHostToDevice, copy));
			HIP_CHECK(stream_wait(copy));   // the caller keeps ownership of its arrays
			n += count;
			return;
		}
		const size_t piece_max = size_t(4) << 20;   // reads per staging buffer (96 MB)
		for (size_t at = 0; at < count; at += piece_max) {
			const size_t m = std::min(piece_max, count - at);
			PinnedBuf<unsigned char> &st = stage[cur];
			if (!done[cur]) HIP_CHECK(hipEventCreateWithFlags(&done[cur], hipEventDisableTiming));
			else HIP_CHECK(event_wait(done[cur]));   // the transfer that used this buffer two pieces ago
			st.ensure(m * 24);
			unsigned char *b = st.p;
			std::memcpy(b, h_cb + at, m * 8); std::memcpy(b + m * 8, h_umi + at, m * 8);
			std::memcpy(b + m * 16, h_gene + at, m * 4); std::memcpy(b + m * 20, h_aux + at, m * 4);
			HIP_CHECK(hipMemcpyAsync(cb.p + n, b, m * 8, hipMemcpyHostToDevice, copy));
			HIP_CHECK(hipMemcpyAsync(umi.p + n, b + m * 8, m * 8, hipMemcpyHostToDevice, copy));
			HIP_CHECK(hipMemcpyAsync(gene.p + n, b + m * 16, m * 4, hipMemcpyHostToDevice, copy));
			HIP_CHECK(hipMemcpyAsync(aux.p + n, b + m * 20, m * 4, hipMemcpyHostToDevice, copy));
			HIP_CHECK(hipEventRecord(done[cur], copy));
			cur ^= 1;
			n += m;
		}
	}
	// The same for several host ranges that follow one another in the stream (dropest_push_reads_gather): they meet in ONE staging buffer
	// and leave with one set of copies -- a BAM reader's workers each hold a dense run of records, and a push per run cost ~50 us of
	// pointer queries, event waits and four small copies each.
	void push_segments(size_t n_seg, const uint64_t *const *h_cb, const uint64_t *const *h_umi, const uint32_t *const *h_gene, const uint32_t *const *h_aux,
	                   const uint64_t *counts) {
		size_t total = 0;
		for (size_t k = 0; k < n_seg; ++k) total += size_t(counts[k]);
		if (!total) return;
		if (total > (size_t(4) << 20)) { for (size_t k = 0; k < n_seg; ++k) push(h_cb[k], h_umi[k], h_gene[k], h_aux[k], size_t(counts[k])); return; }
		reserve(n + total);
		if (!copy) HIP_CHECK(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
		PinnedBuf<unsigned char> &st = stage[cur];
		if (!done[cur]) HIP_CHECK(hipEventCreateWithFlags(&done[cur], hipEventDisableTiming));
		else HIP_CHECK(event_wait(done[cur]));
		st.ensure(total * 24);
		unsigned char *b = st.p;
		size_t at = 0;
		for (size_t k = 0; k < n_seg; ++k) {
			const size_t m = size_t(counts[k]);
			if (!m) continue;
			std::memcpy(b + at * 8, h_cb[k], m * 8); std::memcpy(b + total * 8 + at * 8, h_umi[k], m * 8);
			std::memcpy(b + total * 16 + at * 4, h_gene[k], m * 4); std::memcpy(b + total * 20 + at * 4, h_aux[k], m * 4);
			at += m;
		}
		HIP_CHECK(hipMemcpyAsync(cb.p + n, b, total * 8, hipMemcpyHostToDevice, copy));
		HIP_CHECK(hipMemcpyAsync(umi.p + n, b + total * 8, total * 8, hipMemcpyHostToDevice, copy));
		HIP_CHECK(hipMemcpyAsync(gene.p + n, b + total * 16, total * 4, hipMemcpyHostToDevice, copy));
		HIP_CHECK(hipMemcpyAsync(aux.p + n, b + total * 20, total * 4, hipMemcpyHostToDevice, copy));
		HIP_CHECK(hipEventRecord(done[cur], copy));
		cur ^= 1;
		n += total;
	}
	// Device arrays copied to the tail of the store (dropest_push_reads_device without adoption): the reads stay ONE block, so a
	// view of the container before it is initialised (dropest_resident_reads) and the freeze have nothing to concatenate.
	// (device to device, on the context's own stream `on`: whatever wrote the caller's arrays on that stream is in front of the copies)
	void push_device(const uint64_t *d_cb, const uint64_t *d_umi, const uint32_t *d_gene, const uint32_t *d_aux, size_t count, hipStream_t on) {
		if (!count) return;
		reserve(n + count);
		hipLaunchKernelGGL(store_append_kernel, dim3(unsigned(std::min<size_t>((count + 255) / 256, 4096))), dim3(256), 0, on, reinterpret_cast<const u64 *>(d_cb),
		                   reinterpret_cast<const u64 *>(d_umi), d_gene, d_aux, cb.p + n, umi.p + n, gene.p + n, aux.p + n, count);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(stream_wait(on));   // the caller's buffers are free again (not waiting -- the BAM decoder's later work is queued on `on` too -- left the
		                              // last launch of a file unsubmitted for 23-28 ms: measured, dropped)
		n += count;
	}
	void wait() { if (copy) HIP_CHECK(stream_wait(copy)); }
	void clear() { wait(); n = 0; }
};

// decode a packed 2-bit code (include/dropest_amd.h) to text
inline std::string decode_code(u64 code, const std::vector<std::string> &side) {
	if (code & ESCAPE_BIT) {
		u64 k = code & ~ESCAPE_BIT;
		if (k >= side.size()) throw RangeError("escaped code refers to an unregistered side string");
		return side[k];
	}
	if (code == 0) return std::string();
	int len = (bit_length(code) - 1) / 2;
	std::string s(size_t(len), 'A');
	for (int i = 0; i < len; ++i) s[size_t(i)] = "ACGT"[(code >> (2 * (len - 1 - i))) & 3];
	return s;
}
inline bool encode_code(const std::string &s, u64 &code) {   // false: needs an escape
	if (s.empty() || s.size() > 31) return false;
	u64 c = 1;
	unsigned bad = 0;   // (no branch on a base's value: whitelists are millions of random barcodes)
	for (char ch : s) {
		const unsigned b = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 0x80u;
		bad |= b; c = (c << 2) | (b & 3u);
	}
	if (bad & 0x80u) return false;
	code = c;
	return true;
}

struct UmiOverride { u64 umi; u32 reads; uint8_t mark; u32 src_row; };   // src_row: molecule row whose quality sums it shows   // molecule of a group re-keyed by the N-UMI merge

// The cells a whitelist merge search runs over: the context's own cells, or (sharded runs) the real cells of every
// shard.  Device arrays are indexed by the universe's cell index; the callbacks serve the few host-side look-ups.
struct MergeUniverse {
	CbTable table{};                                   // barcode -> universe index
	const u64 *cell_cb = nullptr;                      // packed barcodes
	const u32 *n_genes = nullptr, *total_umis = nullptr;
	const u32 *real_index = nullptr;                   // universe index -> index in the caller's cell list
	bool any_escaped = false;
	std::function<int32_t(u32)> base_total_umis;       // TOTAL_UMIS stat of base f (position in the searched list)
	std::function<u64(u32)> barcode_code;              // packed barcode of a universe cell
	std::function<std::string(u32)> base_barcode_text; // text of base f's barcode
	// host look-up of a barcode code among the universe's cells (single context only; whitelists of more parts than the device
	// kernel takes search on the host): universe index or -1, with the cell's n_genes, TOTAL_UMIS and index in `real`
	std::function<long(u64, u32 &, int32_t &, u32 &)> find_cell;
};
struct MergeSearch {                                   // neighbour search result (merge_host.h)
	u32 F = 0, ntot = 0;
	size_t lds = 0;
	std::vector<u32> cells, cnt, off, fcell, fumis, fridx, self_ridx;
	std::vector<u32> pair_base /* position f */, pair_cand, pair_umis, pair_ridx, pair_first;
	std::vector<std::vector<u32>> host_order;          // host search: the candidates of every base in the reference's own order
	DevBuf<WlBase> d_bases;
	DevBuf<u32> d_cells, d_cnt, d_lvl, d_off, d_fcell, d_fumis, d_fridx, d_todo;   // device side of the search, kept with the object
	WlArgs args{};
};

// Host loops over millions of cells (C3 size: 2.5 M real-candidate cells): contiguous ranges on worker threads.
// fn(begin, end, worker); ranges are in worker order, so per-worker results can be concatenated in input order; two calls with
// the same n and limits cut the same ranges.  The workers are persistent (one pool per process, created on first use): a C3 pass
// makes about twenty such calls and creating eight threads for each cost more than some of the loops.  One job at a time; a
// caller that finds the pool busy (several shards of one process) runs its loop inline.
class HostPool {
	std::vector<std::thread> threads;
	std::mutex m, job_m;
	std::condition_variable cv_go, cv_done;
	std::function<void(unsigned)> job;
	uint64_t generation = 0;
	unsigned pending = 0, active = 0;
	bool stop = false;
	std::string error;
	HostPool() {
		const unsigned n = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
		for (unsigned t = 0; t < n; ++t)
			threads.emplace_back([this, t] {
				uint64_t seen = 0;
				for (;;) {
					std::function<void(unsigned)> fn;
					{
						std::unique_lock<std::mutex> lk(m);
						cv_go.wait(lk, [&] { return stop || generation != seen; });
						if (stop) return;
						seen = generation;
						if (t >= active) continue;
						fn = job;
					}
					try { fn(t); } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(m); error = e.what(); } catch (...) { std::lock_guard<std::mutex> lk(m); error = "unknown error"; }
					{ std::lock_guard<std::mutex> lk(m); if (--pending == 0) cv_done.notify_all(); }
				}
			});
	}
public:
	static constexpr unsigned MAX = 16;
	~HostPool() {
		{ std::lock_guard<std::mutex> lk(m); stop = true; }
		cv_go.notify_all();
		for (auto &t : threads) t.join();
	}
	static HostPool &get() { static HostPool p; return p; }
	unsigned size() const { return unsigned(threads.size()); }
	// false: the pool is busy with somebody else's job
	bool run(unsigned workers, const std::function<void(unsigned)> &fn) {
		std::unique_lock<std::mutex> one(job_m, std::try_to_lock);
		if (!one.owns_lock()) return false;
		{
			std::lock_guard<std::mutex> lk(m);
			job = fn; active = workers; pending = workers; ++generation; error.clear();
		}
		cv_go.notify_all();
		std::unique_lock<std::mutex> lk(m);
		cv_done.wait(lk, [&] { return pending == 0; });
		if (!error.empty()) throw std::runtime_error(error);
		return true;
	}
};
template <class F>
inline unsigned parallel_ranges(size_t n, F &&fn, size_t min_per_worker = 100000, unsigned max_workers = 8) {
	HostPool &pool = HostPool::get();
	unsigned workers = unsigned(std::min<size_t>(std::min<size_t>(max_workers, pool.size()), std::max<size_t>(1, n / std::max<size_t>(1, min_per_worker))));
	if (workers <= 1) { fn(size_t(0), n, 0u); return 1; }
	// (the number of workers -- hence the ranges -- must not depend on whether the pool was free: callers pair two calls)
	if (pool.run(workers, [&](unsigned w) { fn(n * w / workers, n * (w + 1) / workers, w); })) return workers;
	for (unsigned w = 0; w < workers; ++w) fn(n * w / workers, n * (w + 1) / workers, w);   // pool busy: the same ranges, one after the other
	return workers;
}

// glibc's rand() (random_r, TYPE_3: r[i] = r[i-31] + r[i-3], 310 outputs discarded, result >> 1), restated so that every
// container owns its sequence: MergeUMIsStrategySimple seeds the PROCESS generator with 42 in its constructor
// (MergeUMIsStrategySimple.cpp:15-19) and fix_n_umi_with_random draws rand() % 4 per 'N' (MergeUMIsStrategyAbstract.cpp:11-23);
// a library must not touch the host application's generator, and a second container (or a second pass over the same
// reads) must reproduce the first.  Checked against libc's srand / rand by tests/test_cabi_cpu.py.
struct GlibcRand {
	uint32_t ring[34];
	int pos = 0;              // index of the oldest entry (= r[i-34])
	uint64_t drawn = 0;
	explicit GlibcRand(unsigned seed = 1) { this->seed(seed); }
	void seed(unsigned s) {
		int32_t r[34];
		r[0] = int32_t(s ? s : 1u);
		for (int i = 1; i < 31; ++i) {
			const int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
			int64_t w = 16807 * lo - 2836 * hi;
			if (w < 0) w += 2147483647;
			r[i] = int32_t(w);
		}
		for (int i = 0; i < 31; ++i) ring[i] = uint32_t(r[i]);
		for (int i = 31; i < 34; ++i) ring[i] = ring[i - 31];
		pos = 0; drawn = 0;
		for (int i = 0; i < 310; ++i) step();
	}
	uint32_t step() {            // a[i] = a[i-31] + a[i-3] over a window of the last 34 values
		const uint32_t v = ring[(pos + 3) % 34] + ring[(pos + 31) % 34];
		ring[pos] = v;
		pos = (pos + 1) % 34;
		return v;
	}
	int next() { ++drawn; return int(step() >> 1); }
	void skip_to(uint64_t n_drawn) { while (drawn < n_drawn) next(); }   // forward only
};

struct HostCell {   // host mirror of one REAL-candidate cell (n_genes >= min_genes_before_merge at init)
	u32 id;
	CellRowPod row;         // sizes as of the last device update + stat adjustments (row.barcode = packed code)
	bool merged = false, excluded = false;
};

}  // namespace dropest

// simple_merge.h: common_umigs_per_cell of ALL bases at once: key = base << 32 | other, ascending; ed = device distance or 0xFFFFFFFF
struct SimplePairs { std::vector<uint64_t> key; std::vector<uint32_t> cnt, ed; std::vector<double> prob; /* -M: per pair, 2 = not a neighbour */ };

struct dropest_ctx {
	using u64 = dropest::u64;
	using u32 = dropest::u32;

	dropest_cfg cfg;
	std::string barcodes_file, match_levels;
	u32 query_mask = 0;
	u32 min_before = 0, min_after = 0;
	hipStream_t stream = nullptr;

	std::vector<dropest::ReadChunk> chunks;
	dropest::ReadStore store;        // reads pushed from host memory: one chunk of `chunks`, growing in place
	long store_chunk = -1;           // its index in `chunks`
	uint64_t n_reads = 0;
	dropest::DevBuf<u64> cat_cb, cat_umi;
	dropest::DevBuf<u32> cat_gene, cat_aux;
	const u64 *d_cb = nullptr, *d_umi = nullptr;
	const u32 *d_gene = nullptr, *d_aux = nullptr;
	std::vector<std::string> side;
	// Reads that came out of a sharded run's exchange as 12-byte records (k_cbhash.h: ReadPack): d_cb and d_umi both point at w0, d_gene
	// at w1.  The hot kernels (cb_sample, cb_insert, the sampled statistics, build_keys) read the records as they are; everything else
	// calls need_columns() first, which has the owner restore the four columns.
	dropest::ReadPack rpack{};
	std::function<void()> unpack_reads;
	void need_columns() {
		if (!rpack.on()) return;
		if (!unpack_reads) throw dropest::InvalidError("internal: packed reads without an unpacker");
		unpack_reads();
		rpack = dropest::ReadPack{};
	}

	bool initialized = false, merged = false;
	dropest::GlibcRand rng;          // the container's own rand() sequence (random fills of N-UMIs)
	void reseed_rng() { rng.seed(cfg.umi_merge_kind == DROPEST_UMI_MERGE_SIMPLE ? 42u : 1u); }

	// ---- device results ----
	dropest::DevBuf<dropest::CbSlot> t_slots;
	dropest::DevBuf<u32> slot;
	dropest::DevBuf<u64> hot_key;            // barcodes the 1/64 sample saw most often (k_cbhash.h): LDS table of cb_insert / build_keys
	dropest::DevBuf<u32> hot_slot;
	u32 n_hot = 0;
	dropest::CbTable table{};
	uint64_t forced_table_capacity = 0;      // a pass whose table came out too full rebuilds it larger (this pass only)
	u32 n_cells = 0;
	dropest::DevBuf<u64> cell_cb;
	dropest::DevBuf<u32> cell_first;
	dropest::KeyLayout layout{};
	int umi_clean_bits = 0;          // bits of a clean UMI code inside the key
	u32 wanted_bits[3] = {0, 0, 0};  // cell / gene / UMI field widths of the last layout plan (also when it did not fit 64 bits)
	bool umi_sentinel_stripped = false;
	// UMI dictionary (k_umidict.h): when gene + UMI fields alone reach 64 bits the UMI field of the key is the UMI's rank among
	// the distinct clean UMIs of the gene-bearing reads; the key kernels then read `umi_ranked` in place of the UMI column.
	// umi_dict_mode: 0 = only then, 1 = also before a key wider than 64 bits is refused (dropest_ctx_split not needed), 2 = always (tests).
	int umi_dict_mode = 0;
	bool umi_dict_on = false;                // this pass's keys carry ranks
	u32 umi_dict_n = 0;
	dropest::DevBuf<u64> umi_dict, umi_ranked;
	std::vector<u64> umi_dict_host;          // ascending codes: unmap_umi / map_umi
	const u64 *umi_key_column() const { return umi_dict_on ? umi_ranked.p : d_umi; }
	void build_umi_dict(const std::function<void(dropest::DevBuf<u64> &, u32 &)> *across = nullptr);
	bool umi_dict_wanted(int gene_bits, int cell_bits) const;
	// the sorted distinct values of d[0, n) (device, any order, may repeat) in place: n becomes their number (k_umidict.h kernels; uses keys_a / keys_b)
	void sort_unique_u64(dropest::DevBuf<u64> &d, u32 &n);
	bool map_umi(u64 api_code, u64 &field) const;   // inverse of unmap_umi; false: the code has no place in this pass's key layout
	bool map_umi_or_add(u64 api_code, u64 &field);  // the same for the public mutators: a new clean UMI joins the dictionary
	dropest::IngestStats ingest{};
	dropest::GlobalCounters counters{};

	dropest::DevBuf<u64> keys_a, keys_b;     // sort ping-pong
	dropest::DevBuf<u32> vals_a, vals_b;     // values: u32, or bytes in the same storage (layout.val_bytes)
	bool chr_from_gene = false;              // chromosome is a function of the gene: sort layouts VB 0 / 1
	// The key layout planned from the statistics of a SAMPLE of the reads (large single-context passes): cb_insert then reads
	// the barcodes only, the exact statistics ride along with build_keys and the plan is checked against them afterwards.
	bool lazy_stats = false;
	dropest::DevBuf<u32> gene_chr;           // [GENE_CHR_CAP] gene id -> chromosome id (GENE_CHR_UNSET if never counted)
	dropest::DevBuf<u32> mol_exon, mol_intron, mol_exon2, mol_intron2, cg_exon, cg_intron;

	u32 n_mol = 0;
	dropest::DevBuf<u64> mol_key;
	dropest::DevBuf<u32> mol_reads, mol_mark;
	u32 n_cg = 0;
	dropest::DevBuf<u64> cg_key;
	dropest::DevBuf<u32> cg_mol_begin, cg_n_all, cg_n_req, cg_reads_all, cg_reads_req;
	dropest::DevBuf<u32> cell_cg_count;
	dropest::DevBuf<u32> cell_cg_begin, cell_n_genes, cell_req_genes, cell_req_umis, cell_total_umis, cell_total_reads;
	u32 n_chr_rows = 0;
	dropest::DevBuf<u64> chr_row_key;
	dropest::DevBuf<u32> chr_exon, chr_intron, chr_inter;

	// CB merge
	dropest::Whitelist wl;
	dropest::DevBuf<dropest::WlEntry> d_wl[dropest::WL_MAX_PARTS];
	dropest::DevBuf<u64> d_wl_code[dropest::WL_MAX_PARTS];            // packed copies of the whitelist parts (k_merge.h: wl_edit_distance_code)
	dropest::DevBuf<dropest::WlTabRow> d_wl_tab[dropest::WL_MAX_PARTS];   // neighbour tables: one row per possible value of a part (k_merge.h)
	bool wl_tab_ok = false;
	dropest::DevBuf<u64> mol_key2;           // re-keyed molecule table (swapped in after a merge)
	dropest::DevBuf<u32> mol_reads2, mol_mark2, remap;
	std::unordered_map<u32, u32> reassign;   // merged cell -> final target (MergeStrategyBase cb_reassign_targets, sparse)
	// (cell, gene) groups rewritten by the N-UMI merge: key = molecule key >> umi_bits, value = the group's molecules
	std::unordered_map<u64, std::vector<dropest::UmiOverride>> umi_overrides;

	// pinned staging for the small device->host read-backs of the hot path (pageable copies cost ~0.5 ms each)
	dropest::PinnedBuf<unsigned char> h_stage;
	void fetch(void *dst, const void *d_src, size_t bytes);   // D2H through h_stage + stream sync
	dropest::PinnedBuf<unsigned char> h_up;
	void upload(void *d_dst, const void *src, size_t bytes);  // H2D of a large pageable array through pinned staging (returns when done)

	// scratch
	dropest::DevBuf<u32> tile_counts, tile_prefix, scalars, rs_hist, rs_row_total, rs_digit_base;
	dropest::DevBuf<u32> real_list, m_col_cell, m_col_start;
	dropest::DevBuf<dropest::CellRowPod> real_rows_dev;
	dropest::DevBuf<u32> sizes_dev;
	bool real_list_current = false;          // real_list holds the ids of `real`, in order
	bool real_pristine = false;              // no cell of `real` has changed since fetch_real_cells (flags, sizes, sums): real_rows_dev mirrors it
	dropest::DevBuf<u64> scalars64;
	// count matrices in CSC form: [0] filtered (cm), [1] raw (cm_raw); device staging + pinned host result
	struct MatrixResult {
		dropest::DevBuf<u32> d_row, d_val;
		dropest::PinnedBuf<u32> h_row, h_val;
		// the narrow form (dropest_count_matrix_csc_narrow): 16-bit rows / values, [0] overflow count, then positions, then values
		dropest::DevBuf<uint16_t> d_row16, d_val16;
		dropest::PinnedBuf<uint16_t> h_row16, h_val16;
		dropest::DevBuf<u32> d_ovf;
		dropest::PinnedBuf<u32> h_ovf;
		// the byte form (dropest_count_matrix_csc_bytes): row deltas / values of one byte, the value list above and a second list of exact rows
		dropest::DevBuf<uint8_t> d_drow8, d_val8;
		dropest::PinnedBuf<uint8_t> h_drow8, h_val8;
		dropest::DevBuf<u32> d_rovf;
		dropest::PinnedBuf<u32> h_rovf;
		int narrow = 0;                 // the form of the last emit as the caller sees it: 0 32-bit, 1 16-bit, 2 bytes
		u32 n_ovf = 0, n_rovf = 0;
		u32 rcap = 0, vcap = 0;         // capacities of the row / value lists of the last byte-form emit
		// 32-bit slots that travel as bytes (matrix_decode.h): the chunked copy's events and the job that widens into h_row / h_val
		bool wire = false;
		std::shared_ptr<dropest::DecodeJob> job, late_job;   // late_job: finished, but a decoding thread may not have left it yet
		std::chrono::steady_clock::time_point job_t0;
		// cm as a rider on cm_raw (round 6; k_misc.h: emit_values_on_rows_kernel): this slot's job reads ANOTHER slot's bytes and row slots.  The
		// base names its rider in `dependent` and settles it first; rider_out / rider_cnt: per column of the base, where it lands in this matrix.
		MatrixResult *dependent = nullptr;
		std::vector<u32> rider_out, rider_cnt, wire_chunk_end;
		dropest::DevBuf<u32> d_rider_out;
		void settle() {   // nobody reads or writes this slot's host buffers any more
			if (dependent) { MatrixResult *d = dependent; dependent = nullptr; d->settle(); }
			if (job) { (void)job->wait(); job->quiesce(); job.reset(); }
			if (late_job) { late_job->quiesce(); late_job.reset(); }
		}
		dropest::PinnedBuf<u32> h_flags;   // arrival flags the device raises between the chunks
		u32 wire_epoch = 0;
		std::vector<u32> colptr;
		uint64_t nnz = 0, ncols = 0;
	} mat[3];   // cm, cm_raw, and the filtered matrix under another mark query (emit_matrix_levels)
	dropest::DevBuf<dropest::IngestStats> d_ingest;
	dropest::DevBuf<dropest::GlobalCounters> d_counters;

	// ---- host state over real-candidate cells ----
	std::vector<dropest::HostCell> real;                 // ascending cell id
	std::vector<uint64_t> filtered;                      // cell ids, ascending compare_cells order (built lazily)
	std::vector<u32> filtered_ridx;                      // index in `real` of each filtered cell
	dropest::PinnedBuf<u64> sort_stage;                  // staging of sort_filtered's key columns / permutation
	dropest::DevBuf<u64> sort_cols;
	std::vector<u32> sort_idx;
	std::vector<int32_t> filtered_umis;   // TOTAL_UMIS of the cells of `filtered`, in that order (dense: the merge decision reads it per base)
	std::vector<uint64_t> sort_ids;   // cell ids in the order of sort_idx (dense: the gather of the ordered list reads this, not `real`)
	bool filtered_valid = false;
	u32 filtered_threshold = 0;
	int filtered_max_cells = -1;
	// host mirror of barcode -> cell id for dropest_cell_id_by_cb (FilteringBamProcessor.cpp:14-40 and user code ask per read): level 1 =
	// the real-candidate cells (from `real`, no device access at all), level 2 = every barcode of the pass (cell_cb fetched once, on the first
	// question about a barcode that is not a real cell's).  Valid for one pass: free_results() drops it.
	struct CbMirror {
		std::vector<u64> key; std::vector<u32> id; u64 mask = 0; int level = 0;
		void clear() { key.clear(); id.clear(); mask = 0; level = 0; }
		void build(size_t n, const std::function<void(size_t, u64 &, u32 &)> &item) {
			size_t cap = 16; while (cap < n * 2 + 2) cap <<= 1;
			key.assign(cap, 0ull); id.assign(cap, 0u); mask = cap - 1;
			for (size_t i = 0; i < n; ++i) {
				u64 k; u32 v; item(i, k, v);
				u64 h = dropest::mix64(k) & mask;
				while (key[h] != 0ull && key[h] != k) h = (h + 1) & mask;   // (packed codes carry a sentinel bit: never 0)
				key[h] = k; id[h] = v;
			}
		}
		long find(u64 k) const {
			if (key.empty()) return -1;
			for (u64 h = dropest::mix64(k) & mask;; h = (h + 1) & mask) { if (key[h] == k) return long(id[h]); if (key[h] == 0ull) return -1; }
		}
	} cb_mirror;
	long real_find(u32 cell_id) const {                  // index in `real` (binary search, ids ascending) or -1
		size_t lo = 0, hi = real.size();
		while (lo < hi) { size_t mid = (lo + hi) / 2; if (real[mid].id < cell_id) lo = mid + 1; else hi = mid; }
		return (lo < real.size() && real[lo].id == cell_id) ? long(lo) : -1;
	}
	u32 real_at(u32 cell_id) const {
		long i = real_find(cell_id);
		if (i < 0) throw dropest::RangeError("cell is not a real-candidate cell");
		return u32(i);
	}
	std::string barcode_of(const dropest::HostCell &h) const { return dropest::decode_code(h.row.barcode, side); }
	const std::vector<uint64_t> &filtered_cells();       // sorts on first use
	std::vector<std::pair<uint64_t, uint64_t>> merge_pairs;   // (source, target), target != source
	uint64_t n_real_now = 0;

	// ---- instrumentation ----
	bool profiling = false;
	std::string profile_only;      // non-empty: only launches whose stat name starts with this are timed, host stages are not
	std::map<std::string, dropest::KernelStat> stats;
	struct Pending { std::string name; hipEvent_t a, b; double bytes; };
	std::vector<Pending> pending;
	std::vector<hipEvent_t> event_pool;

	// wall-clock time of a host stage (only while profiling); reported as "host:<name>" in the kernel stats
	bool stage_sync = getenv("DROPEST_STAGE_SYNC") != nullptr;   // host-stage timers synchronise the stream (tuning aid, slows the pass)
	struct HostStage {
		dropest_ctx *c; const char *name; std::chrono::steady_clock::time_point t0;
		HostStage(dropest_ctx *ctx, const char *n) : c(ctx), name(n) {
			if (c->stage_sync && c->profiling) (void)dropest::stream_wait(c->stream);   // diagnostic: charge queued work to the stage that queued it
			t0 = std::chrono::steady_clock::now();
		}
		~HostStage() {
			if (!c->profiling || !c->profile_only.empty()) return;
			if (c->stage_sync) (void)dropest::stream_wait(c->stream);
			auto &s = c->stats[std::string("host:") + name];
			s.launches++;
			s.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		}
	};

	~dropest_ctx();
	void init_from_cfg(const dropest_cfg &c);

	template <class F> void timed(const char *name, double bytes, F &&launch);
	void collect_timings();

	void concat_chunks();
	void free_results();
	// gives the large per-read / per-molecule tables back (the reads stay): a context that hands its reads to split shards
	void release_tables() {
		t_slots.release(); slot.release(); keys_a.release(); keys_b.release(); vals_a.release(); vals_b.release();
		mol_key.release(); mol_reads.release(); mol_mark.release(); mol_key2.release(); mol_reads2.release(); mol_mark2.release();
		mol_exon.release(); mol_intron.release(); mol_exon2.release(); mol_intron2.release();
	}
	void run_set_initialized();
	void run_merge_and_filter();

	void build_cb_table();
	void assign_cell_ids();
	void plan_key_layout();
	// allow_fused: the keys may be built and partitioned into the coarse regions of the splitter sort in one pass (k_keyscatter.h); the
	// fall-backs that need the plain key array (counting partitions, the LSD sort) say no
	void build_keys(bool with_stats = false, bool allow_fused = true);
	bool build_keys_fused(bool with_stats);   // false: not applicable for this pass (layout, size, LDS)
	void ss_splitters_from_sample(u32 n_sample, u32 os, u32 Ff, u32 F2, u64 varying_of_sample);   // ss_sample_a holds the sample -> ss_fine / ss_coarse
	bool keys_in_l1 = false;                  // keys_b / vals_b hold the keys in their coarse regions, ss_cursors their fills
	dropest::DevBuf<u32> ks_flag;
	u32 main_sort_passes = 0, main_sort_kind = 0;   // kind: 0 LSD radix sort, 1 splitter sort
	// exclusive scan of n counters (tile counts) + their total: one workgroup for short arrays, chunk sums + prefix for long ones (2e5 tile
	// counts at C3 size took one workgroup 0.28 ms, five times a pass)
	void scan_counts(const u32 *in, u32 *out, u32 n, u32 *total_out);
	dropest::DevBuf<u32> scan_chunk;
	void radix_sort(u64 *&keys, u32 *&vals, u64 *&keys_alt, u32 *&vals_alt, u32 n, u64 varying_mask, int val_bytes = 4,
	                const char *stat_prefix = nullptr);
	// splitter sort (k_ssort.h): sample, splitters, partition scratch, look-back words
	dropest::DevBuf<u64> ss_sample_a, ss_sample_b, ss_fine, ss_coarse;
	dropest::DevBuf<u32> ss_base1, ss_cnt2, ss_bucket_base, ss_bucket_cnt, ss_tmp, ss_n_loc, ss_prefix, ss_chunk, ss_big_list;
	// (sharded runs) the reads arrive in chunks -- chunk k = piece k of every source's block, ranges of the SAME read arrays; ev fires when it
	// has landed --: the barcode table is built chunk by chunk while the later ones travel (build_cb_table; csrc/shard_run.h sets them per step)
	struct RecvChunk { hipEvent_t ev; dropest::CbRanges rg; };
	std::vector<RecvChunk> recv_chunks;
	double hot_coverage = 0;            // share of the sampled reads whose barcode is on the hot list (a lower bound)
	dropest::DevBuf<u32> ss_cg_loc, ss_cg_cnt, ss_cg_prefix;   // (cell, gene) heads per fine bucket: the compaction makes the (cell, gene) table on its way
	bool cg_from_sort = false;                                 // this pass's (cell, gene) table came out of the splitter sort's compaction
	dropest::DevBuf<u32> ss_cursors;   // partitions by reservation: the regions' cursors (k_ssort.h)
	bool ss_no_reserve = false;        // a region overflowed in this pass: the counting partitions from here on
	bool splitter_sort_reduce();   // false: not applicable / fell back, the caller runs the LSD sort + seg_reduce
	void reduce_all();
	void reduce_molecules_to_cell_gene();
	void reduce_cell_gene_to_cells();
	void refresh_real_rows();
	void upload_whitelist();
	void search_merge_candidates(const std::vector<u32> &cells, const dropest::MergeUniverse &U, dropest::MergeSearch &S);
	void search_merge_candidates_host(const std::vector<u32> &cells, const dropest::MergeUniverse &U, dropest::MergeSearch &S);
	void build_merge_pairs(const std::vector<u32> &cells, dropest::MergeSearch &S);
	void decide_merge_targets(const dropest::MergeUniverse &U, dropest::MergeSearch &S, const std::vector<u32> &inter,
	                          std::vector<long> &targets, std::vector<u32> &target_ridx);
	std::vector<std::vector<u32>> replay_candidate_orders(const dropest::MergeUniverse &U, dropest::MergeSearch &S,
	                                                      const std::vector<u32> &need_order);
	std::vector<u32> pair_intersections(const std::vector<u32> &pair_base_cell, const std::vector<u32> &pair_cand_cell);
	void pair_intersections(const std::vector<u32> &pair_base_cell, const std::vector<u32> &pair_cand_cell, std::vector<u32> &inter);
	// PoissonRealBarcodesMergeStrategy (poisson_merge.h)
	std::vector<double> poisson_expected_intersections(const std::vector<u32> &pair_base_cell, const std::vector<u32> &pair_cand_cell);
	// the estimator's tables (poisson_merge.h): built from the UMI counts of this context, or of all shards (merge_shard.h)
	struct PoissonTables { dropest::DevBuf<double> p, mult, np; dropest::DevBuf<u64> adj; u32 n_classes = 0, max_size = 0; bool ready = false; } ptab;
	void poisson_local_umis(u32 &kept, u32 &max_size);
	void poisson_build_tables(const u32 *d_counts, u32 n_counts, double total, u32 max_size);
	void poisson_expected_from_keys(const u32 *d_off, u32 NP, const u64 *d_keys, u32 NK, double *expected_host);
	// sharded -M (merge_shard.h): this shard's dense UMI histogram; the tables from the summed one; expected sizes of pairs with shipped base rows
	void shard_merge_umi_histogram(dropest::DevBuf<u32> &hist, uint64_t &kept, u32 &max_size);
	void shard_merge_set_umi_distribution(const u32 *d_counts, uint64_t n_counts, uint64_t kept_total, u32 max_size);
	void shard_merge_expected(uint64_t n_pairs, const uint32_t *cand_local, const uint64_t *base_begin, const uint64_t *base_end,
	                          const uint64_t *d_base_low, double *expected);
	void shard_merge_decide_poisson(const uint32_t *inter, const double *expected, int64_t *target_g);
	void decide_poisson_targets(const dropest::MergeUniverse &U, dropest::MergeSearch &S, const std::vector<u32> &inter,
	                            const std::vector<double> &expected, std::vector<long> &targets, std::vector<u32> &target_ridx);
	std::vector<long> compute_merge_targets(const std::vector<u32> &cells, const std::vector<u32> &ridx,
	                                        std::vector<u32> *target_ridx = nullptr);
	void compute_merge_targets(const std::vector<u32> &cells, const std::vector<u32> &ridx, std::vector<long> &targets,
	                           std::vector<u32> *target_ridx);
	dropest::DevBuf<u32> cell_real_index;   // [n_cells] cell id -> index in `real` (0xFFFFFFFF otherwise)
	// Host scratch of the CB merge, kept across passes.  At C3 size the merge works on a dozen arrays of 2.4 M entries; as
	// locals they were 300 MB of fresh anonymous memory per pass -- every page faulted in again, tens of milliseconds of
	// kernel time that swing with the load of the host.  Kept, they are written into warm pages.
	struct MergeScratch {
		dropest::MergeSearch S;
		std::vector<u32> pb, inter, tr, cells, ridx, target_ridx, cur, rank, src, tgt32, lists, ids_dense;
		std::vector<long> targets;
		std::vector<int64_t> tgt;
		std::vector<int32_t> reads, umis;
		std::vector<uint8_t> excl;
		dropest::DevBuf<u32> d_pb, d_pc, d_inter, d_src, d_tgt;   // device side, kept as well (a hipFree synchronises the device)
		dropest::DevBuf<dropest::PairRange> d_pr;
	} ms;
	void run_cb_merge_real();
	void run_cb_merge_simple();                  // SimpleMergeStrategy (simple_merge.h)
	struct SimpleReplayInput { std::vector<u64> query; std::vector<std::vector<u32>> in_order; };
	void simple_pair_table(const u64 *sorted, u32 n_valid, int cell_bits, const u32 *d_cell_size, const u64 *d_cell_code, SimplePairs &P,
	                       dropest::DevBuf<u64> *d_run_key, dropest::DevBuf<u32> *d_run_cnt, u32 *n_runs = nullptr);
	void simple_pairs_to_host(const u64 *d_run_key, const u32 *d_run_cnt, u32 runs, const u64 *d_cell_code, SimplePairs &P);
	void simple_replay_local(const std::vector<u32> &bases, SimpleReplayInput &R, bool globalize);
	void run_cb_merge_all();                     // MergeAllMergeStrategy (merge_all.h)
	void merge_all_targets(std::vector<u64> code, const std::vector<int32_t> &umis, const std::function<std::string(u32)> &text,
	                       const std::vector<u32> *bases, std::vector<u32> &target_pos);
	// public mutators of the container (mutate_host.h)
	std::unordered_set<u32> extra_excluded;      // excluded cells outside the host mirror of real-candidate cells
	std::unordered_set<u32> explicit_sources;    // sources of dropest_merge_cells (not part of the strategy's merge targets)
	void clear_strategy_pairs() {                // a strategy starts over; pairs merged by hand stay (their molecules have moved)
		if (explicit_sources.empty()) { merge_pairs.clear(); return; }
		std::vector<std::pair<uint64_t, uint64_t>> keep;
		for (auto const &pr : merge_pairs) if (explicit_sources.count(u32(pr.first))) keep.push_back(pr);
		merge_pairs.swap(keep);
	}
	void mutate_exclude_cell(u32 cell);
	void mutate_merge_cells(u32 src, u32 tgt);
	void mutate_merge_umis(u32 cell, u32 gene, uint64_t n, const uint64_t *src, const uint64_t *tgt);
	void mutate_add_umi_to_cell(u32 cell, u32 gene, uint64_t umi_code, u32 mark, const uint8_t *quality, u32 quality_length);
	u32 n_qsum_rows = 0;                        // rows of mol_qsum (molecules at initialisation + reads added by add_umi_to_cell)
	void emit_matrix_levels(u32 query_mask, bool reads_output);   // get_count_matrix_filtered(container, query)
	// sharded runs (merge_shard.h): ingest / merge phases with collectives between them
	struct ShardMerge;
	std::shared_ptr<ShardMerge> shard;
	bool ingested = false, external_merge_done = false;
	void run_ingest();
	void shard_export_rows(const std::vector<u32> &local_cells);   // merge_shard.h
	void shard_merge_begin_free();                                    // whitelist-free merges across shards: an empty ShardMerge
	void shard_merge_search(uint64_t n_global, const uint64_t *g_barcode, const uint32_t *g_n_genes, const int32_t *g_total_umis,
	                        uint64_t n_bases, const uint32_t *base_g, const uint32_t *base_local, uint64_t *n_pairs);
	void shard_merge_intersect(uint64_t n_pairs, const uint32_t *cand_local, const uint64_t *base_begin, const uint64_t *base_end,
	                           const uint64_t *d_base_low, uint32_t *inter);
	void shard_merge_decide(const uint32_t *inter, int64_t *target_g);
	void shard_merge_quality_import(uint64_t n_local, const uint32_t *local_rank, const uint32_t *d_import_rank, const uint32_t *d_import_q);
	const u32 *reagg_import_prio = nullptr;     // (device) merge-order places of the rows a sharded merge appended, for the next re-aggregation
	u32 reagg_import_from = 0, reagg_import_n = 0;
	void shard_merge_finish(uint64_t n_local, const uint32_t *local_id, const uint8_t *excluded, const uint8_t *merged_away,
	                        const int32_t *total_reads, const int32_t *total_umis, uint64_t n_moves, const uint32_t *move_src,
	                        const uint32_t *move_tgt, uint64_t n_import, const uint32_t *d_cell, const uint64_t *d_low,
	                        const uint32_t *const d_cols[4]);
	void reaggregate_after_merge();
	void run_umi_merge_simple();
	std::vector<u32> umi_first_positions(const std::vector<u64> &sorted_codes);
	// sharded runs (shard_run.h): the two places where the N-UMI merge needs the other shards
	struct ShardHooks {
		std::function<std::vector<u64>(const std::vector<u64> &)> first_seen_global;   // sorted UMI codes -> smallest global ordinal each
		std::function<std::vector<u64>(const std::vector<u32> &, const std::vector<u32> &, const std::vector<u32> &)> rng_offsets;   // (cell first read, gene, draws) per group -> offset in the one rand() sequence
		// -u: the device table UMI code -> position of its first read on THIS shard (0xFFFFFFFF = never) becomes, in place, the
		// table UMI code -> rank of its first read in the WHOLE stream (the reference's UMI index order), equal on every shard
		std::function<void(u32 *, size_t)> globalize_umi_first;
	};
	std::shared_ptr<ShardHooks> hooks;
	void emit_columns_device(bool filtered_m, bool reads_output, const std::vector<u32> &col_cell, const std::vector<u32> &col_start, uint64_t nnz, bool wait = true);
	struct GatheredGroups { std::vector<u32> size, off, begin, hr, hm, hfirst; std::vector<u64> hk; };   // begin: first molecule row of the group
	void umi_gather_groups(const std::vector<u32> &groups, GatheredGroups &G, const u32 *d_first_table);
	void umi_patch_groups(const std::vector<u32> &p_idx, const std::vector<u32> &p_all, const std::vector<u32> &p_req,
	                      const std::vector<u32> &p_rreq, const std::unordered_map<u32, int> &umis_removed);
	void run_umi_merge_directional();            // -u (umi_directional_host.h)
	void reaggregate_from_keys(u64 varying_mask, bool sorted_already = false); // keys_a / vals_a hold the re-keyed molecule table
	// split + sort the changed rows + merge (k_mergepath.h); d_remap: the new keys are made on the fly from the cell -> target table
	bool resort_changed_rows(u64 varying_mask, const u32 *d_remap = nullptr, int cell_shift = 0, u64 *varying_out = nullptr);
	u32 mol_sorted_rows = 0xFFFFFFFFu;            // rows of the molecule table below this index are sorted (a sharded merge appends behind)
	dropest::DevBuf<u64> mp_bk, mp_bk2;
	dropest::DevBuf<u32> mp_bv, mp_bv2, mp_astart;
	// UMI quality sums (quality.h)
	dropest::DevBuf<uint8_t> umi_qual;          // [qual_reads][qual_len], read order
	dropest::DevBuf<uint8_t> umi_qual_lens;     // [qual_reads] (qual_var) the length of every read's string, <= qual_len
	bool qual_var = false;                      // dropest_set_umi_qualities_var: strings of several lengths, rows padded to qual_len
	u32 qual_stride() const { return (qual_len + 2u) & ~1u; }   // words of a sums row: the sums, padding to whole pairs, the molecule's length last
	u32 qual_len = 0;
	uint64_t qual_reads = 0;
	bool have_qual = false;
	u32 n_mol_at_init = 0;
	dropest::DevBuf<u32> mol_qsum, mol_qrow, mol_qrow2;   // sums per ORIGINAL molecule row; current row -> original row
	const u32 *reagg_prio = nullptr;            // per old molecule row, for the next reaggregate_from_keys (device)
	dropest::DevBuf<u32> reagg_prio_buf;
	std::vector<u32> merge_rank;                // per cell id: position in its target's merge order (0 = not merged away)
	void accumulate_umi_qualities();
	void requality_after_fold(const u64 *sorted_key, const u32 *old_row, u32 n_old, const u64 *new_key, u32 n_new);
	void fetch_quality_rows(const std::vector<u32> &rows, uint32_t *out, uint32_t *out_len = nullptr);
	dropest::DevBuf<u32> umi_first;
	void fetch_real_cells(bool at_init = false);
	void request_filtered(u32 genes_threshold, int max_cells);   // CellsDataContainer::update_filtered_gene_counts, lazily
	void sort_filtered(u32 genes_threshold, int max_cells);
	void emit_matrix(bool filtered_m, bool reads_output, bool to_host = true, int form = 0, bool direct = false);
	bool narrow_possible() const;
	void matrix_outputs(MatrixResult &M, uint64_t nnz, int form, bool to_host, dropest::MatrixArgs &a);
	void matrix_copy_out(MatrixResult &M, uint64_t nnz, hipStream_t st);
	void matrix_finish_overflow(MatrixResult &M, hipStream_t st);
	// 32-bit slots over the wire as bytes: true when this matrix takes that way (large enough, not switched off)
	bool wire_wanted(uint64_t nnz, int form, bool to_host) const;
	bool matrix_wire = true;        // dropest_set_matrix_wire
	// (sharded runs) where a shard's columns go: the 32-bit slots of the GLOBAL matrix (node-shared host memory) and each local column's
	// entry range there; the bytes themselves travel as the byte form of the shard's LOCAL matrix, contiguous, like one context's
	struct WireTarget {
		u32 *rows = nullptr, *vals = nullptr; const u32 *begin = nullptr, *end = nullptr; uint64_t global_nnz = 0;
		const unsigned long long *d_descr = nullptr;   // device: (local start, global start, length) of every local column
	};
	void wire_copy_and_decode(MatrixResult &M, uint64_t nnz, hipStream_t st, const WireTarget *target = nullptr);   // after the byte-form emit on `st`: lists, chunked copies, the job
	// emit (this stream) + lists + chunked copies (copy_st, behind an event) + the widening job of the columns of `col_cell` into the target's slots
	void ship_columns_to_slots(bool filtered_m, bool reads_output, const std::vector<u32> &col_cell, const std::vector<u32> &col_start, uint64_t nnz,
	                           const WireTarget &target, hipStream_t copy_st);
	hipEvent_t ev_ship = nullptr;
	bool wire_finish(MatrixResult &M);                                          // waits for the job; false: the lists overflowed (emit the slots directly)
	// columns of a count matrix from the host rows: cell id of every column, start of every column, number of entries
	void matrix_columns(bool filtered_m, std::vector<u32> &col_cell, std::vector<u32> &colptr, uint64_t &nnz);
	// cm_raw produced and copied to the host on a second stream while the caller goes on (dropest_prefetch_raw_matrix)
	struct RawPrefetch { bool valid = false, reads_output = false, in_flight = false; int narrow = 0; std::vector<u32> col_cell; } raw_pf;
	hipStream_t stream2 = nullptr;
	hipEvent_t ev_fork = nullptr, ev_raw = nullptr, ev_raw_cols = nullptr;   // ev_raw_cols: cm_raw's column arrays are on the device (a rider's emit reads them)
	bool emit_rider(bool reads_output, const std::vector<u32> &col_cell, uint64_t nnz);
	dropest::DevBuf<u32> m2_col_cell, m2_col_start, m2_col_list, m_col_list;
	std::vector<u32> m2_col_list_host, m_col_list_host;   // (host sides of asynchronous uploads: kept with the context)
	void launch_emit_bytes(dropest::MatrixArgs a, const std::vector<u32> &rows_per_column, dropest::DevBuf<u32> &list, std::vector<u32> &host_list, hipStream_t st);
	void prefetch_raw_matrix(bool reads_output, int form = 0, const dropest::CellRowPod *rows = nullptr, const dropest::u32 *ids = nullptr,
	                         dropest::u32 count = 0);   // form: 0 32-bit, 1 16-bit, 2 bytes; rows / ids: the real cells as fetch_real_cells just received them
	int auto_pf_form = -1;          // dropest_set_raw_matrix_prefetch: the form cm_raw will be asked for (-1: not announced)
	bool auto_pf_reads = false;
	bool merge_phase_changes_nothing() const;
	void invalidate_prefetch();
	u64 unmap_umi(u64 ucode) const;
};
