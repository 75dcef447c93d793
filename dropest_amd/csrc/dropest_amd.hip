// dropest_amd.hip -- MI355X-native Estimation hot path of dropEst: device pipeline + C-ABI (include/dropest_amd.h).
//
// Pipeline of dropest_set_initialized (all launches on the context's stream):
//   cb_insert        barcode hash build + first-seen ordinals            (CellsDataContainer.cpp:64-69)
//   cb_first_count / scan_small / cb_assign_ids   first-seen cell ids
//   build_keys       (cell | gene | UMI) sort keys + global read counters (CellsDataContainer.cpp:73-78, :309-327)
//   rs_hist / rs_scan / rs_scatter  x passes      LSD radix sort
//   seg_count / scan_small / seg_reduce  x 4      reads->molecules, reads->(cell,chr), molecules->(cell,gene), ->cells
//   flag_real / gather_cell_rows                  real cells to the host, which orders them (compare_cells)
// This file contains no CPU implementation of the path: without a GPU every entry point fails loudly.

#include "context.h"
extern "C" void dropest_bgzf_warm_up(void *stream);
#include "k_keyscatter.h"
#include "k_umidict.h"

#include <atomic>
#include <sys/syscall.h>
#include <unistd.h>
#include <functional>
#include <map>
#include <mutex>

using namespace dropest;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

template <class F>
static dropest_status guarded(F &&f) {
	try {
		f();
		return DROPEST_OK;
	} catch (const InvalidError &e) { g_last_error = e.what(); return DROPEST_ERR_INVALID;
	} catch (const RangeError &e) { g_last_error = e.what(); return DROPEST_ERR_RANGE;
	} catch (const UnsupportedError &e) { g_last_error = e.what(); return DROPEST_ERR_UNSUPPORTED;
	} catch (const IoError &e) { g_last_error = e.what(); return DROPEST_ERR_IO;
	} catch (const DeviceError &e) { g_last_error = e.what(); return DROPEST_ERR_DEVICE;
	} catch (const std::bad_alloc &) { g_last_error = "host allocation failed"; return DROPEST_ERR_DEVICE;
	} catch (const std::exception &e) { g_last_error = e.what(); return DROPEST_ERR_INVALID; }
}

static inline u32 div_up(uint64_t a, uint64_t b) { return u32((a + b - 1) / b); }

static void partition_by_owner_on(int device, hipStream_t st, const u64 *d_cb, const u64 *d_umi, const u32 *d_gene, const u32 *d_aux, uint64_t n64,
                                  u32 n_parts, u64 *d_out_cb, u64 *d_out_umi, u32 *d_out_gene, u32 *d_out_aux, u32 *d_out_idx, uint64_t *counts,
                                  void *d_scratch, uint64_t scratch_bytes);

// Grid of a grid-stride kernel: exactly as many workgroups as are resident at once (occupancy x CUs), at most `wanted`.
// A latency-bound kernel launched with a few more workgroups than fit pays a whole extra round for them (cb_insert with
// 2048 workgroups where 1792 fit ran its last 256 alone: measured 8192 waves against 7168 resident).
template <class K>
static u32 resident_grid(K kernel, int threads, u32 wanted) {
	int dev = 0, cus = 0, per_cu = 0;
	HIP_CHECK(hipGetDevice(&dev));
	HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
	HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0));
	return std::max<u32>(1u, std::min<u32>(wanted, u32(std::max(1, cus) * std::max(1, per_cu))));
}

// ------------------------------------------------------------------------------------------------
// context basics
// ------------------------------------------------------------------------------------------------
static u32 query_mask_from_code(const std::string &code) {   // UMI::Mark::get_by_code, UMI.cpp:112-154
	u32 m = 0;
	for (char c : code) {
		switch (c) {
			case 'e': m |= 1u << 2; break;
			case 'i': m |= 1u << 4; break;
			case 'E': m |= 1u << 3; break;
			case 'I': m |= 1u << 5; break;
			case 'B': m |= 1u << 6; break;
			case 'A': m |= 1u << 7; break;
			default: throw InvalidError(std::string("Unexpected gene match levels: ") + c);
		}
	}
	return m;
}

void dropest_ctx::init_from_cfg(const dropest_cfg &c) {
	cfg = c;
	barcodes_file = c.barcodes_file ? c.barcodes_file : "";
	match_levels = c.gene_match_levels ? c.gene_match_levels : "eEBA";
	cfg.barcodes_file = barcodes_file.c_str();
	cfg.gene_match_levels = match_levels.c_str();
	query_mask = query_mask_from_code(match_levels);
	if (c.min_genes_before_merge < 0 || c.min_genes_after_merge < 0) throw InvalidError("negative gene threshold");
	min_before = u32(c.min_genes_before_merge);
	min_after = std::max(u32(c.min_genes_after_merge), min_before);   // MergeStrategyAbstract.cpp:8-11
	if (c.merge_kind != DROPEST_MERGE_NONE && c.merge_kind != DROPEST_MERGE_REAL_BARCODES && c.merge_kind != DROPEST_MERGE_SIMPLE &&
	    c.merge_kind != DROPEST_MERGE_POISSON_REAL && c.merge_kind != DROPEST_MERGE_POISSON_SIMPLE && c.merge_kind != DROPEST_MERGE_ALL)
		throw InvalidError("unknown merge_kind");
	if (c.umi_merge_kind != DROPEST_UMI_MERGE_SIMPLE && c.umi_merge_kind != DROPEST_UMI_MERGE_DIRECTIONAL)
		throw InvalidError("unknown umi_merge_kind");
	if (c.max_umi_merge_edit_distance < 0) throw InvalidError("negative max_umi_merge_edit_distance");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
		throw DeviceError("no HIP device visible: the dropEst hot path has no CPU implementation");
	if (c.device < 0 || c.device >= ndev) throw InvalidError("device ordinal out of range");
	HIP_CHECK(hipSetDevice(c.device));
	HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
	// The stream's hardware queue -- and the null stream's, which synchronous copies and the BAM decoder's uploads use -- are made by their first
	// dispatch (~8 ms each, during which every other HIP call of the process waits): here, with the other start-up costs, not inside the first
	// window of a BAM file or the first pass.
	{
		void *scratch = nullptr;
		HIP_CHECK(hipMalloc(&scratch, 256));
		(void)hipMemsetAsync(scratch, 0, 256, stream);
		(void)hipMemsetAsync(scratch, 0, 256, nullptr);
		// (... and the first copy from pinned memory on the null stream sets up the DMA path: 7.5 ms, measured in the BAM reader's first window)
		// (a megabyte: small copies take another road)
		{
			void *pin = nullptr, *big = nullptr;
			const size_t mb = size_t(1) << 20;
			if (hipHostMalloc(&pin, mb, hipHostMallocDefault) == hipSuccess && hipMalloc(&big, mb) == hipSuccess) {
				(void)hipMemcpyAsync(big, pin, mb, hipMemcpyHostToDevice, nullptr);
				(void)hipMemcpyAsync(pin, big, mb, hipMemcpyDeviceToHost, nullptr);
				(void)hipStreamSynchronize(nullptr);
				// (and on the context's stream, at the sizes between the runtime's small and large copies too: the first copy of ~50 KB either way took
				// 7.5-9 ms in whichever BAM window made it)
				for (size_t bytes : {size_t(8) << 10, size_t(48) << 10, size_t(256) << 10, mb}) {
					(void)hipMemcpyAsync(big, pin, bytes, hipMemcpyHostToDevice, stream);
					(void)hipMemcpyAsync(pin, big, bytes, hipMemcpyDeviceToHost, stream);
				}
				(void)hipStreamSynchronize(stream);
			}
			if (big) (void)hipFree(big);
			if (pin) (void)hipHostFree(pin);
		}
		// (... and this library's kernels are loaded onto the device by the first launch of one of them: ~10 ms)
		hipLaunchKernelGGL(store_append_kernel, dim3(1), dim3(256), 0, stream, (const u64 *)nullptr, (const u64 *)nullptr, (const u32 *)nullptr, (const u32 *)nullptr,
		                   (u64 *)nullptr, (u64 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, size_t(0));
		(void)hipGetLastError();
		dropest_bgzf_warm_up(stream);      // (bgzf_api.hip's, for a BAM read on the device)
		(void)hipStreamSynchronize(stream);
		(void)hipStreamSynchronize(nullptr);
		(void)hipFree(scratch);
	}
	// MergeUMIsStrategySimple's constructor seeds rand() with 42 (MergeUMIsStrategySimple.cpp:15-19); MergeUMIsStrategyDirectional
	// does not seed, i.e. a fresh reference process draws from srand(1).  The container keeps its own restated generator.
	reseed_rng();
}

dropest_ctx::~dropest_ctx() {
	for (auto &p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
	for (auto e : event_pool) (void)hipEventDestroy(e);
	for (auto &M : mat) M.settle();
	if (stream2) { (void)stream_wait(stream2); (void)hipStreamDestroy(stream2); }
	if (ev_fork) (void)hipEventDestroy(ev_fork);
	if (ev_raw) (void)hipEventDestroy(ev_raw);
	if (ev_ship) (void)hipEventDestroy(ev_ship);
	if (stream) (void)hipStreamDestroy(stream);
}

template <class F>
void dropest_ctx::timed(const char *name, double bytes, F &&launch) {
	auto selected = [&]() {   // profile_only: prefixes separated by '|'
		if (profile_only.empty()) return true;
		size_t at = 0;
		while (at <= profile_only.size()) {
			size_t end = profile_only.find('|', at);
			if (end == std::string::npos) end = profile_only.size();
			if (end > at && std::strncmp(name, profile_only.c_str() + at, end - at) == 0) return true;
			at = end + 1;
		}
		return false;
	};
	// DROPEST_SYNC_TRACE=1 (hunting a device fault): every launch is named on stderr before it goes and waited for behind it -- the last
	// name without its "done" is the kernel that faulted
	static const bool sync_trace = getenv("DROPEST_SYNC_TRACE") != nullptr;
	if (sync_trace) {
		fprintf(stderr, "[launch %p] %s\n", static_cast<void *>(this), name);
		launch();
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(hipStreamSynchronize(stream));
		if (stream2) HIP_CHECK(hipStreamSynchronize(stream2));
		fprintf(stderr, "[done   %p] %s\n", static_cast<void *>(this), name);
		return;
	}
	if (!profiling || !selected()) {
		launch(); HIP_CHECK(hipGetLastError()); return;
	}
	auto get = [&]() {
		if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
		hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); return e;
	};
	hipEvent_t a = get(), b = get();
	HIP_CHECK(hipEventRecord(a, stream));
	launch();
	HIP_CHECK(hipGetLastError());
	HIP_CHECK(hipEventRecord(b, stream));
	pending.push_back(Pending{name, a, b, bytes});
}

void dropest_ctx::fetch(void *dst, const void *d_src, size_t bytes) {
	if (!bytes) return;
	h_stage.ensure(std::max<size_t>(bytes, 4096));
	HIP_CHECK(hipMemcpyAsync(h_stage.p, d_src, bytes, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(stream_wait(stream));
	if (bytes < (size_t(4) << 20)) std::memcpy(dst, h_stage.p, bytes);
	else   // tens of megabytes (candidate lists, cell rows at C3 size): the copy out of the staging buffer on several threads
		dropest::parallel_ranges(bytes, [&](size_t b, size_t e, unsigned) { std::memcpy(static_cast<char *>(dst) + b, h_stage.p + b, e - b); }, size_t(2) << 20, dropest::HostPool::MAX);
}

// Megabytes of a pageable host array to the device: staged through pinned memory on several threads, then one DMA (the
// runtime's own path for pageable memory runs at a fifth of the PCIe rate).  Returns when the copy is done.
void dropest_ctx::upload(void *d_dst, const void *src, size_t bytes) {
	if (!bytes) return;
	if (bytes < (size_t(1) << 20)) { HIP_CHECK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, stream)); HIP_CHECK(stream_wait(stream)); return; }
	h_up.ensure(bytes);
	dropest::parallel_ranges(bytes, [&](size_t b, size_t e, unsigned) { std::memcpy(h_up.p + b, static_cast<const char *>(src) + b, e - b); }, size_t(2) << 20, dropest::HostPool::MAX);
	HIP_CHECK(hipMemcpyAsync(d_dst, h_up.p, bytes, hipMemcpyHostToDevice, stream));
	HIP_CHECK(stream_wait(stream));
}

// DROPEST_TAIL_TRACE=1: host time stamps (us since the pass's first mark) along the end of a pass, to stderr
static void tail_mark(const char *what, bool restart = false) {
	static const bool on = getenv("DROPEST_TAIL_TRACE") != nullptr;
	if (!on) return;
	static thread_local std::chrono::steady_clock::time_point t0;
	const auto now = std::chrono::steady_clock::now();
	if (restart) t0 = now;
	fprintf(stderr, "[tail] %9.1f us  %s\n", std::chrono::duration<double, std::micro>(now - t0).count(), what);
}

void dropest_ctx::collect_timings() {
	if (pending.empty()) return;
	HIP_CHECK(stream_wait(stream));
	for (auto &p : pending) {
		float ms = 0;
		HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
		auto &s = stats[p.name];
		s.launches++; s.ms += ms; s.bytes += p.bytes;
		event_pool.push_back(p.a); event_pool.push_back(p.b);
	}
	pending.clear();
}

void dropest_ctx::concat_chunks() {
	if (d_cb) return;
	if (n_reads == 0) return;
	store.wait();   // pushed batches still on their way over PCIe
	if (store_chunk >= 0) {
		dropest::ReadChunk &c = chunks[size_t(store_chunk)];
		c.p_cb = store.cb.p; c.p_umi = store.umi.p; c.p_gene = store.gene.p; c.p_aux = store.aux.p; c.n = store.n;
	}
	if (n_reads >= 0xFFFFFFFEull) throw UnsupportedError("more than 2^32-2 reads in one context (shard across GPUs)");
	if (chunks.size() == 1) {
		d_cb = chunks[0].p_cb; d_umi = chunks[0].p_umi; d_gene = chunks[0].p_gene; d_aux = chunks[0].p_aux;
		return;
	}
	cat_cb.alloc(n_reads); cat_umi.alloc(n_reads); cat_gene.alloc(n_reads); cat_aux.alloc(n_reads);
	cat_cb.mark_persistent(); cat_umi.mark_persistent(); cat_gene.mark_persistent(); cat_aux.mark_persistent();
	uint64_t off = 0;
	for (auto &c : chunks) {
		HIP_CHECK(hipMemcpyAsync(cat_cb.p + off, c.p_cb, c.n * 8, hipMemcpyDeviceToDevice, stream));
		HIP_CHECK(hipMemcpyAsync(cat_umi.p + off, c.p_umi, c.n * 8, hipMemcpyDeviceToDevice, stream));
		HIP_CHECK(hipMemcpyAsync(cat_gene.p + off, c.p_gene, c.n * 4, hipMemcpyDeviceToDevice, stream));
		HIP_CHECK(hipMemcpyAsync(cat_aux.p + off, c.p_aux, c.n * 4, hipMemcpyDeviceToDevice, stream));
		off += c.n;
	}
	HIP_CHECK(stream_wait(stream));
	chunks.clear();
	chunks.emplace_back();
	chunks[0].p_cb = cat_cb.p; chunks[0].p_umi = cat_umi.p; chunks[0].p_gene = cat_gene.p; chunks[0].p_aux = cat_aux.p;
	chunks[0].n = n_reads;
	d_cb = cat_cb.p; d_umi = cat_umi.p; d_gene = cat_gene.p; d_aux = cat_aux.p;
}

void dropest_ctx::free_results() {
	HostStage hs(this, "reset");
	invalidate_prefetch();
	initialized = merged = ingested = external_merge_done = false;
	ss_no_reserve = false;
	reseed_rng();    // a second pass over the same reads reproduces the first
	shard.reset();
	n_cells = n_mol = n_cg = n_chr_rows = 0;
	cb_mirror.clear(); real_pristine = false;
	real.clear(); filtered.clear(); filtered_valid = false; merge_pairs.clear(); reassign.clear(); umi_overrides.clear(); n_real_now = 0;
	merge_rank.clear(); reagg_prio = nullptr; extra_excluded.clear(); explicit_sources.clear(); mol_sorted_rows = 0xFFFFFFFFu;
	layout = dropest::KeyLayout{}; umi_clean_bits = 0; umi_sentinel_stripped = false; chr_from_gene = false;   // nothing of the previous pass's key plan survives
	umi_dict_on = false; umi_dict_n = 0; umi_dict_host.clear();
}

// ------------------------------------------------------------------------------------------------
// stage: barcode table + cell ids
// ------------------------------------------------------------------------------------------------
static constexpr u32 GENE_CHR_CAP = 1u << 20;   // genes beyond this id fall back to the general sort layout

void dropest_ctx::build_cb_table() {
	const u32 n = u32(n_reads);
	uint64_t cap = forced_table_capacity ? forced_table_capacity : cfg.cb_table_capacity;
	// A sharded run whose reads arrive in chunks (recv_chunks: piece k of every source's block, ranges of the same arrays): whatever SAMPLES
	// the reads looks at chunk 0, the insert pass takes the chunks one after the other as their events fire -- the later ones are still
	// on the links meanwhile.  (A rebuilt table, forced_table_capacity, finds everything arrived: the first attempt waited for every chunk.)
	const bool chunked = !recv_chunks.empty() && !forced_table_capacity;
	const u32 n_chunks_in = chunked ? u32(recv_chunks.size()) : 1u;
	if (chunked) HIP_CHECK(hipStreamWaitEvent(stream, recv_chunks[0].ev, 0));
	uint64_t n_first = n;   // reads a sampling kernel may look at now
	if (chunked) n_first = recv_chunks[0].rg.total();
	// f(first read, reads) over the ranges a sampling kernel may read
	auto over_sample_ranges = [&](auto &&f) {
		if (!chunked) { f(u32(0), n); return; }
		const CbRanges &rg = recv_chunks[0].rg;
		for (u32 q = 0; q < rg.n; ++q) if (rg.cnt[q]) f(rg.off[q], rg.cnt[q]);
	};
	uint64_t sample_min = uint64_t(1) << 22;
	if (const char *e = getenv("DROPEST_CB_SAMPLE_MIN")) sample_min = uint64_t(std::max(1ll, atoll(e)));   // tests: the sampled sizing and the hot list on small streams
	if (cap == 0 && n_reads >= sample_min && !getenv("DROPEST_CB_NO_SAMPLE")) {
		// size the table from the distinct barcodes of every 64th read: a new barcode shows up at most 64 times as often in
		// the whole stream (too small an estimate is caught below: the table is rebuilt larger when its load passes 0.7)
		const u32 stride = 64, n_s = u32(div_up(u32(std::min<uint64_t>(n_first, n)), stride)) + (chunked ? 64u : 0u);   // (chunked: every range rounds up)
		uint64_t cap_s = 1024; while (cap_s < uint64_t(n_s) * 2) cap_s <<= 1;
		t_slots.ensure(cap_s); scalars.ensure(4 + CB_HOT_LEVELS + 4);
		CbTable ts{t_slots.p, cap_s - 1};
		HIP_CHECK(hipMemsetAsync(t_slots.p, 0, cap_s * sizeof(CbSlot), stream));
		HIP_CHECK(hipMemsetAsync(scalars.p, 0, 4, stream));
		timed("cb_sample", double(n_s) * 8, [&] {
			static const u32 count_every = [] { const char *e = getenv("DROPEST_CB_SAMPLE_COUNT_EVERY"); return e && atoi(e) >= 1 ? u32(atoi(e)) : 4u; }();
			over_sample_ranges([&](u32 off, u32 cnt) {
				hipLaunchKernelGGL(cb_sample_distinct_kernel, dim3(std::min<u32>(div_up(div_up(cnt, stride), 256), 2048u)), dim3(256), 0, stream, d_cb + off, cnt, stride, ts, scalars.p, rpack, count_every);
			});
		});
		// ... and how many of the sampled barcodes reach 4, 16, ... sample hits: the hot list is the largest such set of at
		// most CB_HOT_MAX barcodes (C2: the 5 000 real cells carry 92 % of the reads)
		HIP_CHECK(hipMemsetAsync(scalars.p + 4, 0, 4 * CB_HOT_LEVELS, stream));
		const bool want_hot = !getenv("DROPEST_CB_NO_HOT");
		if (want_hot) timed("cb_hot_count", double(cap_s) * sizeof(CbSlot), [&] { hipLaunchKernelGGL(cb_hot_count_kernel, dim3(1024), dim3(256), 0, stream, ts, scalars.p + 4); });
		u32 head[4 + CB_HOT_LEVELS] = {0};
		fetch(head, scalars.p, sizeof(head));
		const u32 distinct = head[0];
		if (getenv("DROPEST_CB_TRACE")) {
			fprintf(stderr, "[cb] sample of %u reads: %u distinct; barcodes with >= t sample hits:", n_s, distinct);
			for (int l = 0; l < CB_HOT_LEVELS; ++l) fprintf(stderr, " %u:%u", cb_hot_threshold(l), head[4 + l]);
			fprintf(stderr, "\n");
		}
		// (sampled from the first chunk only: scaled to the whole stream)
		const uint64_t est = std::min<uint64_t>(n_reads, uint64_t(double(distinct) * stride * (double(n) / double(std::max<uint64_t>(1, n_first)))));
		cap = 1024; while (cap < est + est / 2) cap <<= 1;   // load <= 0.67 even when the estimate is exact
		n_hot = 0; hot_coverage = 0;
		if (want_hot && cap < (1ull << 31)) {
			int level = -1;
			for (int l = 0; l < CB_HOT_LEVELS; ++l) if (head[4 + l] <= CB_HOT_MAX) { level = l; break; }
			if (level >= 0 && head[4 + level] >= 16) {
				hot_key.ensure(CB_HOT_MAX); hot_slot.ensure(CB_HOT_MAX);
				HIP_CHECK(hipMemsetAsync(scalars.p + 1, 0, 4, stream));
				timed("cb_hot_collect", double(cap_s) * sizeof(CbSlot), [&] {
					hipLaunchKernelGGL(cb_hot_collect_kernel, dim3(1024), dim3(256), 0, stream, ts, cb_hot_threshold(level), hot_key.p, scalars.p + 1);
				});
				n_hot = head[4 + level];
				// how many of the sampled reads the list covers, from below: a barcode counted at threshold l and not at l + 1 has at least
				// cb_hot_threshold(l) sample hits (the thresholds grow by 1.33-1.5x: the bound is within that of the truth)
				double covered = 0;
				for (int l = level; l < CB_HOT_LEVELS; ++l) covered += double(head[4 + l] - (l + 1 < CB_HOT_LEVELS ? head[4 + l + 1] : 0u)) * cb_hot_threshold(l);
				hot_coverage = covered / double(std::max<u32>(1u, n_s));
			}
		}
		if (profiling) stats["count:hot_barcodes"].launches = n_hot;
	} else n_hot = 0;
	if (cap == 0) { cap = 1024; while (cap < n_reads / 2) cap <<= 1; }
	if (cap & (cap - 1)) throw InvalidError("cb_table_capacity must be a power of two");
	slot.ensure(n);
	d_ingest.ensure(1);
	for (int attempt = 0;; ++attempt) {
		if (cap > (1ull << 32)) throw UnsupportedError("barcode table would exceed 2^32 slots");
		t_slots.ensure(cap);
		table.slots = t_slots.p; table.mask = cap - 1;
		HIP_CHECK(hipMemsetAsync(t_slots.p, 0, cap * sizeof(CbSlot), stream));
		IngestStats init{};
		init.umi_clean_min = ~0ull;
		HIP_CHECK(hipMemcpyAsync(d_ingest.p, &init, sizeof(init), hipMemcpyHostToDevice, stream));
		gene_chr.ensure(GENE_CHR_CAP);
		HIP_CHECK(hipMemsetAsync(gene_chr.p, 0xFF, size_t(GENE_CHR_CAP) * 4, stream));
		if (n >= (1u << 20) || lazy_stats) {   // the popular genes' entries are set before the big pass starts (k_cbhash.h)
			const u32 stride = 2048, n_s = div_up(n, stride);
			timed("gene_chr_seed", double(n_s) * 8, [&] {
				over_sample_ranges([&](u32 off, u32 cnt) {
					hipLaunchKernelGGL(gene_chr_seed_kernel, dim3(std::min<u32>(div_up(div_up(cnt, stride), 256), 64u)), dim3(256), 0, stream, d_gene + off, d_aux + off, cnt, stride, gene_chr.p, GENE_CHR_CAP, d_ingest.p, rpack);
				});
			});
		}
		const bool vec = ((uintptr_t(d_cb) | uintptr_t(d_umi) | uintptr_t(d_gene) | uintptr_t(d_aux)) & 15u) == 0;   // adopted arrays may sit anywhere
		static const u32 grid_v = resident_grid(cb_insert_kernel<256, true>, 256, ~0u), grid_s = resident_grid(cb_insert_kernel<256, false>, 256, ~0u);
		const u32 blocks = std::min<u32>(div_up(n, 256 * 4), vec ? grid_v : grid_s);
		if (lazy_stats)   // the plan's statistics: every 256th read (the exact ones come with build_keys)
			timed("ingest_sample_stats", double(div_up(n, 256u)) * 16, [&] {
				const u32 every = std::max<u32>(1u, 256u / n_chunks_in);   // (from the first chunk only: as many reads as every 256th of all)
				over_sample_ranges([&](u32 off, u32 cnt) {
					hipLaunchKernelGGL(ingest_sample_stats_kernel, dim3(std::min<u32>(div_up(div_up(cnt, every), 256), 1024u)), dim3(256), 0, stream, d_umi + off, d_gene + off, d_aux + off, cnt, every, d_ingest.p, rpack);
				});
			});
		if (n_hot && attempt == 0 && cap < (1ull << 31)) {
			// the hot list needs 128 KB of dynamic LDS in one workgroup: a device (or partition mode) that does not grant it takes the
			// plain kernel instead of failing the pass
			const int want = int(size_t(CB_HOT_LDS) * 16);
			bool granted = true;
			for (const void *k : {reinterpret_cast<const void *>(cb_insert_hot_kernel<true, false>), reinterpret_cast<const void *>(cb_insert_hot_kernel<false, false>),
			                      reinterpret_cast<const void *>(cb_insert_hot_kernel<true>), reinterpret_cast<const void *>(cb_insert_hot_kernel<false>)})
				if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, want) != hipSuccess) { (void)hipGetLastError(); granted = false; }
			if (!granted) n_hot = 0;
		}
		if (n_hot && attempt == 0 && cap < (1ull << 31)) {
			// the hot barcodes take their slots first; one workgroup of 1024 threads per CU with the 128 KB LDS table
			hipLaunchKernelGGL(cb_hot_preinsert_kernel, dim3(div_up(n_hot, 256)), dim3(256), 0, stream, hot_key.p, n_hot, table, hot_slot.p, &d_ingest.p->overflow);
			HIP_CHECK(hipGetLastError());
			const size_t lds = size_t(CB_HOT_LDS) * 16;
			int cus = 0;
			HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg.device));
			const u32 hb = std::min<u32>(div_up(n, 1024 * 4), u32(std::max(1, cus)));
			const CbHot hot{hot_key.p, hot_slot.p, n_hot};
			timed("cb_insert", double(n) * (lazy_stats ? 8 + 4 : 8 + 8 + 4 + 4 + 4), [&] {
				auto go = [&](auto kernel) {
					HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
					for (u32 k = 0; k < n_chunks_in; ++k) {   // (one launch, or one per chunk of a chunked exchange as its reads land)
						if (chunked && k) HIP_CHECK(hipStreamWaitEvent(stream, recv_chunks[k].ev, 0));
						hipLaunchKernelGGL(kernel, dim3(hb), dim3(1024), lds, stream, d_cb, d_umi, d_gene, d_aux, chunked ? recv_chunks[k].rg : CbRanges::whole(n), table, hot, slot.p, gene_chr.p, GENE_CHR_CAP, d_ingest.p, rpack);
					}
				};
				if (lazy_stats) { if (vec) go(cb_insert_hot_kernel<true, false>); else go(cb_insert_hot_kernel<false, false>); }
				else if (vec) go(cb_insert_hot_kernel<true>); else go(cb_insert_hot_kernel<false>);
			});
#ifdef DROPEST_CBI_PROBE
			// tuning aid (DROPEST_EXTRA_HIPCC_FLAGS=-DDROPEST_CBI_PROBE at build time, DROPEST_CBI_PROBE=1 at run time): the same pass again
			// over the finished table with parts of it switched off, slots into a scratch array -- what each part costs (profiles/NOTES_r03.md)
			if (lazy_stats && vec && getenv("DROPEST_CBI_PROBE")) {
				keys_a.ensure(n);
				auto again = [&](const char *name, auto kernel) {
					HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
					timed(name, double(n) * 12, [&] { hipLaunchKernelGGL(kernel, dim3(hb), dim3(1024), lds, stream, d_cb, d_umi, d_gene, d_aux, CbRanges::whole(n), table, hot, reinterpret_cast<u32 *>(keys_a.p), gene_chr.p, GENE_CHR_CAP, d_ingest.p); });
				};
				again("cbi:again", cb_insert_hot_kernel<true, false, 0>);
				again("cbi:no_probe", cb_insert_hot_kernel<true, false, 1>);
				again("cbi:no_lds", cb_insert_hot_kernel<true, false, 2>);
				again("cbi:no_atomics", cb_insert_hot_kernel<true, false, 4>);
				again("cbi:no_probe_no_lds", cb_insert_hot_kernel<true, false, 3>);
				again("cbi:no_probe_no_atomics", cb_insert_hot_kernel<true, false, 5>);
			}
#endif
		} else {
		n_hot = 0;   // (a rebuilt table: the slots of the first attempt are gone)
		timed("cb_insert", double(n) * (lazy_stats ? 8 + 4 : 8 + 8 + 4 + 4 + 4), [&] {
			auto go = [&](auto kernel) {
				const bool by_chunk = chunked && attempt == 0;
				for (u32 k = 0; k < (by_chunk ? n_chunks_in : 1u); ++k) {
					if (by_chunk && k) HIP_CHECK(hipStreamWaitEvent(stream, recv_chunks[k].ev, 0));
					hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, stream, d_cb, d_umi, d_gene, d_aux, by_chunk ? recv_chunks[k].rg : CbRanges::whole(n), table, slot.p, gene_chr.p, GENE_CHR_CAP, d_ingest.p, rpack);
				}
			};
			if (lazy_stats) { if (vec) go(cb_insert_kernel<256, true, false>); else go(cb_insert_kernel<256, false, false>); }
			else if (vec) go(cb_insert_kernel<256, true>); else go(cb_insert_kernel<256, false>);
		});
		}
		fetch(&ingest, d_ingest.p, sizeof(ingest));
		if (!ingest.overflow) break;
		if (attempt >= 6) throw DeviceError("barcode table overflow after repeated growth");
		cap <<= 2;
	}
}

void dropest_ctx::assign_cell_ids() {
	const u32 n = u32(n_reads);
	// occupied slots -> (first ordinal, slot) records, sorted by first ordinal: position = first-seen cell id
	const uint64_t cap = table.mask + 1;
	keys_a.ensure(n); keys_b.ensure(n);   // (the key buffers of the main sort: n >= number of barcodes; keys-only sort below)
	vals_a.ensure(1); vals_b.ensure(1);
	scalars.ensure(16);
	HIP_CHECK(hipMemsetAsync(scalars.p, 0, 4, stream));
	timed("cb_compact_slots", double(cap) * 16, [&] {
		hipLaunchKernelGGL(cb_compact_slots_kernel, dim3(u32(std::min<uint64_t>((cap + 4095) / 4096, 4096))), dim3(256), 0, stream, table,
		                   keys_a.p, scalars.p);
	});
	u32 total = 0;
	fetch(&total, scalars.p, 4);
	n_cells = total;
	if (uint64_t(n_cells) * 10 > cap * 7)   // load factor > 0.7: rebuild larger for short probe chains
	{
		forced_table_capacity = cap << 2;   // for THIS pass only (the configuration is not touched: later passes size from their own sample)
		build_cb_table();
		return assign_cell_ids();
	}
	u64 *k = keys_a.p, *k_alt = keys_b.p;
	u32 *v = vals_a.p, *v_alt = vals_b.p;
	const int ord_bits = std::max(1, bit_length(uint64_t(n ? n - 1 : 0)));
	radix_sort(k, v, k_alt, v_alt, n_cells, ((1ull << ord_bits) - 1ull) << 32, 0, "cell_ids:");
	cell_cb.ensure(n_cells); cell_first.ensure(n_cells);
	timed("cb_assign_ids", double(n_cells) * 40, [&] {
		hipLaunchKernelGGL(cb_assign_sorted_kernel, dim3(div_up(n_cells, 256)), dim3(256), 0, stream, k, n_cells, table, cell_cb.p, cell_first.p);
	});
}

// ------------------------------------------------------------------------------------------------
// stage: key layout + keys
// ------------------------------------------------------------------------------------------------
void dropest_ctx::plan_key_layout() {
	KeyLayout L{};
	// UMI field
	const bool any_clean = ingest.umi_clean_max != 0;
	int clean_bits = 0;
	umi_sentinel_stripped = false;
	if (any_clean) {
		const int bl_min = bit_length(ingest.umi_clean_min), bl_max = bit_length(ingest.umi_clean_max);
		if (bl_min == bl_max) {   // one UMI length: drop the sentinel bit
			umi_sentinel_stripped = true;
			clean_bits = bl_max - 1;
			L.umi_strip_mask = clean_bits ? ((1ull << clean_bits) - 1ull) : 0ull;
		} else {
			clean_bits = bl_max;
			L.umi_strip_mask = ~0ull;
		}
	}
	// gene field: ids 0..gene_max, plus the "no gene" code = all ones
	L.gene_bits = bit_length(uint64_t(ingest.gene_max_plus1));
	if ((1ull << L.gene_bits) - 1 < ingest.gene_max_plus1) L.gene_bits++;
	if (L.gene_bits == 0) L.gene_bits = 1;
	L.gene_none = (1ull << L.gene_bits) - 1ull;
	L.cell_bits = std::max(1, bit_length(uint64_t(n_cells ? n_cells - 1 : 0)));
	// The UMI's own code as the field, unless the key cannot hold it: then its rank in a dictionary of the stream's UMIs
	// (k_umidict.h; StringIndexer.cpp:10-18 has no width limit).  Once a pass has the dictionary it keeps it.
	// (A sharded / split run builds ONE dictionary over all shards before the keys are planned -- csrc/shard_run.h: global_umi_dictionary, round 6 --
	// and arrives here with it.)
	if (!umi_dict_on && n_reads && !hooks && !rpack.on() && umi_dict_wanted(L.gene_bits, L.cell_bits)) build_umi_dict();
	if (umi_dict_on) {
		umi_sentinel_stripped = false;
		clean_bits = std::max(1, bit_length(uint64_t(umi_dict_n) + 255u));   // (room for UMIs the public mutators bring in: add_umi / merge_umis)
		L.umi_strip_mask = ~0ull;   // ranks pass as they are
	}
	umi_clean_bits = clean_bits;
	L.umi_escape_base = 1ull << clean_bits;
	L.umi_bits = clean_bits;
	if (ingest.umi_escape_max_plus1) L.umi_bits = bit_length(L.umi_escape_base + ingest.umi_escape_max_plus1 - 1);
	wanted_bits[0] = u32(L.cell_bits); wanted_bits[1] = u32(L.gene_bits); wanted_bits[2] = u32(L.umi_bits);
	if (L.gene_bits + L.umi_bits >= 64)
		throw UnsupportedError("gene + UMI fields of the sort key need " + std::to_string(L.gene_bits + L.umi_bits) + " bits and leave no room for the cells; "
		                       "the pass should have keyed the UMIs by their rank in a dictionary (k_umidict.h; sharded runs: shard_run.h global_umi_dictionary)");
	if (L.umi_bits + L.gene_bits + L.cell_bits > 64)
		throw UnsupportedError("sort key needs " + std::to_string(L.umi_bits + L.gene_bits + L.cell_bits) +
		                       " bits (cell " + std::to_string(L.cell_bits) + " + gene " + std::to_string(L.gene_bits) +
		                       " + UMI " + std::to_string(L.umi_bits) + "); one context sorts 64-bit keys -- the cell field shrinks when the stream is split by barcode: dropest_ctx_split");
	// sort-record layout (k_misc.h): derive the chromosome from the gene when that is a function and the chromosome
	// of a gene-less read fits the UMI field; then the mark rides in the key if 3 bits are free, else as one byte
	chr_from_gene = !ingest.gene_chr_conflict && (L.umi_bits >= 16 || (u64(ingest.chr_max_plus1) <= (1ull << L.umi_bits)));
	if (getenv("DROPEST_FORCE_GENERAL_LAYOUT")) chr_from_gene = false;   // tests exercise the general path on any data
	if (chr_from_gene) {
		const bool fits = L.umi_bits + L.gene_bits + L.cell_bits + 3 <= 64 && !getenv("DROPEST_FORCE_BYTE_VALUES");
		L.mark_shift = fits ? 3 : 0;
		L.val_bytes = fits ? 0 : 1;
	} else {
		L.mark_shift = 0; L.val_bytes = 4;
	}
	layout = L;
}

// Does this pass key its UMIs by their rank in a dictionary?  From the ingest statistics alone (a sharded run asks after its shards have
// agreed on them, so all of them answer alike); cell_bits only matters for mode 1.
bool dropest_ctx::umi_dict_wanted(int gene_bits, int cell_bits) const {
	int clean_bits = 0;
	if (ingest.umi_clean_max != 0) {
		const int bl_min = bit_length(ingest.umi_clean_min), bl_max = bit_length(ingest.umi_clean_max);
		clean_bits = bl_min == bl_max ? bl_max - 1 : bl_max;
	}
	int plain_bits = clean_bits;
	if (ingest.umi_escape_max_plus1) plain_bits = bit_length((1ull << clean_bits) + ingest.umi_escape_max_plus1 - 1);
	const int mode = getenv("DROPEST_UMI_DICT") ? atoi(getenv("DROPEST_UMI_DICT")) : umi_dict_mode;
	return mode >= 2 || gene_bits + plain_bits >= 64 || (mode == 1 && plain_bits + gene_bits + cell_bits > 64);
}

// The distinct clean UMIs of the gene-bearing reads, ascending, and every read's rank among them (k_umidict.h).
// across: a sharded run's exchange -- given this shard's dictionary (device, ascending) it leaves the dictionary of ALL shards in its place
// (csrc/shard_run.h); the ranks are then the same on every shard and still ascend with the codes.
void dropest_ctx::build_umi_dict(const std::function<void(dropest::DevBuf<u64> &, u32 &)> *across) {
	HostStage hs(this, "umi_dict");
	need_columns();
	const u32 n = u32(n_reads);
	u32 total = 0;
	umi_dict.ensure(1);
	if (n) {   // (a shard without reads still takes part in the exchange below)
		keys_a.ensure(n); keys_b.ensure(n); vals_a.ensure(1); vals_b.ensure(1);
		u64 *k = keys_a.p, *k_alt = keys_b.p;
		u32 *v = vals_a.p, *v_alt = vals_b.p;
		timed("umi_dict:fill", double(n) * 20, [&] {
			hipLaunchKernelGGL(umi_dict_fill_kernel, dim3(std::min<u32>(div_up(n, 256), 8192u)), dim3(256), 0, stream, d_umi, d_gene, n, k);
		});
		radix_sort(k, v, k_alt, v_alt, n, ~0ull, 0, "umi_dict:");
		const u32 nb = div_up(n, u32(UD_T * UD_PER));
		DevBuf<u32> cnt, base;
		cnt.alloc(nb); base.alloc(nb);
		scalars.ensure(16);
		timed("umi_dict:unique", double(n) * 16, [&] {
			hipLaunchKernelGGL(umi_dict_count_kernel, dim3(nb), dim3(UD_T), 0, stream, k, n, cnt.p);
			scan_counts(cnt.p, base.p, nb, scalars.p);
		});
		fetch(&total, scalars.p, 4);
		umi_dict.ensure(std::max<u32>(total, 1u));
		if (total) hipLaunchKernelGGL(umi_dict_write_kernel, dim3(nb), dim3(UD_T), 0, stream, k, n, base.p, umi_dict.p);
		HIP_CHECK(hipGetLastError());
	}
	if (across) { HIP_CHECK(stream_wait(stream)); (*across)(umi_dict, total); }
	umi_dict_n = total;
	umi_ranked.ensure(std::max<u32>(n, 1u));
	timed("umi_dict:rank", double(n) * 20, [&] {
		if (n) hipLaunchKernelGGL(umi_dict_rank_kernel, dim3(std::min<u32>(div_up(n, 256), 8192u)), dim3(256), 0, stream, d_umi, d_gene, n, umi_dict.p, total, umi_ranked.p);
	});
	HIP_CHECK(hipGetLastError());
	umi_dict_host.resize(total);
	if (total) fetch(umi_dict_host.data(), umi_dict.p, size_t(total) * 8);
	else HIP_CHECK(stream_wait(stream));
	umi_dict_on = true;
	if (profiling) stats["count:umi_dictionary"].launches += 1;
}

void dropest_ctx::sort_unique_u64(dropest::DevBuf<u64> &d, u32 &n) {
	if (n == 0) return;
	keys_a.ensure(n); keys_b.ensure(n); vals_a.ensure(1); vals_b.ensure(1);
	HIP_CHECK(hipMemcpyAsync(keys_a.p, d.p, size_t(n) * 8, hipMemcpyDeviceToDevice, stream));
	u64 *k = keys_a.p, *k_alt = keys_b.p;
	u32 *v = vals_a.p, *v_alt = vals_b.p;
	radix_sort(k, v, k_alt, v_alt, n, ~0ull, 0, "umi_dict:all:");
	const u32 nb = div_up(n, u32(UD_T * UD_PER));
	DevBuf<u32> cnt, base;
	cnt.alloc(nb); base.alloc(nb);
	scalars.ensure(16);
	hipLaunchKernelGGL(umi_dict_count_kernel, dim3(nb), dim3(UD_T), 0, stream, k, n, cnt.p);
	scan_counts(cnt.p, base.p, nb, scalars.p);
	u32 total = 0;
	fetch(&total, scalars.p, 4);
	d.ensure(std::max<u32>(total, 1u));
	if (total) hipLaunchKernelGGL(umi_dict_write_kernel, dim3(nb), dim3(UD_T), 0, stream, k, n, base.p, d.p);
	HIP_CHECK(hipGetLastError());
	HIP_CHECK(stream_wait(stream));
	n = total;
}

dropest_ctx::u64 dropest_ctx::unmap_umi(u64 ucode) const {
	if (ucode >= layout.umi_escape_base && ingest.umi_escape_max_plus1) return ESCAPE_BIT | (ucode - layout.umi_escape_base);
	if (umi_dict_on) {
		if (ucode >= umi_dict_host.size()) throw InvalidError("internal: UMI rank beyond the dictionary");
		return umi_dict_host[size_t(ucode)];
	}
	if (umi_sentinel_stripped) return (1ull << umi_clean_bits) | ucode;
	return ucode;
}

bool dropest_ctx::map_umi(u64 api_code, u64 &field) const {
	const u64 umask = layout.umi_bits ? ((1ull << layout.umi_bits) - 1ull) : 0ull;
	if (api_code & ESCAPE_BIT) {
		const u64 id = api_code & ~ESCAPE_BIT;
		if (id >= ingest.umi_escape_max_plus1) return false;
		field = layout.umi_escape_base + id;
		return true;
	}
	if (umi_dict_on) {   // the first umi_dict_n entries ascend (the device built them); what the mutators appended follows in their order
		const auto sorted_end = umi_dict_host.begin() + umi_dict_n;
		auto it = std::lower_bound(umi_dict_host.begin(), sorted_end, api_code);
		if (it == sorted_end || *it != api_code) it = std::find(sorted_end, umi_dict_host.end(), api_code);
		if (it == umi_dict_host.end()) return false;   // (a UMI no gene-bearing read of this pass carries)
		field = u64(it - umi_dict_host.begin());
		return true;
	}
	if (umi_sentinel_stripped && bit_length(api_code) - 1 != umi_clean_bits) return false;
	field = api_code & layout.umi_strip_mask;
	return !(field > umask || (ingest.umi_escape_max_plus1 && field >= layout.umi_escape_base));
}

// A clean UMI a public mutator names that no read carried (CellsDataContainer::add_umi_to_cell / Cell::merge_umis take any string;
// StringIndexer::add hands it the next index): the dictionary takes it at its end while the UMI field has room.
bool dropest_ctx::map_umi_or_add(u64 api_code, u64 &field) {
	if (map_umi(api_code, field)) return true;
	if (!umi_dict_on || (api_code & ESCAPE_BIT) || u64(umi_dict_host.size()) >= layout.umi_escape_base) return false;
	field = u64(umi_dict_host.size());
	umi_dict_host.push_back(api_code);
	return true;
}

static int sort_mode_override() {   // read at every pass: tests switch it inside one process
	const char *e = getenv("DROPEST_SORT");
	if (!e) return 0;
	return !strcmp(e, "lsd") ? 1 : (!strcmp(e, "splitter") ? 2 : 0);
}

// The splitter sort's fan-out for n records and, when the partitions place their records by reservation (k_ssort.h: no histogram
// passes), the capacities of the bucket regions.  build_keys sizes the key buffers from it: the regions of the second level take 1.75 n
// records of address space.  DROPEST_SS_NO_RESERVE=1: always the counting partitions.
struct SsPlan { bool applicable = false, reserve = false; int tb = 0, fb1 = 0, fb2 = 0; uint64_t cap1 = 0, cap2 = 0, span = 0, os = 64; };
static SsPlan ss_plan(uint64_t n_reads, bool chr_from_gene, int val_bytes) {
	SsPlan P;
	const int mode = sort_mode_override();
	const char *e_min = getenv("DROPEST_SSORT_MIN");
	const uint64_t min_reads = e_min ? uint64_t(atoll(e_min)) : (uint64_t(1) << 22);
	if (mode == 1 || !chr_from_gene || val_bytes > 1) return P;
	if (mode != 2 && n_reads < min_reads) return P;
	if (n_reads < 512 || n_reads >= 0xFFFFFFFEull) return P;
	int tb = 8;
	while (tb < 20 && (n_reads >> tb) > 1600) ++tb;
	if (const char *e = getenv("DROPEST_SSORT_TB")) tb = std::min(20, std::max(8, atoi(e)));
	if ((n_reads >> tb) > 4096) return P;
	P.applicable = true; P.tb = tb; P.fb1 = tb / 2; P.fb2 = tb - P.fb1;
	const uint64_t F1 = 1ull << P.fb1, F2 = 1ull << tb, mean1 = (n_reads + F1 - 1) / F1, mean2 = (n_reads + F2 - 1) / F2;
	P.cap1 = (mean1 + mean1 / 16 + 2 * SS_TILE + 15) & ~15ull;
	P.cap2 = (mean2 + mean2 * 3 / 4 + 64 + 15) & ~15ull;
	// Samples per fine bucket.  A bucket spans `os` sample gaps: its size scatters like a Gamma(os) variable, 12.5 % at 64, 17.7 % at 32.
	// Where the mean is at most half of what the finishing launch holds (2 048 records) -- 1e9 reads: 2^20 buckets of 954 -- half the
	// sample does: the sample sort was 5.4 of 120 ms of kernels there, 2.6 now.  The regions then hold 2.5 x the mean
	// (a Gamma(32) bucket beyond that: 1e-11; at 2.0 x it happened in a million buckets).
	// (Only for large streams: below ~2.5e8 reads the sample sort is a matter of launch latencies, and the wider regions cost the second
	// scatter what the smaller sample saves: 6e7 reads 4.77 against 4.80 ms of kernels.)
	if (mean2 <= 1024 && n_reads >= (1ull << 28) && !getenv("DROPEST_SSORT_OS")) {
		P.os = 32;
		P.cap2 = (mean2 * 5 / 2 + 64 + 15) & ~15ull;
	}
	if (const char *e = getenv("DROPEST_SS_CAP2_PERCENT")) P.cap2 = (mean2 * uint64_t(std::max(100, atoi(e))) / 100 + 15) & ~15ull;   // tests: regions that overflow
	P.span = std::max(F1 * P.cap1, F2 * P.cap2);
	P.reserve = !getenv("DROPEST_SS_NO_RESERVE") && P.span < 0xFFFFFFF0ull;
	if (!P.reserve) P.span = n_reads;
	return P;
}

void dropest_ctx::ss_splitters_from_sample(u32 n_sample, u32 os, u32 Ff, u32 F2, u64 varying_of_sample) {
	u64 *k = ss_sample_a.p, *k_alt = ss_sample_b.p;
	u32 *v = nullptr, *v_alt = nullptr;
	radix_sort(k, v, k_alt, v_alt, n_sample, varying_of_sample, 0, "ss_sample:");
	hipLaunchKernelGGL(ss_pick_splitters_kernel, dim3(div_up(F2, 256)), dim3(256), 0, stream, k, os, Ff, F2, ss_fine.p, ss_coarse.p);
	HIP_CHECK(hipGetLastError());
}

// The keys built and partitioned into the coarse regions in ONE pass (k_keyscatter.h): the sample of the splitter sort comes from the reads
// before the key pass, so the coarse splitters exist when the keys are made.
bool dropest_ctx::build_keys_fused(bool with_stats) {
	const bool off = getenv("DROPEST_NO_FUSED_KEYS") != nullptr;   // (read at every pass: tests run both ways in one process)
	if (off || ss_no_reserve || !chr_from_gene || layout.val_bytes > 1) return false;
	// One workgroup per CU hides less latency than the two kernels' many: the fused pass wins where most reads find their cell id in LDS (C2:
	// 0.86 + 0.55 -> 1.17 ms) and merely ties where most gather it from the table (C3 at 1e9 reads, 1 365 of 50 000 cells listed: 11.7 +
	// 12.9 -> 25.0 ms).  Taken when the hot list covers a good part of the sampled reads.
	const char *e_cov = getenv("DROPEST_FUSED_KEYS_MIN_COVERAGE");
	const double min_cov = e_cov ? atof(e_cov) : 0.4;
	if (hot_coverage < min_cov) return false;
	const SsPlan plan = ss_plan(n_reads, chr_from_gene, layout.val_bytes);
	if (!plan.applicable || !plan.reserve) return false;
	const u32 n = u32(n_reads);
	const bool vec = ((uintptr_t(umi_key_column()) | uintptr_t(d_gene) | uintptr_t(d_aux)) & 15u) == 0;
	if (!vec) return false;
	const int fb1 = plan.fb1, fb2 = plan.fb2, ms = layout.mark_shift, VB = layout.val_bytes;
	const u32 F1 = 1u << fb1, Ff = 1u << fb2, F2 = F1 * Ff;
	if (F1 > KS_MAXF) return false;
	uint64_t os_max = plan.os;
	if (const char *e = getenv("DROPEST_SSORT_OS")) os_max = uint64_t(std::max(1, atoi(e)));
	const u32 os = u32(std::max<uint64_t>(1, std::min<uint64_t>(os_max, uint64_t(n) / (uint64_t(F2) * 2))));
	const u32 n_sample = F2 * os;
	const u32 lds_genes = with_stats && !getenv("DROPEST_NO_LDS_GENE_TABLE") ? std::min<u32>((ingest.gene_max_plus1 + 3u) & ~3u, BK_LDS_GENES_MAX) : 0u;
	const size_t lds = ks_dynamic_lds(lds_genes);
	const size_t span = size_t(plan.span);
	const size_t val_words = (span * size_t(VB) + 3) / 4 + 1;
	keys_a.ensure(span); keys_b.ensure(span); vals_a.ensure(val_words); vals_b.ensure(val_words);
	const CbHot hot{hot_key.p, hot_slot.p, n_hot};
	// sample of the reads -> sorted -> fine / coarse splitters
	ss_sample_a.ensure(n_sample); ss_sample_b.ensure(n_sample); ss_fine.ensure(F2); ss_coarse.ensure(F1);
	timed("ss_sample", double(n_sample) * (20.0 / 16 * 8 + 8), [&] {
		if (rpack.on()) hipLaunchKernelGGL(ss_sample_reads_kernel<true>, dim3(div_up(n_sample, 256)), dim3(256), 0, stream, umi_key_column(), d_gene, d_aux, slot.p, n, table, layout, hot, rpack, n_sample, ss_sample_a.p);
		else hipLaunchKernelGGL(ss_sample_reads_kernel<false>, dim3(div_up(n_sample, 256)), dim3(256), 0, stream, umi_key_column(), d_gene, d_aux, slot.p, n, table, layout, hot, rpack, n_sample, ss_sample_a.p);
	});
	const int key_bits = layout.cell_bits + layout.gene_bits + layout.umi_bits;
	ss_splitters_from_sample(n_sample, os, Ff, F2, key_bits >= 64 ? ~0ull : ((1ull << key_bits) - 1ull));   // (every bit of the key fields may vary: the sample's own OR / AND would cost a round trip)
	// regions, cursors, flags
	const u32 cap1 = u32(plan.cap1), CS1 = 32;
	ss_cursors.ensure(size_t(F1) * CS1 + F2); ks_flag.ensure(1);
	HIP_CHECK(hipMemsetAsync(ss_cursors.p, 0, (size_t(F1) * CS1 + F2) * 4, stream));
	HIP_CHECK(hipMemsetAsync(ks_flag.p, 0, 4, stream));
	d_counters.ensure(1);
	GlobalCounters init{};
	init.key_and = ~0ull;
	HIP_CHECK(hipMemcpyAsync(d_counters.p, &init, sizeof(init), hipMemcpyHostToDevice, stream));
	if (with_stats) {
		IngestStats zero{};
		zero.umi_clean_min = ~0ull;
		zero.gene_chr_conflict = ingest.gene_chr_conflict;
		HIP_CHECK(hipMemcpyAsync(d_ingest.p, &zero, sizeof(zero), hipMemcpyHostToDevice, stream));
	}
	int cus = 0, dev = 0;
	HIP_CHECK(hipGetDevice(&dev));
	HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
	const SsReserve r1{ss_cursors.p, CS1, cap1, 0u, ks_flag.p, 0u};
	bool launched = true;
	timed("build_keys+L1", double(n) * (8 + 4 + 4 + 4 + 4 + 8 + VB), [&] {
		auto go = [&](auto kernel) {
			if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess) { (void)hipGetLastError(); launched = false; return; }
			const u32 blocks = std::max<u32>(1u, std::min<u32>(div_up(n, u32(KS_TILE)), u32(std::max(1, cus))));
			hipLaunchKernelGGL(kernel, dim3(blocks), dim3(KS_T), lds, stream, umi_key_column(), d_gene, d_aux, slot.p, n, table, layout, d_counters.p, hot, gene_chr.p, GENE_CHR_CAP, d_ingest.p,
			                   lds_genes, rpack, keys_b.p, reinterpret_cast<uint8_t *>(vals_b.p), ms, fb1, ss_coarse.p, r1);
		};
		auto pick = [&](auto vb) {
			constexpr int V = decltype(vb)::value;
			const int mode = with_stats ? (lds_genes ? 2 : 1) : 0;
			if (rpack.on()) { if (mode == 2) go(build_keys_scatter_kernel<V, 2, true>); else if (mode == 1) go(build_keys_scatter_kernel<V, 1, true>); else go(build_keys_scatter_kernel<V, 0, true>); }
			else { if (mode == 2) go(build_keys_scatter_kernel<V, 2, false>); else if (mode == 1) go(build_keys_scatter_kernel<V, 1, false>); else go(build_keys_scatter_kernel<V, 0, false>); }
		};
		if (VB == 0) pick(std::integral_constant<int, 0>{}); else pick(std::integral_constant<int, 1>{});
	});
	if (!launched) return false;   // (a device that does not grant the LDS: the two kernels)
	fetch(&counters, d_counters.p, sizeof(counters));
	if (with_stats) {
		IngestStats exact{};
		fetch(&exact, d_ingest.p, sizeof(exact));
		exact.cb_escape_count = ingest.cb_escape_count;
		exact.overflow = 0;
		ingest = exact;
	}
	keys_in_l1 = true;
	return true;
}

void dropest_ctx::build_keys(bool with_stats, bool allow_fused) {
	keys_in_l1 = false;
	if (allow_fused && build_keys_fused(with_stats)) return;
	const u32 n = u32(n_reads);
	// value buffers hold val_bytes per record (0, 1 or 4): nothing at all for the keys-only layout
	// (the splitter sort's partitions by reservation write into bucket regions with gaps: up to 1.75 n records of address space)
	const size_t span = std::max<size_t>(n, ss_no_reserve ? 0 : size_t(ss_plan(n_reads, chr_from_gene, layout.val_bytes).span));
	const size_t val_words = (span * size_t(layout.val_bytes) + 3) / 4 + 1;
	keys_a.ensure(span); keys_b.ensure(span); vals_a.ensure(val_words); vals_b.ensure(val_words);
	d_counters.ensure(1);
	GlobalCounters init{};
	init.key_and = ~0ull;
	HIP_CHECK(hipMemcpyAsync(d_counters.p, &init, sizeof(init), hipMemcpyHostToDevice, stream));
	if (with_stats) {   // the exact ingest statistics ride along (the layout in use was planned from a sample); the gene ->
		IngestStats zero{};   // chromosome table holds what the seed pass put there: same protocol, nothing to clear
		zero.umi_clean_min = ~0ull;
		zero.gene_chr_conflict = ingest.gene_chr_conflict;
		HIP_CHECK(hipMemcpyAsync(d_ingest.p, &zero, sizeof(zero), hipMemcpyHostToDevice, stream));
	}
	const bool vec = ((uintptr_t(umi_key_column()) | uintptr_t(d_gene) | uintptr_t(d_aux)) & 15u) == 0;
	timed("build_keys", double(n) * (8 + 4 + 4 + 4 + 4 + 8 + layout.val_bytes), [&] {
		void *v = vals_a.p;
		const CbHot hot{hot_key.p, hot_slot.p, n_hot};   // n_hot: slot[] holds CB_HOT_FLAG | hot index for the reads of the hot barcodes
		// 12-16 waves per CU run this pass fastest (scripts/probe/bk_probe.hip: 0.74 ms with the 32 that fit, 0.68 with 16, 0.66 with 8
		// at the C2 shape): at most four workgroups per CU
		int cus = 0, dev = 0;
		HIP_CHECK(hipGetDevice(&dev));
		HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
		// the gene -> chromosome check out of LDS (k_misc.h: GCL) when the sample's genes fit the byte table; later genes take the exact path
		const u32 lds_genes = with_stats && !getenv("DROPEST_NO_LDS_GENE_TABLE") ? std::min<u32>((ingest.gene_max_plus1 + 3u) & ~3u, BK_LDS_GENES_MAX) : 0u;
		auto go = [&](auto kernel, u32 lds = 0u) {
			int per_cu = 0;
			HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds));
			const u32 blocks = std::max<u32>(1u, std::min<u32>(div_up(n, 256 * 4), u32(std::max(1, cus) * std::max(1, std::min(per_cu, 4)))));
			hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds, stream, umi_key_column(), d_gene, d_aux, slot.p, n, table, layout, keys_a.p, v, d_counters.p, hot,
			                   gene_chr.p, GENE_CHR_CAP, d_ingest.p, lds, rpack);
		};
		if (rpack.on() && !vec) throw InvalidError("internal: packed exchange records that are not 16-byte aligned");
		auto pick = [&](auto vb) {
			constexpr int VB = decltype(vb)::value;
			if (rpack.on()) {   // a sharded run's records as they arrived (PK)
				if (with_stats && lds_genes) { if (n_hot) go(build_keys_kernel<256, VB, true, true, true, true, true>, lds_genes); else go(build_keys_kernel<256, VB, true, false, true, true, true>, lds_genes); }
				else if (with_stats) { if (n_hot) go(build_keys_kernel<256, VB, true, true, true, false, true>); else go(build_keys_kernel<256, VB, true, false, true, false, true>); }
				else { if (n_hot) go(build_keys_kernel<256, VB, true, true, false, false, true>); else go(build_keys_kernel<256, VB, true, false, false, false, true>); }
			} else if (with_stats && lds_genes && vec) {
				if (n_hot) go(build_keys_kernel<256, VB, true, true, true, true>, lds_genes); else go(build_keys_kernel<256, VB, true, false, true, true>, lds_genes);
			} else if (with_stats) {
				if (n_hot) { if (vec) go(build_keys_kernel<256, VB, true, true, true>); else go(build_keys_kernel<256, VB, false, true, true>); }
				else { if (vec) go(build_keys_kernel<256, VB, true, false, true>); else go(build_keys_kernel<256, VB, false, false, true>); }
			} else {
				if (n_hot) { if (vec) go(build_keys_kernel<256, VB, true, true>); else go(build_keys_kernel<256, VB, false, true>); }
				else { if (vec) go(build_keys_kernel<256, VB, true>); else go(build_keys_kernel<256, VB, false>); }
			}
		};
		if (layout.val_bytes == 0) pick(std::integral_constant<int, 0>{});
		else if (layout.val_bytes == 1) pick(std::integral_constant<int, 1>{});
		else pick(std::integral_constant<int, 4>{});
	});
	fetch(&counters, d_counters.p, sizeof(counters));
	if (with_stats) {
		IngestStats exact{};
		fetch(&exact, d_ingest.p, sizeof(exact));
		exact.cb_escape_count = ingest.cb_escape_count;   // counted by cb_insert, which read the barcodes
		exact.overflow = 0;
		ingest = exact;
	}
}

// ------------------------------------------------------------------------------------------------
// stage: radix sort
// ------------------------------------------------------------------------------------------------
// Tile shape per value width, by measurement: the narrow records (keys only, key + 1 byte) run best as
// 512 threads x 8 records without register prefetch (96-112 VGPRs, 45 KB LDS: 3 blocks per CU hide the latency instead);
// the 12-byte record keeps 512 x 16 with the next tile prefetched into registers (1 block per CU).
static constexpr int RS_T = 512, RS_I = 16, RS_TILE_REC = RS_T * RS_I;
// Tile shapes of the keys-only pass (THREADS x ITEMS must equal RS_TILE_REC or divide it: the histogram pass counts per
// block range, not per tile).  Chosen by measurement; DROPEST_RS_SHAPE=<n> selects another for tuning.
template <int T, int I, bool PF, int VB, int RB = 8>
static void rs_launch_shape(dim3 grid, hipStream_t st, const u64 *k, const void *v, u64 *ok, void *ov, u32 n, int shift, u32 tpb,
                            const u32 *hist, const u32 *base) {
	static_assert(RS_TILE_REC % (T * I) == 0, "shape must tile the histogram ranges");
	hipLaunchKernelGGL((rs_scatter_kernel_t<T, I, PF, VB, RB>), grid, dim3(T), 0, st, k, v, ok, ov, n, shift, tpb * u32(RS_TILE_REC / (T * I)), hist, base);
}
static int rs_shape_override() {
	static const int shape = [] { const char *e = getenv("DROPEST_RS_SHAPE"); return e ? atoi(e) : -1; }();
	return shape;
}
template <int VB>
static void rs_launch_vb(int shape, dim3 grid, hipStream_t st, const u64 *k, const void *v, u64 *ok, void *ov, u32 n, int shift, u32 tpb,
                         const u32 *hist, const u32 *base) {
	switch (shape) {
		case 1: return rs_launch_shape<512, 8, true, VB>(grid, st, k, v, ok, ov, n, shift, tpb, hist, base);
		case 2: return rs_launch_shape<512, 8, false, VB>(grid, st, k, v, ok, ov, n, shift, tpb, hist, base);
		case 3: return rs_launch_shape<256, 8, false, VB>(grid, st, k, v, ok, ov, n, shift, tpb, hist, base);
		case 4: return rs_launch_shape<512, 4, false, VB>(grid, st, k, v, ok, ov, n, shift, tpb, hist, base);
		case 5: return rs_launch_shape<1024, 4, false, VB>(grid, st, k, v, ok, ov, n, shift, tpb, hist, base);
		case 6: return rs_launch_shape<256, 16, false, VB>(grid, st, k, v, ok, ov, n, shift, tpb, hist, base);
		case 7: return rs_launch_shape<1024, 8, false, VB>(grid, st, k, v, ok, ov, n, shift, tpb, hist, base);
		default: return rs_launch_shape<512, 16, true, VB>(grid, st, k, v, ok, ov, n, shift, tpb, hist, base);
	}
}
static void rs_launch(int val_bytes, int bits, dim3 grid, hipStream_t st, const u64 *k, const void *v, u64 *ok, void *ov, u32 n, int shift, u32 tpb,
                      const u32 *hist, const u32 *base) {
	if (bits == 9) {   // the widened pass: the default shapes only (512 threads = one per digit)
		if (val_bytes == 0) return rs_launch_shape<512, 8, false, 0, 9>(grid, st, k, v, ok, ov, n, shift, tpb, hist, base);
		if (val_bytes == 1) return rs_launch_shape<512, 8, false, 1, 9>(grid, st, k, v, ok, ov, n, shift, tpb, hist, base);
		return rs_launch_shape<512, 16, true, 4, 9>(grid, st, k, v, ok, ov, n, shift, tpb, hist, base);
	}
	const int o = rs_shape_override();
	if (val_bytes == 0) rs_launch_vb<0>(o >= 0 ? o : 2, grid, st, k, v, ok, ov, n, shift, tpb, hist, base);
	else if (val_bytes == 1) rs_launch_vb<1>(o >= 0 ? o : 2, grid, st, k, v, ok, ov, n, shift, tpb, hist, base);
	else rs_launch_vb<4>(o >= 0 ? o : 0, grid, st, k, v, ok, ov, n, shift, tpb, hist, base);
}

// Digit windows of an LSD sort over the bits set in `varying_mask` (the caller masks out bits that need no ordering,
// e.g. a mark folded under the key; bits equal in all keys are not set either).  8-bit windows from the lowest varying
// bit, windows without a varying bit skipped: ceil(width / 8) passes.  When ceil(width / 9) is smaller (C2: 22 + 15 +
// 20 = 57 bits -> 7 passes instead of 8; C4: 53 bits -> 6 instead of 7) the TOP windows take 9 bits, as many as needed,
// and a whole pass (one histogram + one scatter over all records) disappears.  A 9-bit scatter launch costs ~10 % more
// than an 8-bit one (runs per tile half as long; measured 0.59 vs 0.55 ms on the C4 shape, 0.53 vs 0.47 ms for the one
// widened pass of C2, whose digit -- the high bits of the cell id -- is so skewed that most of the 512 buckets are
// empty), a pass less saves 0.65-0.75 ms: C2 20.05 -> 19.40 ms, C4 shape 33.8 -> 31.8 ms, C3 at 1e8 reads 36.7 -> 36.2.
struct RadixPass { int shift, bits; };
static constexpr u32 RS_RADIX_MAX = 512;
static std::vector<RadixPass> plan_radix_passes(u64 varying_mask) {
	std::vector<RadixPass> plain, wide;
	if (!varying_mask) return plain;
	const int lo = __builtin_ctzll(varying_mask), hi = 64 - __builtin_clzll(varying_mask), width = hi - lo;
	for (int shift = lo; shift < 64; shift += 8)
		if ((varying_mask >> shift) & 0xFFull) plain.push_back({shift, 8});
	static const bool allow_wide = getenv("DROPEST_RS_NO_WIDE") == nullptr;
	const int passes = (width + 8) / 9, n9 = width - 8 * passes;   // passes = ceil(width / 9)
	static const int max_wide = [] { const char *e = getenv("DROPEST_RS_WIDE_MAX"); return e ? atoi(e) : 64; }();
	if (!allow_wide || n9 < 1 || n9 > max_wide || size_t(passes) >= plain.size()) return plain;
	int shift = lo;
	for (int i = 0; i < passes; ++i) { const int bits = i >= passes - n9 ? 9 : 8; wide.push_back({shift, bits}); shift += bits; }
	return wide;
}

void dropest_ctx::scan_counts(const u32 *in, u32 *out, u32 n, u32 *total_out) {
	if (n <= 16384 || n > 1024u * 1024u) {
		hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, stream, in, out, n, total_out);
		return;
	}
	const u32 n_chunks = div_up(n, 1024);
	scan_chunk.ensure(1024);
	hipLaunchKernelGGL(ss_chunk_sums_kernel<1024>, dim3(n_chunks), dim3(1024), 0, stream, in, n, scan_chunk.p);
	hipLaunchKernelGGL(ss_prefix_kernel<1024>, dim3(n_chunks), dim3(1024), 0, stream, in, n, scan_chunk.p, n_chunks, out, total_out);
}

void dropest_ctx::radix_sort(u64 *&keys, u32 *&vals, u64 *&keys_alt, u32 *&vals_alt, u32 n, u64 varying_mask, int val_bytes,
                             const char *stat_prefix) {
	if (n == 0) return;
	const u32 n_tiles = div_up(n, RS_TILE_REC);
	u32 nblocks = std::min<u32>(n_tiles, 1024);   // measured flat between 256 and 2048 blocks
	const u32 tpb = div_up(n_tiles, nblocks);
	nblocks = div_up(n_tiles, tpb);
	rs_hist.ensure(size_t(RS_RADIX_MAX) * nblocks); rs_row_total.ensure(RS_RADIX_MAX); rs_digit_base.ensure(RS_RADIX_MAX);
	const std::string scatter_s = std::string(stat_prefix ? stat_prefix : "") +
	                              (val_bytes == 4 ? "rs_scatter" : (val_bytes == 1 ? "rs_scatter:key+1B" : "rs_scatter:keys"));
	const std::string hist_s = std::string(stat_prefix ? stat_prefix : "") + "rs_hist", scan_s = std::string(stat_prefix ? stat_prefix : "") + "rs_scan";
	const char *scatter_name = scatter_s.c_str();
	for (const RadixPass &ps : plan_radix_passes(varying_mask)) {
		const int shift = ps.shift;
		const u32 radix = 1u << ps.bits;
		timed(hist_s.c_str(), double(n) * 8, [&] {
			if (ps.bits == 9) hipLaunchKernelGGL(rs_hist_kernel<9>, dim3(nblocks), dim3(RS_THREADS), 0, stream, keys, n, shift, tpb, u32(RS_TILE_REC), rs_hist.p);
			else hipLaunchKernelGGL(rs_hist_kernel<8>, dim3(nblocks), dim3(RS_THREADS), 0, stream, keys, n, shift, tpb, u32(RS_TILE_REC), rs_hist.p);
		});
		timed(scan_s.c_str(), double(radix) * nblocks * 8, [&] {
			hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(radix), dim3(256), 0, stream, rs_hist.p, nblocks, rs_row_total.p);
			if (ps.bits == 9) hipLaunchKernelGGL(rs_scan_totals_kernel<512>, dim3(1), dim3(512), 0, stream, rs_row_total.p, rs_digit_base.p);
			else hipLaunchKernelGGL(rs_scan_totals_kernel<256>, dim3(1), dim3(256), 0, stream, rs_row_total.p, rs_digit_base.p);
		});
		timed(scatter_name, double(n) * 2 * (8 + val_bytes), [&] {
			rs_launch(val_bytes, ps.bits, dim3(nblocks), stream, keys, vals, keys_alt, vals_alt, n, shift, tpb, rs_hist.p, rs_digit_base.p);
		});
		std::swap(keys, keys_alt);
		std::swap(vals, vals_alt);
	}
}

// ------------------------------------------------------------------------------------------------
// stage: splitter sort + reads -> molecules (k_ssort.h)
// ------------------------------------------------------------------------------------------------
// Which path sorts the reads: "splitter" (two partitions on sampled splitters + the LDS-resident finishing sort fused
// with the molecule reduce) for the keys-only / key + mark byte layouts from DROPEST_SSORT_MIN reads on (default 2^22),
// the LSD radix sort otherwise and as the fall-back when a fine bucket does not fit the LDS sort (one molecule with more
// than ~8 000 reads).  DROPEST_SORT=lsd | splitter forces either (tests run both on the same streams).
// Does the LDS apply the lanes of one atomic instruction that hit the same address in lane order?  (ss_local's one-instruction
// ranking is stable only then.)  Checked once per device on 256 random digit patterns with 1 .. 64 distinct values.
static std::mutex g_lds_order_mu;
static std::map<int, bool> g_lds_order_known;
// a pass found a bucket out of order behind the one-atomic ranking: this device ranks with ballots from now on
static void lds_atomics_mark_unordered(int device) {
	std::lock_guard<std::mutex> lk(g_lds_order_mu);
	g_lds_order_known[device] = false;
	fprintf(stderr, "[dropest_amd] device %d: a bucket came out of its LDS sort unsorted behind the one-atomic ranking; the finishing sort ranks with ballots from now on (slower, same results)\n", device);
}
static bool lds_atomics_lane_ordered(int device, hipStream_t stream) {
	std::mutex &mu = g_lds_order_mu;
	std::map<int, bool> &known = g_lds_order_known;
	std::lock_guard<std::mutex> lk(mu);
	auto it = known.find(device);
	if (it != known.end()) return it->second;
	if (getenv("DROPEST_SS_BALLOT_RANK")) return known[device] = false;
	const u32 rounds = 256;
	std::vector<u32> h(size_t(rounds) * 64), got(size_t(rounds) * 64);
	u32 x = 2463534242u;
	for (u32 r = 0; r < rounds; ++r) {
		const u32 nd = 1u + r % 64u;
		for (u32 l = 0; l < 64; ++l) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; h[size_t(r) * 64 + l] = (x >> 8) % nd; }
	}
	DevBuf<u32> d_in, d_out;
	d_in.alloc(h.size()); d_out.alloc(h.size());
	HIP_CHECK(hipMemcpyAsync(d_in.p, h.data(), h.size() * 4, hipMemcpyHostToDevice, stream));
	hipLaunchKernelGGL(ss_lds_order_probe_kernel, dim3(1), dim3(64), 0, stream, d_in.p, rounds, d_out.p);
	HIP_CHECK(hipGetLastError());
	HIP_CHECK(hipMemcpyAsync(got.data(), d_out.p, got.size() * 4, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(stream_wait(stream));
	bool ordered = true;
	for (u32 r = 0; r < rounds && ordered; ++r) {
		u32 seen[64] = {0};
		for (u32 l = 0; l < 64; ++l) { const u32 d = h[size_t(r) * 64 + l]; if (got[size_t(r) * 64 + l] != seen[d]) ordered = false; ++seen[d]; }
	}
	// (said once per device: results never depend on it -- every pass verifies its buckets --, the speed of ss_local does)
	if (!ordered) fprintf(stderr, "[dropest_amd] device %d: LDS atomics of one instruction are not applied in lane order here; the finishing sort of the splitter sort ranks with ballots (slower, same results)\n", device);
	return known[device] = ordered;
}

bool dropest_ctx::splitter_sort_reduce() {
	const u32 n = u32(n_reads);
	// fan-out: F1 coarse x F2 fine buckets (powers of two, 16 .. 512 each) of <= ~1600 records on average: nearly all fit
	// the small finishing launch (256 threads x 4 or 8 records), the tail and a hot molecule's bucket go to the big one.
	// (Measured at 1e8 reads: 256 x 256 buckets 4.05 ms for the whole sort, 256 x 512 buckets 4.2 ms -- the finer second
	// partition costs what the finishing launch gains.)
	const SsPlan plan = ss_plan(n_reads, chr_from_gene, layout.val_bytes);
	if (!plan.applicable) return false;
	const int tb = plan.tb, fb1 = plan.fb1, fb2 = plan.fb2;
	const bool wide = fb2 > 9;                        // more than 512 buckets per level: the MAXF = 1024 kernels
	const u32 F1 = 1u << fb1, Ff = 1u << fb2, F2 = F1 * Ff;
	const int ms = layout.mark_shift, VB = layout.val_bytes;
	(void)tb;
	// 64 samples per fine bucket: bucket sizes scatter by ~12 % around n / F2, so few exceed the small finishing launch
	uint64_t os_max = plan.os;
	if (const char *e = getenv("DROPEST_SSORT_OS")) os_max = uint64_t(std::max(1, atoi(e)));
	const u32 os = u32(std::max<uint64_t>(1, std::min<uint64_t>(os_max, uint64_t(n) / (uint64_t(F2) * 2))));
	const u32 n_sample = F2 * os;
	const u64 order_mask = ~((1ull << ms) - 1ull);
	const u64 varying = (counters.key_or ^ counters.key_and) & order_mask;
	// by reservation only when build_keys sized the buffers for it (same plan, same pass) and no region overflowed earlier in this pass
	const bool reserve = plan.reserve && !ss_no_reserve && keys_a.n >= plan.span && keys_b.n >= plan.span &&
	                     (!VB || (vals_a.bytes() >= plan.span && vals_b.bytes() >= plan.span));
	const size_t span = reserve ? size_t(plan.span) : size_t(n);   // address space of the fine buckets (and of ss_local's sparse rows)

	HostStage hs(this, "splitter_sort");
	u64 *keys = keys_a.p, *keys_alt = keys_b.p;
	uint8_t *vals = reinterpret_cast<uint8_t *>(vals_a.p), *vals_alt = reinterpret_cast<uint8_t *>(vals_b.p);

	// sample -> sorted -> splitters (build_keys_fused took the sample from the reads and has the keys in their coarse regions already)
	const bool in_l1 = keys_in_l1 && reserve;
	if (keys_in_l1 && !reserve) throw InvalidError("internal: keys partitioned by the key pass, but the sort does not place by reservation");
	// key + mark byte layout with the key filling the word (C3): the first level hands on ONE word per record -- the key counted from its
	// coarse bucket's lower splitter, the mark in the three bits that frees (k_ssort.h: REBASE) -- and everything behind it runs on keys only
	const bool rebase = VB == 1 && ms == 0 && reserve && !in_l1 && !getenv("DROPEST_SS_NO_REBASE");
	const int VB2 = rebase ? 0 : VB, ms2 = rebase ? 3 : ms;          // layout behind the first level
	const u64 rebase_mask = rebase ? ~((1ull << layout.umi_bits) - 1ull) : 0ull;
	if (!in_l1) {
		ss_sample_a.ensure(n_sample); ss_sample_b.ensure(n_sample); ss_fine.ensure(F2); ss_coarse.ensure(F1);
		timed("ss_sample", double(n_sample) * 16, [&] {
			hipLaunchKernelGGL(ss_sample_kernel, dim3(div_up(n_sample, 256)), dim3(256), 0, stream, keys, n, ms, n_sample, ss_sample_a.p);
		});
		ss_splitters_from_sample(n_sample, os, Ff, F2, varying >> ms);
	}

	const u32 n_tiles = div_up(n, SS_TILE);
	u32 nblocks = std::min<u32>(n_tiles, 1024);
	const u32 tpb = div_up(n_tiles, nblocks);
	nblocks = div_up(n_tiles, tpb);
	const u32 parts = std::max<u32>(1, std::min<u32>(16, 2048 / F1));
	ss_bucket_base.ensure(F2); ss_bucket_cnt.ensure(F2); scalars.ensure(16);
	HIP_CHECK(hipMemsetAsync(scalars.p, 0, 16, stream));   // [0] largest bucket, [1] molecule total, [2] a bucket was out of order after its sort, [3] a region overflowed
	u32 max_cnt = 0;
	// finishing sort (k_ssort.h: ss_local): sparse molecule rows at each bucket's own record offset (key rows re-use the partition's
	// alternate buffer), then a scan of the per-bucket row counts and the compaction into the dense table.  The small launch skips buckets
	// beyond its LDS by itself, so it is queued BEFORE the host reads the partition's scalars back (a round trip of ~25 us the device would
	// otherwise sit out).
	static const u32 SMALL_MAX = [] { const char *e = getenv("DROPEST_SS_SMALL_MAX"); return e && atoi(e) >= 256 && atoi(e) <= 2048 ? u32(atoi(e)) : 2048u; }();   // (experiments)
	const bool atomic_rank = lds_atomics_lane_ordered(cfg.device, stream);
	ss_tmp.ensure(span * 2 + 2); ss_n_loc.ensure(F2); ss_prefix.ensure(F2); ss_chunk.ensure(1024);
	SsLocalArgs a{};
	a.keys = keys; a.vals = vals; a.bucket_base = ss_bucket_base.p; a.bucket_cnt = ss_bucket_cnt.p; a.n_buckets = F2; a.ms = ms2;
	if (rebase) { a.rebase_coarse = ss_coarse.p; a.rebase_mask = rebase_mask; a.rebase_div = Ff; }
	a.t_key = keys_alt; a.t_reads = ss_tmp.p; a.t_agg = ss_tmp.p + span; a.n_loc = ss_n_loc.p;
	if (const char *e = getenv("DROPEST_SS_DEBUG")) a.debug = u32(atoi(e));
	a.order_flag = scalars.p + 2;
	a.atomic_below = u32(ms2 + layout.umi_bits);   // the UMI field: random digits
	if (const char *e = getenv("DROPEST_SS_ATOMIC_BELOW")) a.atomic_below = u32(atoi(e));
	a.cap = SMALL_MAX; a.skip_above = SMALL_MAX;
	// the (cell, gene) table out of the compaction (k_ssort.h: ss_compact_cg) instead of seg_count + seg_reduce over the dense molecule table
	const bool fused_cg_off = getenv("DROPEST_SS_NO_FUSED_CG") != nullptr;   // (read at every pass: the soak scripts run both ways in one process)
	const bool fuse_cg = !fused_cg_off;
	cg_from_sort = false;
	if (fuse_cg) {
		ss_cg_loc.ensure(F2); ss_cg_cnt.ensure(F2); ss_cg_prefix.ensure(F2);
		HIP_CHECK(hipMemsetAsync(ss_cg_loc.p, 0, size_t(F2) * 4, stream));
		a.cg_loc = ss_cg_loc.p; a.cg_shift = ms2 + layout.umi_bits;
	}
	auto launch_small = [&] {
		static const int wave_mode = [] { const char *e = getenv("DROPEST_SS_LOCAL_WAVE"); return e ? atoi(e) : 0; }();
		if ((wave_mode == 64 || wave_mode == 128) && atomic_rank) {
			const size_t lds1 = ss_local_lds_bytes(a.cap, wave_mode);
			timed(VB2 ? "ss_local:key+1B" : "ss_local:keys", double(n) * (8 + VB2), [&] {
				if (wave_mode == 64) { if (VB2) hipLaunchKernelGGL((ss_local_wave_kernel<64, 1, true>), dim3(F2), dim3(64), lds1, stream, a); else hipLaunchKernelGGL((ss_local_wave_kernel<64, 0, true>), dim3(F2), dim3(64), lds1, stream, a); }
				else { if (VB2) hipLaunchKernelGGL((ss_local_wave_kernel<128, 1, true>), dim3(F2), dim3(128), lds1, stream, a); else hipLaunchKernelGGL((ss_local_wave_kernel<128, 0, true>), dim3(F2), dim3(128), lds1, stream, a); }
			});
			return;
		}
		const size_t lds = ss_local_lds_bytes(a.cap, 256);
		timed(VB2 ? "ss_local:key+1B" : "ss_local:keys", double(n) * (8 + VB2), [&] {   // + 16 B per molecule row, added below once n_mol is known
			if (atomic_rank) { if (VB2) hipLaunchKernelGGL((ss_local_kernel<1, true>), dim3(F2), dim3(256), lds, stream, a); else hipLaunchKernelGGL((ss_local_kernel<0, true>), dim3(F2), dim3(256), lds, stream, a); }
			else if (VB2) hipLaunchKernelGGL(ss_local_kernel<1>, dim3(F2), dim3(256), lds, stream, a);
			else hipLaunchKernelGGL(ss_local_kernel<0>, dim3(F2), dim3(256), lds, stream, a);
		});
	};
	if (reserve) {
		// both partitions by reservation (k_ssort.h): a tile takes its places in the buckets' regions with one atomic per bucket; no histograms
		const u32 cap1 = u32(plan.cap1), cap2 = u32(plan.cap2), CS1 = 32;
		ss_cursors.ensure(size_t(F1) * CS1 + F2);
		if (!in_l1) HIP_CHECK(hipMemsetAsync(ss_cursors.p, 0, (size_t(F1) * CS1 + F2) * 4, stream));
		else HIP_CHECK(hipMemcpyAsync(scalars.p + 3, ks_flag.p, 4, hipMemcpyDeviceToDevice, stream));   // (a coarse region the key pass overflowed)
		u32 *cur1 = ss_cursors.p, *cur2 = ss_cursors.p + size_t(F1) * CS1;
		const u32 probe = [] { const char *e = getenv("DROPEST_SS_PROBE"); return e ? u32(atoi(e)) : 0u; }();
		const bool no256 = getenv("DROPEST_SS_NO_F256") != nullptr;
		const u32 rebase_bits = [] { const char *e = getenv("DROPEST_SS_REBASE_BITS"); return e ? u32(std::min(61, std::max(1, atoi(e)))) : 61u; }();
		const SsReserve r1{cur1, CS1, cap1, 0u, scalars.p + 3, probe, rebase_bits}, r2{cur2, 1u, cap2, 0u, scalars.p + 3, probe, 61u};
		if (!in_l1) timed(VB ? "ss_scatter:L1:key+1B" : "ss_scatter:L1:keys", double(n) * (8 + VB + 8 + VB2), [&] {
			auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(nblocks), dim3(SS_T), 0, stream, keys, vals, keys_alt, vals_alt, n, ms, fb1, ss_coarse.p, tpb, r1, rebase_mask); };
			if (rebase) { if (wide) go(ss_scatter_res_l1_kernel<1, 1024, true>); else if (fb1 <= 8 && !no256) go(ss_scatter_res_l1_kernel<1, 256, true>); else go(ss_scatter_res_l1_kernel<1, 512, true>); }
			else if (wide) { if (VB) go(ss_scatter_res_l1_kernel<1, 1024>); else go(ss_scatter_res_l1_kernel<0, 1024>); }
			else if (fb1 <= 8 && !no256) { if (VB) go(ss_scatter_res_l1_kernel<1, 256>); else go(ss_scatter_res_l1_kernel<0, 256>); }   // 45 KB of LDS: three workgroups per CU
			else { if (VB) go(ss_scatter_res_l1_kernel<1, 512>); else go(ss_scatter_res_l1_kernel<0, 512>); }
		});
		timed(VB2 ? "ss_scatter:L2:key+1B" : "ss_scatter:L2:keys", double(n) * 2 * (8 + VB2), [&] {
			auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(F1 * parts), dim3(SS_T), 0, stream, keys_alt, vals_alt, keys, vals, ms2, fb2, ss_fine.p, cur1, CS1, cap1, parts, r2,
			                                                 rebase ? ss_coarse.p : static_cast<const u64 *>(nullptr), rebase_mask); };
			if (wide) { if (VB2) go(ss_scatter_res_l2_kernel<1, 1024>); else go(ss_scatter_res_l2_kernel<0, 1024>); }
			else if (fb2 <= 8 && !no256) { if (VB2) go(ss_scatter_res_l2_kernel<1, 256>); else go(ss_scatter_res_l2_kernel<0, 256>); }
			else { if (VB2) go(ss_scatter_res_l2_kernel<1, 512>); else go(ss_scatter_res_l2_kernel<0, 512>); }
		});
		timed("ss_scan", double(F2) * 12, [&] {
			hipLaunchKernelGGL(ss_res_buckets_kernel, dim3(div_up(F2, 256)), dim3(256), 0, stream, cur2, F2, cap2, ss_bucket_base.p, ss_bucket_cnt.p, scalars.p);
		});
		launch_small();   // (an overflowed region holds `cap` records: the launch is harmless, its rows are discarded below)
		u32 head[4] = {0, 0, 0, 0};
		fetch(head, scalars.p, 16);
		max_cnt = head[0];
		if (head[3]) {
			// A bucket outgrew its region (a heavy key: the sample's quantiles say nothing about a molecule with a good share of all reads).
			// The partitions consumed the keys: they are built again and this pass takes the counting partitions.
			stats["count:ss_reserve_overflow"].launches += 1;
			ss_no_reserve = true;
			build_keys(false);
			return splitter_sort_reduce();
		}
		if (max_cnt > SS_LOCAL_MAX) { build_keys(false, false); return false; }   // (cannot happen while a region holds fewer records than the LDS sort takes)
	} else {
	// L1: all records into the F1 coarse buckets
	rs_hist.ensure(size_t(F1) * nblocks); rs_row_total.ensure(1024); ss_base1.ensure(F1 + 1);
	timed("ss_hist:L1", double(n) * 8, [&] {
		if (wide) hipLaunchKernelGGL(ss_hist_l1_kernel<1024>, dim3(nblocks), dim3(SS_T), 0, stream, keys, n, ms, fb1, ss_coarse.p, tpb, rs_hist.p);
		else hipLaunchKernelGGL(ss_hist_l1_kernel<512>, dim3(nblocks), dim3(SS_T), 0, stream, keys, n, ms, fb1, ss_coarse.p, tpb, rs_hist.p);
	});
	timed("ss_scan", double(F1) * nblocks * 8, [&] {
		hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(F1), dim3(256), 0, stream, rs_hist.p, nblocks, rs_row_total.p);
		if (wide) hipLaunchKernelGGL(ss_scan_totals_kernel<1024>, dim3(1), dim3(1024), 0, stream, rs_row_total.p, F1, n, ss_base1.p);
		else hipLaunchKernelGGL(ss_scan_totals_kernel<512>, dim3(1), dim3(512), 0, stream, rs_row_total.p, F1, n, ss_base1.p);
	});
	timed(VB ? "ss_scatter:L1:key+1B" : "ss_scatter:L1:keys", double(n) * 2 * (8 + VB), [&] {
		auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(nblocks), dim3(SS_T), 0, stream, keys, vals, keys_alt, vals_alt, n, ms, fb1, ss_coarse.p, tpb, rs_hist.p, ss_base1.p); };
		if (wide) { if (VB) go(ss_scatter_l1_kernel<1, 1024>); else go(ss_scatter_l1_kernel<0, 1024>); }
		else { if (VB) go(ss_scatter_l1_kernel<1, 512>); else go(ss_scatter_l1_kernel<0, 512>); }
	});

	// L2: every coarse bucket into its own fine buckets
	ss_cnt2.ensure(size_t(F2) * parts);
	timed("ss_hist:L2", double(n) * 8, [&] {
		if (wide) hipLaunchKernelGGL(ss_hist_l2_kernel<1024>, dim3(F1 * parts), dim3(SS_T), 0, stream, keys_alt, ms, fb2, ss_fine.p, ss_base1.p, parts, ss_cnt2.p);
		else hipLaunchKernelGGL(ss_hist_l2_kernel<512>, dim3(F1 * parts), dim3(SS_T), 0, stream, keys_alt, ms, fb2, ss_fine.p, ss_base1.p, parts, ss_cnt2.p);
	});
	timed("ss_scan", double(F2) * parts * 8, [&] {
		if (wide) hipLaunchKernelGGL(ss_scan_seg_kernel<1024>, dim3(F1), dim3(1024), 0, stream, ss_cnt2.p, Ff, parts, ss_base1.p, ss_bucket_base.p, ss_bucket_cnt.p, scalars.p);
		else hipLaunchKernelGGL(ss_scan_seg_kernel<512>, dim3(F1), dim3(512), 0, stream, ss_cnt2.p, Ff, parts, ss_base1.p, ss_bucket_base.p, ss_bucket_cnt.p, scalars.p);
	});
	fetch(&max_cnt, scalars.p, 4);
	if (max_cnt > SS_LOCAL_MAX) return false;   // keys_a / vals_a are untouched: the LSD sort takes over
	timed(VB ? "ss_scatter:L2:key+1B" : "ss_scatter:L2:keys", double(n) * 2 * (8 + VB), [&] {
		auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(F1 * parts), dim3(SS_T), 0, stream, keys_alt, vals_alt, keys, vals, ms, fb2, ss_fine.p, ss_base1.p, parts, ss_cnt2.p); };
		if (wide) { if (VB) go(ss_scatter_l2_kernel<1, 1024>); else go(ss_scatter_l2_kernel<0, 1024>); }
		else { if (VB) go(ss_scatter_l2_kernel<1, 512>); else go(ss_scatter_l2_kernel<0, 512>); }
	});
	}

	if (!reserve) launch_small();
	if (max_cnt > SMALL_MAX) {   // the few buckets beyond the small launch (the tail of the size distribution, a hot molecule), listed by the host
		std::vector<u32> cnts(F2), medium, big;
		fetch(cnts.data(), ss_bucket_cnt.p, size_t(F2) * 4);
		for (u32 b = 0; b < F2; ++b) if (cnts[b] > SMALL_MAX) (cnts[b] <= 4096 ? medium : big).push_back(b);
		ss_big_list.ensure(medium.size() + big.size());
		if (!medium.empty()) HIP_CHECK(hipMemcpyAsync(ss_big_list.p, medium.data(), medium.size() * 4, hipMemcpyHostToDevice, stream));
		if (!big.empty()) HIP_CHECK(hipMemcpyAsync(ss_big_list.p + medium.size(), big.data(), big.size() * 4, hipMemcpyHostToDevice, stream));
		auto launch = [&](auto kernel, u32 threads, u32 cap, const u32 *list, size_t count) {
			SsLocalArgs g = a;
			g.big_list = list; g.cap = cap; g.skip_above = cap;
			const size_t lds = ss_local_lds_bytes(cap, int(threads));
			HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
			hipLaunchKernelGGL(kernel, dim3(u32(count)), dim3(threads), lds, stream, g);
		};
		timed("ss_local:big", 0, [&] {
			if (atomic_rank) {
				if (!medium.empty()) { if (VB2) launch(ss_local_big_kernel<256, 1, true>, 256, 4096, ss_big_list.p, medium.size()); else launch(ss_local_big_kernel<256, 0, true>, 256, 4096, ss_big_list.p, medium.size()); }
				if (!big.empty()) { if (VB2) launch(ss_local_big_kernel<512, 1, true>, 512, SS_LOCAL_MAX, ss_big_list.p + medium.size(), big.size()); else launch(ss_local_big_kernel<512, 0, true>, 512, SS_LOCAL_MAX, ss_big_list.p + medium.size(), big.size()); }
			} else {
				if (!medium.empty()) { if (VB2) launch(ss_local_big_kernel<256, 1>, 256, 4096, ss_big_list.p, medium.size()); else launch(ss_local_big_kernel<256, 0>, 256, 4096, ss_big_list.p, medium.size()); }
				if (!big.empty()) { if (VB2) launch(ss_local_big_kernel<512, 1>, 512, SS_LOCAL_MAX, ss_big_list.p + medium.size(), big.size()); else launch(ss_local_big_kernel<512, 0>, 512, SS_LOCAL_MAX, ss_big_list.p + medium.size(), big.size()); }
			}
		});
		HIP_CHECK(stream_wait(stream));   // the host lists must outlive their copies
	}
	const u32 n_chunks = div_up(F2, 1024);
	timed("ss_scan", double(F2) * 12, [&] {
		hipLaunchKernelGGL(ss_chunk_sums_kernel<1024>, dim3(n_chunks), dim3(1024), 0, stream, ss_n_loc.p, F2, ss_chunk.p);
		hipLaunchKernelGGL(ss_prefix_kernel<1024>, dim3(n_chunks), dim3(1024), 0, stream, ss_n_loc.p, F2, ss_chunk.p, n_chunks, ss_prefix.p, scalars.p + 1);
		if (fuse_cg) {   // heads per bucket (those inside it + whether its first molecule opens a pair), their prefix, their total
			hipLaunchKernelGGL(ss_cg_counts_kernel, dim3(div_up(F2, 256)), dim3(256), 0, stream, ss_bucket_base.p, ss_n_loc.p, keys_alt, layout.umi_bits, F2, ss_cg_loc.p, ss_cg_cnt.p);
			hipLaunchKernelGGL(ss_chunk_sums_kernel<1024>, dim3(n_chunks), dim3(1024), 0, stream, ss_cg_cnt.p, F2, ss_chunk.p);
			hipLaunchKernelGGL(ss_prefix_kernel<1024>, dim3(n_chunks), dim3(1024), 0, stream, ss_cg_cnt.p, F2, ss_chunk.p, n_chunks, ss_cg_prefix.p, scalars.p + 4);
		}
	});
	u32 total_flag[4] = {0, 0, 0, 0};   // molecules, order flag, (overflow flag), (cell, gene) pairs
	fetch(total_flag, scalars.p + 1, 16);
	if (total_flag[1]) {
		// A bucket came out of its LDS sort unsorted (the in-kernel check of ss_local, every pass): never silently.  The partitions
		// consumed the keys, so they are built again and the LSD sort takes the pass; the one-atomic ranking is off for this device.
		if (atomic_rank) lds_atomics_mark_unordered(cfg.device);
		stats["count:ss_order_violation"].launches += 1;
		build_keys(false, false);   // (the LSD sort wants the plain key array)
		return false;
	}
	const u32 total = total_flag[0];
	n_mol = total;
	for (auto it = pending.rbegin(); it != pending.rend(); ++it)   // the sparse rows ss_local wrote: 16 B per molecule (profiling only)
		if (it->name.compare(0, 10, "ss_local:k") == 0) { it->bytes += double(n_mol) * 16; break; }
	mol_key.ensure(size_t(n_mol) + 1);
	for (DevBuf<u32> *b : {&mol_reads, &mol_mark, &mol_exon, &mol_intron}) b->ensure(size_t(n_mol) + 1);
	SsCompactArgs c{};
	c.bucket_base = ss_bucket_base.p; c.n_loc = ss_n_loc.p; c.prefix = ss_prefix.p; c.n_buckets = F2;
	c.t_key = keys_alt; c.t_reads = ss_tmp.p; c.t_agg = ss_tmp.p + span;
	c.mol_key = mol_key.p; c.mol_reads = mol_reads.p; c.mol_mark = mol_mark.p; c.mol_exon = mol_exon.p; c.mol_intron = mol_intron.p;
	if (fuse_cg) {
		n_cg = total_flag[3];
		cg_key.ensure(size_t(n_cg) + 1); cg_mol_begin.ensure(size_t(n_cg) + 1);
		for (DevBuf<u32> *b : {&cg_n_all, &cg_n_req, &cg_reads_all, &cg_reads_req, &cg_exon, &cg_intron}) b->ensure(size_t(n_cg) + 1);
		SsCompactCgArgs g{};
		g.bucket_base = ss_bucket_base.p; g.n_loc = ss_n_loc.p; g.prefix = ss_prefix.p; g.cg_loc = ss_cg_loc.p; g.cg_cnt = ss_cg_cnt.p; g.cg_prefix = ss_cg_prefix.p;
		g.n_buckets = F2; g.t_key = keys_alt; g.t_reads = ss_tmp.p; g.t_agg = ss_tmp.p + span;
		g.mol_key = mol_key.p; g.mol_reads = mol_reads.p; g.mol_mark = mol_mark.p; g.mol_exon = mol_exon.p; g.mol_intron = mol_intron.p;
		g.cg_key = cg_key.p; g.cg_mol_begin = cg_mol_begin.p;
		g.out[0] = cg_n_all.p; g.out[1] = cg_n_req.p; g.out[2] = cg_reads_all.p; g.out[3] = cg_reads_req.p; g.out[4] = cg_exon.p; g.out[5] = cg_intron.p;
		g.umi_bits = layout.umi_bits; g.query_mask = query_mask; g.n_cg = n_cg;
		if (const char *e = getenv("DROPEST_CG_DBG")) g.dbg = u32(atoi(e));
		timed("ss_compact:cell_gene", double(n_mol) * (16 + 24) + double(n_cg) * 36, [&] {
			hipLaunchKernelGGL(ss_cg_zero_borders_kernel, dim3(div_up(F2, 256)), dim3(256), 0, stream, g);
			hipLaunchKernelGGL(ss_compact_cg_kernel<false>, dim3(div_up(F2, 4)), dim3(256), 0, stream, g);
		});
		HIP_CHECK(hipMemcpyAsync(cg_mol_begin.p + n_cg, &n_mol, 4, hipMemcpyHostToDevice, stream));   // row i owns molecules [cg_mol_begin[i], cg_mol_begin[i + 1])
		cg_from_sort = true;
	} else
	timed("ss_compact", double(n_mol) * (16 + 24), [&] {
		hipLaunchKernelGGL(ss_compact_kernel, dim3(div_up(F2, 4)), dim3(256), 0, stream, c);
	});
	for (DevBuf<u32> *b : {&mol_reads, &mol_mark, &mol_exon, &mol_intron}) HIP_CHECK(hipMemsetAsync(b->p + n_mol, 0, 4, stream));   // sentinel row
	main_sort_passes = 3; main_sort_kind = 1;
	return true;
}

// ------------------------------------------------------------------------------------------------
// stage: segmented reduces
// ------------------------------------------------------------------------------------------------
// Runs count -> scan -> reduce for one policy.  `prepare(total)` allocates + zeroes the outputs and wires the
// policy's pointers once the number of runs is known.  Returns the number of runs.
// Grid of the persistent seg_reduce kernel: as many workgroups as are resident at once (occupancy x CUs), at most one
// per tile.
template <class P>
static u32 seg_reduce_grid(u32 tiles) {
	static const u32 resident = [] {
		int dev = 0, cus = 0, per_cu = 0;
		HIP_CHECK(hipGetDevice(&dev));
		HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
		HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, seg_reduce_kernel<P>, SR_THREADS, 0));
		return u32(std::max(1, cus) * std::max(1, per_cu));
	}();
	return std::max<u32>(1u, std::min(tiles, resident));
}

template <class P, class Prep>
static u32 run_segmented_reduce(dropest_ctx &c, const char *tag, P &policy, u32 n, double bytes_per_row, Prep &&prepare,
                                double out_bytes_per_run = 0) {
	if (n == 0) {
		prepare(0);
		if constexpr (!P::DIRECT) for (int ch = 0; ch < P::NV; ++ch) if (policy.out[ch]) HIP_CHECK(hipMemsetAsync(policy.out[ch], 0, 4, c.stream));
		return 0;
	}
	const u32 tiles = div_up(n, SR_THREADS * P::ITEMS);
	c.tile_counts.ensure(tiles); c.tile_prefix.ensure(tiles); c.scalars.ensure(16);
	const std::string n_count = std::string("seg_count:") + tag, n_reduce = std::string("seg_reduce:") + tag;
	c.timed(n_count.c_str(), double(n) * 8, [&] {
		hipLaunchKernelGGL(seg_count_kernel<P>, dim3(tiles), dim3(SR_THREADS), 0, c.stream, policy, n, c.tile_counts.p);
	});
	c.timed("scan_small", double(tiles) * 8, [&] {
		c.scan_counts(c.tile_counts.p, c.tile_prefix.p, tiles, c.scalars.p);
	});
	u32 total = 0;
	c.fetch(&total, c.scalars.p, 4);
	prepare(total);
	if constexpr (!P::DIRECT) {   // the rows the reduce adds into with atomics (tile-border runs) + the sentinel row; see k_segreduce.h
		hipLaunchKernelGGL(seg_zero_borders_kernel<P>, dim3(div_up(tiles, 256)), dim3(256), 0, c.stream, policy, c.tile_prefix.p, tiles, total);
	}
	c.timed(n_reduce.c_str(), double(n) * bytes_per_row + double(total) * out_bytes_per_run, [&] {
		hipLaunchKernelGGL(seg_reduce_kernel<P>, dim3(seg_reduce_grid<P>(tiles)), dim3(SR_THREADS), 0, c.stream, policy, n, c.tile_prefix.p);
	});
	return total;
}

static void zero_async(dropest_ctx &c, void *p, size_t bytes) { HIP_CHECK(hipMemsetAsync(p, 0, bytes, c.stream)); }
// several u32 arrays of one length zeroed by ONE launch (a memset each is a launch or two of ~5 us: the seven per-cell arrays of a pass
// took fourteen); the allocations are 256-byte aligned: 16-byte stores, the last words one by one
struct ZeroArrays { uint32_t *p[8]; uint32_t n_arrays; };
__global__ __launch_bounds__(256) void zero_arrays_kernel(ZeroArrays a, size_t words) {
	const size_t quads = words / 4;
	for (uint32_t k = 0; k < a.n_arrays; ++k) {
		uint4 *q = reinterpret_cast<uint4 *>(a.p[k]);
		for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < quads; i += size_t(gridDim.x) * 256) q[i] = make_uint4(0, 0, 0, 0);
		if (blockIdx.x == 0 && threadIdx.x < words - quads * 4) a.p[k][quads * 4 + threadIdx.x] = 0;
	}
}

void dropest_ctx::reduce_all() {
	const u32 n = u32(n_reads);
	u64 *keys = keys_a.p, *keys_alt = keys_b.p;
	u32 *vals = vals_a.p, *vals_alt = vals_b.p;
	// a mark folded under the key needs no ordering: its bits are masked out of the sort
	const u64 order_mask = ~((1ull << layout.mark_shift) - 1ull);
	const u64 varying = (counters.key_or ^ counters.key_and) & order_mask;
	if (splitter_sort_reduce()) {
		n_chr_rows = 0;
		if (!cg_from_sort) reduce_molecules_to_cell_gene();
		reduce_cell_gene_to_cells();
		HIP_CHECK(stream_wait(stream));
		return;
	}
	main_sort_passes = u32(plan_radix_passes(varying).size()); main_sort_kind = 0;
	radix_sort(keys, vals, keys_alt, vals_alt, n, varying, layout.val_bytes);

	if (chr_from_gene) {
		// reads -> molecules with exon / intron read counts; no second pass over the reads for the chromosomes
		auto run = [&](auto &p) {
			p.keys = keys;
			n_mol = run_segmented_reduce(*this, "molecules", p, n, 8 + layout.val_bytes, [&](u32 total) {
				mol_key.ensure(total + 1);
				for (DevBuf<u32> *b : {&mol_reads, &mol_mark, &mol_exon, &mol_intron}) b->ensure(total + 1);
				p.mol_key = mol_key.p; p.out[0] = mol_reads.p; p.out[1] = mol_mark.p; p.out[2] = mol_exon.p; p.out[3] = mol_intron.p;
			}, 24);
		};
		if (layout.val_bytes == 0) { ReadsToMoleculesX<0> p{}; run(p); }
		else { ReadsToMoleculesX<1> p{}; p.marks = reinterpret_cast<const uint8_t *>(vals); run(p); }
		n_chr_rows = 0;
	} else {
		// general layout: a gene may sit on several chromosomes, the chromosome travels with every read
		{
			ReadsToMolecules p{};
			p.keys = keys; p.vals = vals;
			n_mol = run_segmented_reduce(*this, "molecules", p, n, 12, [&](u32 total) {
				mol_key.ensure(total + 1); mol_reads.ensure(total + 1); mol_mark.ensure(total + 1);
				p.mol_key = mol_key.p; p.out[0] = mol_reads.p; p.out[1] = mol_mark.p;
			}, 16);
		}
		{
			ReadsToChrRows p{};
			p.keys = keys; p.vals = vals;
			p.cell_shift = layout.gene_bits + layout.umi_bits; p.umi_bits = layout.umi_bits; p.gene_mask = layout.gene_none;
			n_chr_rows = run_segmented_reduce(*this, "chr_rows", p, n, 12 + 2, [&](u32 total) {
				chr_row_key.ensure(total + 1); chr_exon.ensure(total + 1); chr_intron.ensure(total + 1); chr_inter.ensure(total + 1);
				p.row_key = chr_row_key.p; p.out[0] = chr_exon.p; p.out[1] = chr_intron.p; p.out[2] = chr_inter.p;
			});
		}
	}
	reduce_molecules_to_cell_gene();
	reduce_cell_gene_to_cells();
	HIP_CHECK(stream_wait(stream));
	// the sort ping-pong buffers stay allocated: the next run_set_initialized on this context reuses them
}

void dropest_ctx::reduce_molecules_to_cell_gene() {
	auto prepare_common = [&](u32 total) {
		cg_key.ensure(total + 1); cg_mol_begin.ensure(total + 1);
		for (DevBuf<u32> *b : {&cg_n_all, &cg_n_req, &cg_reads_all, &cg_reads_req}) b->ensure(total + 1);
	};
	if (chr_from_gene) {
		MoleculesToCellGeneX p{};
		p.mol_key = mol_key.p; p.mol_reads = mol_reads.p; p.mol_mark = mol_mark.p; p.mol_exon = mol_exon.p; p.mol_intron = mol_intron.p;
		p.umi_bits = layout.umi_bits; p.query_mask = query_mask;
		n_cg = run_segmented_reduce(*this, "cell_gene", p, n_mol, 24, [&](u32 total) {
			prepare_common(total);
			for (DevBuf<u32> *b : {&cg_exon, &cg_intron}) b->ensure(total + 1);
			p.cg_key = cg_key.p; p.cg_mol_begin = cg_mol_begin.p;
			p.out[0] = cg_n_all.p; p.out[1] = cg_n_req.p; p.out[2] = cg_reads_all.p; p.out[3] = cg_reads_req.p;
			p.out[4] = cg_exon.p; p.out[5] = cg_intron.p;
		}, 36);
	} else {
		MoleculesToCellGene p{};
		p.mol_key = mol_key.p; p.mol_reads = mol_reads.p; p.mol_mark = mol_mark.p;
		p.umi_bits = layout.umi_bits; p.query_mask = query_mask;
		n_cg = run_segmented_reduce(*this, "cell_gene", p, n_mol, 16, [&](u32 total) {
			prepare_common(total);
			p.cg_key = cg_key.p; p.cg_mol_begin = cg_mol_begin.p;
			p.out[0] = cg_n_all.p; p.out[1] = cg_n_req.p; p.out[2] = cg_reads_all.p; p.out[3] = cg_reads_req.p;
		}, 28);
	}
	// sentinel so that row i owns molecules [cg_mol_begin[i], cg_mol_begin[i+1])
	HIP_CHECK(hipMemcpyAsync(cg_mol_begin.p + n_cg, &n_mol, 4, hipMemcpyHostToDevice, stream));
}

void dropest_ctx::reduce_cell_gene_to_cells() {
	// DIRECT segmented reduce: output row = cell id, no count/scan pass; cells without rows stay zero
	CellGeneToCells p{};
	p.cg_key = cg_key.p; p.n_all = cg_n_all.p; p.n_req = cg_n_req.p; p.reads_all = cg_reads_all.p;
	p.gene_bits = layout.gene_bits; p.gene_mask = layout.gene_none;
	const size_t nc = size_t(n_cells) + 1;
	cell_cg_begin.ensure(nc); cell_cg_count.ensure(nc); cell_n_genes.ensure(nc); cell_req_genes.ensure(nc);
	cell_req_umis.ensure(nc); cell_total_umis.ensure(nc); cell_total_reads.ensure(nc);
	{
		ZeroArrays z{};
		for (DevBuf<u32> *b : {&cell_cg_begin, &cell_cg_count, &cell_n_genes, &cell_req_genes, &cell_req_umis, &cell_total_umis, &cell_total_reads})
			z.p[z.n_arrays++] = b->p;
		hipLaunchKernelGGL(zero_arrays_kernel, dim3(u32(std::min<size_t>(div_up(u32(std::min<size_t>(nc / 4 + 1, 0xFFFFFFFFull)), 256u), 2048u))), dim3(256), 0, stream, z, nc);
		HIP_CHECK(hipGetLastError());
	}
	p.cell_cg_begin = cell_cg_begin.p;
	p.out[0] = cell_n_genes.p; p.out[1] = cell_req_genes.p; p.out[2] = cell_req_umis.p;
	p.out[3] = cell_total_umis.p; p.out[4] = cell_total_reads.p; p.out[5] = cell_cg_count.p;
	if (n_cg == 0) return;
	const u32 tiles = div_up(n_cg, SR_THREADS * CellGeneToCells::ITEMS);
	timed("seg_reduce:cells", double(n_cg) * (24 + 4), [&] {
		hipLaunchKernelGGL(seg_reduce_kernel<CellGeneToCells>, dim3(seg_reduce_grid<CellGeneToCells>(tiles)), dim3(SR_THREADS), 0, stream, p, n_cg,
		                   static_cast<const u32 *>(nullptr));
	});
}

// ------------------------------------------------------------------------------------------------
// stage: real cells to the host; ordering (CellsDataContainer::update_filtered_gene_counts)
// ------------------------------------------------------------------------------------------------
void dropest_ctx::fetch_real_cells(bool at_init) {
	invalidate_prefetch();
	real.clear(); real_list_current = false;
	if (n_cells == 0) return;
	DevBuf<u32> &list = real_list; list.ensure(n_cells);
	scalars.ensure(16);
	const u32 tiles = div_up(n_cells, RC_TILE);
	tile_counts.ensure(tiles); tile_prefix.ensure(tiles);
	timed("flag_real", double(n_cells) * 8, [&] {
		hipLaunchKernelGGL(count_real_kernel, dim3(tiles), dim3(RC_THREADS), 0, stream, cell_n_genes.p, n_cells, min_before, tile_counts.p);
		scan_counts(tile_counts.p, tile_prefix.p, tiles, scalars.p);
		hipLaunchKernelGGL(write_real_kernel, dim3(tiles), dim3(RC_THREADS), 0, stream, cell_n_genes.p, n_cells, min_before,
		                   tile_prefix.p, list.p);
	});
	tail_mark("fetch_real_cells: flag kernels enqueued", true);
	u32 count = 0;
	fetch(&count, scalars.p, 4);
	tail_mark("count fetched");
	if (count == 0) return;
	DevBuf<CellRowPod> &rows = real_rows_dev; rows.ensure(count);
	CellArrays a{cell_cb.p, cell_first.p, cell_n_genes.p, cell_req_genes.p, cell_req_umis.p, cell_total_umis.p, cell_total_reads.p};
	timed("gather_cell_rows", double(count) * 72, [&] {
		hipLaunchKernelGGL(gather_cell_rows_kernel, dim3(div_up(count, 256)), dim3(256), 0, stream, a, list.p, 0u, count, rows.p);
	});
	// ids and rows come back in one wait, and the host mirror is filled from the pinned buffer on a few threads (2.5 M cells at C3 size)
	const size_t row_bytes = size_t(count) * sizeof(CellRowPod), id_off = (row_bytes + 15) & ~size_t(15);
	h_stage.ensure(id_off + size_t(count) * 4);
	HIP_CHECK(hipMemcpyAsync(h_stage.p, rows.p, row_bytes, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipMemcpyAsync(h_stage.p + id_off, list.p, size_t(count) * 4, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(stream_wait(stream));
	tail_mark("rows on the host");
	const CellRowPod *host_rows = reinterpret_cast<const CellRowPod *>(h_stage.p);
	const u32 *ids = reinterpret_cast<const u32 *>(h_stage.p + id_off);
	// cm_raw announced (dropest_set_raw_matrix_prefetch) and nothing in merge_and_filter can change it: its columns are exactly these
	// rows, in this order -- the emit and the copy to the host start HERE, before the host mirror below is filled (the matrices' way
	// over PCIe is the tail of a pass: every microsecond the first byte leaves earlier is one off the pass)
	if (at_init && auto_pf_form >= 0 && n_reads && merge_phase_changes_nothing() && (auto_pf_form != 1 || narrow_possible()))
		prefetch_raw_matrix(auto_pf_reads, auto_pf_form, host_rows, ids, count);
	tail_mark("prefetch enqueued");
	real.resize(count);   // ids arrive ascending (ordered compaction)
	parallel_ranges(count, [&](size_t b, size_t e, unsigned) {
		for (size_t i = b; i < e; ++i) {
			HostCell &h = real[i];
			h.id = ids[i];
			h.row = host_rows[i];
			h.merged = h.excluded = false;
		}
	});
	real_list_current = true;   // (real_list holds exactly these ids)
	real_pristine = true;       // ... and real_rows_dev their rows, until something changes a cell (sort_filtered orders them on the device)
	tail_mark("host mirror filled");
}

void dropest_ctx::request_filtered(u32 genes_threshold, int max_cells) {
	// the real-cell count is cheap and always current; the ordering is produced when somebody reads it
	filtered_threshold = genes_threshold; filtered_max_cells = max_cells; filtered_valid = false;
	n_real_now = 0;
	constexpr unsigned W = dropest::HostPool::MAX;
	uint64_t part[W + 1] = {0};   // (2.4e6 rows at C3 size: a millisecond or two for one thread)
	const unsigned workers = parallel_ranges(real.size(), [&](size_t b, size_t e, unsigned w) {
		uint64_t c = 0;
		for (size_t i = b; i < e; ++i) { const HostCell &h = real[i]; c += (!h.merged && !h.excluded && h.row.n_genes >= min_before); }
		part[w] = c;
	}, 100000, W);
	for (unsigned w = 0; w < workers; ++w) n_real_now += part[w];
}

const std::vector<uint64_t> &dropest_ctx::filtered_cells() {
	if (!filtered_valid) { HostStage hs(this, "sort_filtered"); sort_filtered(filtered_threshold, filtered_max_cells); }
	return filtered;
}

void dropest_ctx::sort_filtered(u32 genes_threshold, int max_cells) {
	// CellsDataContainer.cpp:250-276 with compare_cells :329-344, on compact keys
	auto passes = [&](const HostCell &h) {
		return !(h.merged || h.excluded || h.row.n_genes < min_before) && h.row.requested_genes >= genes_threshold;
	};
	// (Up to 1e5 cells the host orders them in a millisecond or two -- the 5e4 filtered cells of C3 AFTER the merge, whose device
	// read-back would cross PCIe behind cm_raw's prefetch: 10 ms of waiting for 200 KB.  The 2.4e6 candidates BEFORE the merge take the
	// device path: 3 ms against 60 on the host, and nothing else is on the link then.)
	size_t device_min = 100000;   // (C4's 1.5e5 candidates before the merge: 5.0 ms on the host, 1.15 on the device)
	if (const char *e = getenv("DROPEST_DEVICE_SORT_MIN")) device_min = size_t(std::max(1, atoi(e)));   // tests force the device path

	// Large lists (10^5..10^6 cells at BASELINE sizes) are ordered on the device: three stable LSD radix sorts
	// (barcode, then TOTAL_UMIS, then the packed sizes) when every barcode is a clean code of one length, so that
	// the numeric order of the codes IS the string order.  The key is total, so any correct sort gives the same list.
	// The host side of it (two passes over `real`, one over the result) runs on a few worker threads.
	const size_t R = real.size();
	// Nothing has touched the real cells since fetch_real_cells (no merge, no mutator): their rows and ids still stand on the device as they
	// were gathered, every one of them passes with threshold 0 -- the ordering before a barcode merge, 2.4e6 candidates at C3 -- and the
	// whole ordering runs there: no host pass over `real`, no columns over PCIe, 12 bytes per cell back (8.4 -> ~1.5 ms at C3 size).
	if (R >= device_min && R < 0x50000000ull && real_pristine && real_list_current && genes_threshold == 0 && !getenv("DROPEST_SORTF_HOST")) {
		HostStage st(this, "sort_filtered:device_only");
		const u32 m = u32(R);
		sort_cols.ensure(size_t(m) * 3); scalars64.ensure(8);
		keys_a.ensure(m); keys_b.ensure(m); vals_a.ensure(m); vals_b.ensure(m);
		u64 init[8] = {0, ~0ull, 0, ~0ull, 0, ~0ull, 64, 0};
		HIP_CHECK(hipMemcpyAsync(scalars64.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
		u64 *d_code = sort_cols.p, *d_umis = sort_cols.p + m, *d_sizes = sort_cols.p + 2 * size_t(m);
		hipLaunchKernelGGL(sortf_columns_kernel, dim3(std::min<u32>(div_up(m, 256), 2048u)), dim3(256), 0, stream, real_rows_dev.p, m, d_code, d_umis, d_sizes, scalars64.p);
		HIP_CHECK(hipGetLastError());
		u64 st8[8];
		fetch(st8, scalars64.p, sizeof(st8));
		if (st8[6] == st8[7] && !(st8[0] & ESCAPE_BIT)) {   // clean codes of one length: their numeric order is the string order
			u64 *k = keys_a.p, *k_alt = keys_b.p;
			u32 *v = vals_a.p, *v_alt = vals_b.p;
			hipLaunchKernelGGL(iota_kernel, dim3(div_up(m, 256)), dim3(256), 0, stream, v, m);
			HIP_CHECK(hipMemcpyAsync(k, d_code, size_t(m) * 8, hipMemcpyDeviceToDevice, stream));
			radix_sort(k, v, k_alt, v_alt, m, st8[0] ^ st8[1]);
			const std::pair<const u64 *, u64> more[2] = {{d_umis, st8[2] ^ st8[3]}, {d_sizes, st8[4] ^ st8[5]}};
			for (auto const &nx : more) {
				hipLaunchKernelGGL(gather_u64_kernel, dim3(div_up(m, 256)), dim3(256), 0, stream, nx.first, v, m, k);
				HIP_CHECK(hipGetLastError());
				radix_sort(k, v, k_alt, v_alt, m, nx.second);
			}
			sort_stage.ensure((size_t(m) * 3 + 1) / 2 + 1);
			u32 *out = reinterpret_cast<u32 *>(sort_stage.p);
			hipLaunchKernelGGL(sortf_out_kernel, dim3(std::min<u32>(div_up(m, 256), 2048u)), dim3(256), 0, stream, v, real_rows_dev.p, real_list.p, m, out);
			HIP_CHECK(hipGetLastError());
			HIP_CHECK(stream_wait(stream));
			size_t start = 0;
			if (max_cells > 0 && size_t(max_cells) < size_t(m)) start = size_t(m) - size_t(max_cells);
			filtered.resize(m - start); filtered_ridx.resize(m - start); filtered_umis.resize(m - start);
			parallel_ranges(m - start, [&](size_t b, size_t e, unsigned) {
				for (size_t i = b; i < e; ++i) { filtered[i] = out[start + i]; filtered_ridx[i] = out[size_t(m) + start + i]; filtered_umis[i] = int32_t(out[2 * size_t(m) + start + i]); }
			});
			filtered_valid = true;
			return;
		}
	}
	if (R >= device_min) {
		constexpr unsigned W = dropest::HostPool::MAX;
		auto st1 = std::make_unique<HostStage>(this, "sort_filtered:scan");
		size_t count[W] = {0};
		u64 any[W] = {0}; int bl[W]; bool uniform[W];
		for (unsigned w = 0; w < W; ++w) { bl[w] = -1; uniform[w] = true; }
		const unsigned workers = parallel_ranges(R, [&](size_t b, size_t e, unsigned w) {
			size_t c = 0; u64 a = 0; int l0 = -1; bool uni = true;   // locals: the per-worker slots share cache lines
			for (size_t i = b; i < e; ++i) {
				const HostCell &h = real[i];
				if (!passes(h)) continue;
				++c;
				a |= h.row.barcode;
				const int l = bit_length(h.row.barcode);
				if (l0 < 0) l0 = l; else if (l != l0) uni = false;
			}
			count[w] = c; any[w] = a; bl[w] = l0; uniform[w] = uni;
		}, 100000, W);
		size_t m64 = 0; u64 any_all = 0; int bl_all = -1; bool device_sort = true;
		size_t offset[W + 1];
		for (unsigned w = 0; w < workers; ++w) {
			offset[w] = m64; m64 += count[w]; any_all |= any[w];
			if (count[w]) { if (bl_all < 0) bl_all = bl[w]; device_sort &= uniform[w] && bl[w] == bl_all; }
		}
		if (m64 >= device_min && m64 < 0x50000000ull && device_sort && !(any_all & ESCAPE_BIT)) {
			const u32 m = u32(m64);
			st1.reset();
			auto st2 = std::make_unique<HostStage>(this, "sort_filtered:fill");
			sort_stage.ensure(size_t(m) * 3);
			sort_cols.ensure(size_t(m) * 3);
			sort_idx.resize(m); sort_ids.resize(m);   // (ids in a dense array: the gather below would otherwise miss the cache once per cell in `real`)
			keys_a.ensure(m); keys_b.ensure(m); vals_a.ensure(m); vals_b.ensure(m);
			u64 *h_code = sort_stage.p, *h_umis = sort_stage.p + m, *h_sizes = sort_stage.p + 2 * size_t(m);
			u64 o[W][3], a[W][3];
			parallel_ranges(R, [&](size_t b, size_t e, unsigned w) {
				u64 lo[3] = {0, 0, 0}, la[3] = {~0ull, ~0ull, ~0ull};
				size_t at = offset[w];
				for (size_t i = b; i < e; ++i) {
					const HostCell &h = real[i];
					if (!passes(h)) continue;
					const u64 sizes = (u64(h.row.requested_genes) << 32) | h.row.requested_umis;
					const u64 umis = u64(size_t(h.row.total_umis));   // Cell::umis_number casts the int stat to size_t
					h_code[at] = h.row.barcode; h_umis[at] = umis; h_sizes[at] = sizes; sort_idx[at] = u32(i); sort_ids[at] = h.id;
					lo[0] |= h.row.barcode; la[0] &= h.row.barcode; lo[1] |= umis; la[1] &= umis; lo[2] |= sizes; la[2] &= sizes;
					++at;
				}
				for (int c = 0; c < 3; ++c) { o[w][c] = lo[c]; a[w][c] = la[c]; }
			}, 100000, W);   // same n and limits as above: the same ranges
			u64 vary[3];
			for (int c = 0; c < 3; ++c) {
				u64 oo = 0, aa = ~0ull;
				for (unsigned w = 0; w < workers; ++w) if (count[w]) { oo |= o[w][c]; aa &= a[w][c]; }
				vary[c] = oo ^ aa;                                      // bits that vary: constant digits are skipped
			}
			st2.reset();
			auto st3 = std::make_unique<HostStage>(this, "sort_filtered:device");
			{ HostStage sx(this, "sort_filtered:device:h2d");
			// read from the pinned buffer by a kernel, not by the copy engine (busy with cm_raw's prefetch for tens of milliseconds at C3 size)
			hipLaunchKernelGGL(load_u64_from_host_kernel, dim3(std::min<u32>(div_up(m * 3u, 256), 4096u)), dim3(256), 0, stream, sort_stage.p, m * 3u, sort_cols.p);
			HIP_CHECK(hipGetLastError()); }
			auto sx2 = std::make_unique<HostStage>(this, "sort_filtered:device:sorts");
			const u64 *d_code = sort_cols.p, *d_umis = sort_cols.p + m, *d_sizes = sort_cols.p + 2 * size_t(m);
			u64 *k = keys_a.p, *k_alt = keys_b.p;
			u32 *v = vals_a.p, *v_alt = vals_b.p;
			hipLaunchKernelGGL(iota_kernel, dim3(div_up(m, 256)), dim3(256), 0, stream, v, m);
			HIP_CHECK(hipMemcpyAsync(k, d_code, size_t(m) * 8, hipMemcpyDeviceToDevice, stream));
			radix_sort(k, v, k_alt, v_alt, m, vary[0]);
			const std::pair<const u64 *, u64> more[2] = {{d_umis, vary[1]}, {d_sizes, vary[2]}};
			for (auto const &nx : more) {
				hipLaunchKernelGGL(gather_u64_kernel, dim3(div_up(m, 256)), dim3(256), 0, stream, nx.first, v, m, k);
				HIP_CHECK(hipGetLastError());
				radix_sort(k, v, k_alt, v_alt, m, nx.second);
			}
			sx2.reset();
			u32 *perm = reinterpret_cast<u32 *>(sort_stage.p);              // the staging buffer is free again (stream order)
			HostStage sx3(this, "sort_filtered:device:d2h");
			// written by a kernel, not by the copy engine: cm_raw's prefetch may hold that for tens of milliseconds
			hipLaunchKernelGGL(store_u32_to_host_kernel, dim3(std::min<u32>(div_up(m, 256), 2048u)), dim3(256), 0, stream, v, m, perm);
			HIP_CHECK(hipGetLastError());
			HIP_CHECK(stream_wait(stream));
			st3.reset();
			HostStage st4(this, "sort_filtered:gather");
			size_t start = 0;
			if (max_cells > 0 && size_t(max_cells) < size_t(m)) start = size_t(m) - size_t(max_cells);
			filtered.resize(m - start); filtered_ridx.resize(m - start); filtered_umis.resize(m - start);
			parallel_ranges(m - start, [&](size_t b, size_t e, unsigned) {   // (h_umis: the second key column, still in the staging buffer behind the permutation)
				for (size_t i = b; i < e; ++i) { const u32 at = perm[start + i]; filtered[i] = sort_ids[at]; filtered_ridx[i] = sort_idx[at]; filtered_umis[i] = int32_t(h_umis[at]); }
			});
			filtered_valid = true;
			return;
		}
	}

	struct Key { u64 sizes; u64 umis; u64 code; u32 idx; };
	std::vector<Key> keys;
	auto key_of = [&](u32 i) {
		const HostCell &h = real[i];
		return Key{(u64(h.row.requested_genes) << 32) | h.row.requested_umis,
		           u64(size_t(h.row.total_umis)),   // Cell::umis_number casts the int stat to size_t
		           h.row.barcode, i};
	};
	if (R >= 200000) {   // millions of real-candidate cells (C3: 2.5 M), a few 10^4 of them filtered: counted and collected on worker threads
		constexpr unsigned W = dropest::HostPool::MAX;
		size_t count[W] = {0}, offset[W + 1] = {0};
		const unsigned workers = parallel_ranges(R, [&](size_t b, size_t e, unsigned w) {
			size_t c = 0;
			for (size_t i = b; i < e; ++i) c += passes(real[i]) ? 1 : 0;
			count[w] = c;
		}, 100000, W);
		for (unsigned w = 0; w < workers; ++w) offset[w + 1] = offset[w] + count[w];
		keys.resize(offset[workers]);
		parallel_ranges(R, [&](size_t b, size_t e, unsigned w) {
			size_t at = offset[w];
			for (size_t i = b; i < e; ++i) if (passes(real[i])) keys[at++] = key_of(u32(i));
		}, 100000, W);   // same n and limits: the same ranges
	} else {
		keys.reserve(8192);
		for (u32 i = 0; i < real.size(); ++i) if (passes(real[i])) keys.push_back(key_of(i));
	}
	auto less = [&](const Key &a, const Key &b) {
		if (a.sizes != b.sizes) return a.sizes < b.sizes;
		if (a.umis != b.umis) return a.umis < b.umis;
		// barcode strings: clean codes of equal length order like their strings; anything else is decoded
		const bool plain = !((a.code | b.code) & ESCAPE_BIT) && bit_length(a.code) == bit_length(b.code);
		if (plain) return a.code < b.code;
		return barcode_of(real[a.idx]) < barcode_of(real[b.idx]);
	};
	// The sizes decide almost every comparison: a stable LSD counting sort on their varying bytes (three or four passes over a few
	// thousand keys), then only the runs of equal sizes by the full comparison.  std::sort on the 32-byte keys was 0.25 ms for the
	// 5 001 filtered cells of C2 -- host time between cm_raw's copy and cm's emit, with the PCIe link waiting at the end of the pass.
	if (keys.size() > 64) {
		u64 o = 0, a = ~0ull;
		for (const Key &k : keys) { o |= k.sizes; a &= k.sizes; }
		const u64 vary = o ^ a;
		std::vector<Key> other(keys.size());
		for (int shift = 0; shift < 64; shift += 8) {
			if (!((vary >> shift) & 0xFFull)) continue;
			size_t cnt[257] = {0};
			for (const Key &k : keys) ++cnt[((k.sizes >> shift) & 0xFFull) + 1];
			for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
			for (const Key &k : keys) other[cnt[(k.sizes >> shift) & 0xFFull]++] = k;
			keys.swap(other);
		}
		for (size_t i = 0; i < keys.size();) {
			size_t j = i + 1;
			while (j < keys.size() && keys[j].sizes == keys[i].sizes) ++j;
			if (j - i > 1) std::sort(keys.begin() + long(i), keys.begin() + long(j), less);
			i = j;
		}
	} else
		std::sort(keys.begin(), keys.end(), less);
	filtered.clear(); filtered_ridx.clear();
	size_t start = 0;
	if (max_cells > 0 && size_t(max_cells) < keys.size()) start = keys.size() - size_t(max_cells);
	filtered_umis.clear();
	for (size_t i = start; i < keys.size(); ++i) { filtered.push_back(real[keys[i].idx].id); filtered_ridx.push_back(keys[i].idx); filtered_umis.push_back(int32_t(keys[i].umis)); }
	filtered_valid = true;
}

#include "quality.h"
#include "merge_host.h"
#include "merge_shard.h"
#include "umi_merge_host.h"
#include "umi_directional_host.h"
#include "poisson_merge.h"
#include "simple_merge.h"
#include "merge_all.h"
#include "mutate_host.h"

// ------------------------------------------------------------------------------------------------
// top-level stages
// ------------------------------------------------------------------------------------------------
// First half of set_initialized: barcode table, cell ids and the statistics that size the sort key.  A sharded run
// stops here, all-reduces the statistics (dropest_ingest_summary_get / _set) and goes on with set_initialized.
void dropest_ctx::run_ingest() {
	if (initialized) throw InvalidError("Container is already initialized");
	if (ingested) return;
	HostStage hs_all(this, "ingest");
	concat_chunks();
	ingest = IngestStats{};
	ingest.umi_clean_min = ~0ull;
	forced_table_capacity = 0;
	if (n_reads > 0) {
		{ HostStage hs(this, "cb_table"); build_cb_table(); }
		{ HostStage hs(this, "cell_ids"); assign_cell_ids(); }
	}
	ingested = true;
}

void dropest_ctx::run_set_initialized() {
	if (initialized) throw InvalidError("Container is already initialized");
	// one context, the whole pass in one call, a stream long enough for the sampled table: the statistics move out of cb_insert
	// (a sharded run agrees on them between the two halves and keeps the exact ones of cb_insert)
	{
		uint64_t sample_min = uint64_t(1) << 22;
		if (const char *e = getenv("DROPEST_CB_SAMPLE_MIN")) sample_min = uint64_t(std::max(1ll, atoll(e)));
		lazy_stats = !ingested && n_reads >= sample_min && !getenv("DROPEST_EXACT_INGEST_STATS");
	}
	run_ingest();
	HostStage hs_all(this, "set_initialized");
	if (n_reads > 0) {
		{
			HostStage hs(this, "keys");
			plan_key_layout();
			if (lazy_stats) {
				// the plan came from every 256th read: build the keys with it, gather the exact statistics on the way, and keep the
				// keys if the exact plan is the same one (a different one -- a field a bit wider, a gene on two chromosomes the
				// sample did not see, an escaped UMI it missed -- costs one more key pass, never a wrong key)
				const KeyLayout planned = layout;
				const bool planned_chr = chr_from_gene, planned_strip = umi_sentinel_stripped;
				const int planned_clean = umi_clean_bits;
				build_keys(true);
				lazy_stats = false;
				plan_key_layout();
				const bool same = planned.umi_bits == layout.umi_bits && planned.gene_bits == layout.gene_bits && planned.cell_bits == layout.cell_bits &&
				                  planned.mark_shift == layout.mark_shift && planned.val_bytes == layout.val_bytes &&
				                  planned.umi_strip_mask == layout.umi_strip_mask && planned.umi_escape_base == layout.umi_escape_base &&
				                  planned.gene_none == layout.gene_none;
				if (!same || planned_chr != chr_from_gene || planned_strip != umi_sentinel_stripped || planned_clean != umi_clean_bits) {
					if (profiling) stats["count:key_plan_redone"].launches += 1;
					build_keys(false);
				}
			} else build_keys();
		}
		{ HostStage hs(this, "sort+reduce"); reduce_all(); }
		accumulate_umi_qualities();
		{ HostStage hs(this, "real_cells"); fetch_real_cells(true); }
	} else {
		// no reads (a shard that owns no barcode): the key fields are still laid out from the statistics the shards agreed on
		// (or from nothing), so that the tables the later stages size from the layout -- the UMI first-occurrence table of -u
		// that every shard all-gathers -- have the same shape on every shard
		plan_key_layout();
	}
	request_filtered(0, -1);   // update_cell_sizes(query, 0, -1), CellsDataContainer.cpp:168
	initialized = true;
	// cm_raw announced (dropest_set_raw_matrix_prefetch) and nothing in merge_and_filter can change it: its emit and its copy to the
	// host start now and run under the rest of the pass (the host work of ordering the filtered cells, the emit of cm)
	if (auto_pf_form >= 0 && n_reads && merge_phase_changes_nothing() && (auto_pf_form != 1 || narrow_possible())) prefetch_raw_matrix(auto_pf_reads, auto_pf_form);
	collect_timings();
}

// No CB merge, the Simple UMI merge and no UMI with N anywhere: merge_and_filter only re-filters the cells.
bool dropest_ctx::merge_phase_changes_nothing() const {
	return cfg.merge_kind == DROPEST_MERGE_NONE && cfg.umi_merge_kind != DROPEST_UMI_MERGE_DIRECTIONAL && ingest.umi_escape_max_plus1 == 0 && !hooks &&
	       !external_merge_done;
}

void dropest_ctx::run_merge_and_filter() {
	if (!initialized) throw InvalidError("You must initialize container");
	if (merged) throw InvalidError("merge_and_filter was already run");
	if (!merge_phase_changes_nothing()) invalidate_prefetch();   // (whatever rewrites the tables discards a prefetch on its own way in, too)
	HostStage hs(this, "merge_and_filter");
	tail_mark("merge_and_filter entered");
	if (cfg.merge_kind == DROPEST_MERGE_REAL_BARCODES && n_cells && !external_merge_done) run_cb_merge_real();
	if (cfg.merge_kind == DROPEST_MERGE_POISSON_REAL && n_cells && !external_merge_done) run_cb_merge_real();   // same loop, Poisson decisions
	if ((cfg.merge_kind == DROPEST_MERGE_SIMPLE || cfg.merge_kind == DROPEST_MERGE_POISSON_SIMPLE) && n_cells && !external_merge_done) run_cb_merge_simple();
	if (cfg.merge_kind == DROPEST_MERGE_ALL && n_cells && !external_merge_done) run_cb_merge_all();
	// MergeUMIsStrategy*::merge, after the CB merge (CellsDataContainer.cpp:45)
	if (cfg.umi_merge_kind == DROPEST_UMI_MERGE_DIRECTIONAL) run_umi_merge_directional(); else run_umi_merge_simple();
	request_filtered(min_after, cfg.max_cells);   // CellsDataContainer.cpp:47-49
	merged = true;
	if (auto_pf_form >= 0 && n_reads && !raw_pf.valid && (auto_pf_form != 1 || narrow_possible())) prefetch_raw_matrix(auto_pf_reads, auto_pf_form);
	collect_timings();
}

// ------------------------------------------------------------------------------------------------
// count matrices
// ------------------------------------------------------------------------------------------------
void dropest_ctx::matrix_columns(bool filtered_m, std::vector<u32> &col_cell, std::vector<u32> &colptr, uint64_t &nnz) {
	col_cell.clear(); colptr.clear();
	nnz = 0;
	if (filtered_m) {
		filtered_cells();
		for (u32 ri : filtered_ridx) {
			const HostCell &h = real[ri];
			col_cell.push_back(h.id); colptr.push_back(u32(nnz)); nnz += h.row.requested_genes;
		}
	} else {
		// every real cell in cell-id order: millions of rows at C3 size -- counted and filled over contiguous ranges on a few threads
		constexpr unsigned W = dropest::HostPool::MAX;
		size_t cols[W] = {0}; uint64_t sums[W] = {0};
		auto is_col = [&](const HostCell &h) { return !(h.merged || h.excluded || h.row.n_genes < min_before); };
		const unsigned workers = parallel_ranges(real.size(), [&](size_t b, size_t e, unsigned w) {
			size_t c = 0; uint64_t s = 0;
			for (size_t i = b; i < e; ++i) if (is_col(real[i])) { ++c; s += real[i].row.n_genes; }
			cols[w] = c; sums[w] = s;
		}, 100000, W);
		size_t col0[W + 1] = {0}; uint64_t nnz0[W + 1] = {0};
		for (unsigned w = 0; w < workers; ++w) { col0[w + 1] = col0[w] + cols[w]; nnz0[w + 1] = nnz0[w] + sums[w]; }
		nnz = nnz0[workers];
		if (nnz > 0xFFFFFFF0ull) throw UnsupportedError("count matrix with more than 2^32 non-zeros");
		col_cell.resize(col0[workers]); colptr.resize(col0[workers]);
		parallel_ranges(real.size(), [&](size_t b, size_t e, unsigned w) {   // same n and limits: the same ranges
			size_t at = col0[w]; uint64_t run = nnz0[w];
			for (size_t i = b; i < e; ++i) if (is_col(real[i])) { col_cell[at] = real[i].id; colptr[at] = u32(run); run += real[i].row.n_genes; ++at; }
		}, 100000, W);
	}
	if (nnz > 0xFFFFFFF0ull) throw UnsupportedError("count matrix with more than 2^32 non-zeros");
	colptr.push_back(u32(nnz));
}

void dropest_ctx::invalidate_prefetch() {
	if (raw_pf.in_flight && stream2) HIP_CHECK(stream_wait(stream2));   // its buffers are about to be reused
	mat[1].settle();   // ... also by the host threads that widen them
	raw_pf.valid = raw_pf.in_flight = false;
}

// Narrow CSC (16-bit row indices and values + an exact overflow list) is possible when every gene id fits 16 bits.
bool dropest_ctx::narrow_possible() const { return n_reads == 0 || ingest.gene_max_plus1 <= 0x10000u; }

static constexpr u32 MATRIX_OVF_CAP = 1u << 20;   // value lists (a count beyond 254 / 65534 is rare in every matrix)
// Row lists of the byte form.  A sparse column (a small cell of cm_raw: a few dozen of 30 000 genes) lists most of its rows; an eighth of
// all entries listed costs as much again as the bytes themselves, and beyond that a wider form is the better wire.
static u32 matrix_row_list_cap(uint64_t nnz) {
	if (const char *e = getenv("DROPEST_MATRIX_ROW_LIST_CAP")) return u32(std::max(1L, atol(e)));   // (tests: a small list overflows on a small matrix)
	return u32(std::min<uint64_t>(std::max<uint64_t>(nnz / 8, MATRIX_OVF_CAP), 1ull << 25));
}

// Wires the output side of an emit launch for matrix slot M (device form 0 / 1 / 2) and makes sure the buffers exist.
void dropest_ctx::matrix_outputs(MatrixResult &M, uint64_t nnz, int form, bool to_host, dropest::MatrixArgs &a) {
	M.settle();   // (a decoding thread may still be leaving the slot's previous job: its buffers are about to be rewritten or regrown)
	M.n_ovf = M.n_rovf = 0;
	if (form) {
		M.vcap = MATRIX_OVF_CAP;
		M.d_ovf.ensure(1 + 2 * size_t(M.vcap));
		if (to_host) M.h_ovf.ensure(1 + 2 * size_t(M.vcap));
		a.ovf_count = M.d_ovf.p; a.ovf_pos = M.d_ovf.p + 1; a.ovf_val = M.d_ovf.p + 1 + M.vcap; a.ovf_cap = M.vcap;
	}
	if (form == 2) {
		M.rcap = matrix_row_list_cap(nnz);
		M.d_drow8.ensure(nnz); M.d_val8.ensure(nnz); M.d_rovf.ensure(1 + 2 * size_t(M.rcap));
		if (to_host) { M.h_drow8.ensure(nnz); M.h_val8.ensure(nnz); M.h_rovf.ensure(1 + 2 * size_t(M.rcap)); }
		a.t_drow8 = M.d_drow8.p; a.t_val8 = M.d_val8.p;
		a.rovf_count = M.d_rovf.p; a.rovf_pos = M.d_rovf.p + 1; a.rovf_row = M.d_rovf.p + 1 + M.rcap; a.rovf_cap = M.rcap;
	} else if (form == 1) {
		M.d_row16.ensure(nnz); M.d_val16.ensure(nnz);
		if (to_host) { M.h_row16.ensure(nnz); M.h_val16.ensure(nnz); }
		a.t_gene16 = M.d_row16.p; a.t_val16 = M.d_val16.p;
	} else {
		M.d_row.ensure(nnz); M.d_val.ensure(nnz);
		if (to_host) { M.h_row.ensure(nnz); M.h_val.ensure(nnz); }
		a.t_gene = M.d_row.p; a.t_val = M.d_val.p;
	}
}

// Device-to-host copies of a matrix slot on stream `st` (after its emit launch).  Forms 1 / 2: the counts of the overflow lists travel
// with the arrays (listed entries are rare); the lists themselves are fetched by matrix_finish_overflow.
void dropest_ctx::matrix_copy_out(MatrixResult &M, uint64_t nnz, hipStream_t st) {
	if (M.narrow == 2) {
		HIP_CHECK(hipMemcpyAsync(M.h_drow8.p, M.d_drow8.p, nnz, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(M.h_val8.p, M.d_val8.p, nnz, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(M.h_ovf.p, M.d_ovf.p, 4, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(M.h_rovf.p, M.d_rovf.p, 4, hipMemcpyDeviceToHost, st));
	} else if (M.narrow == 1) {
		HIP_CHECK(hipMemcpyAsync(M.h_row16.p, M.d_row16.p, nnz * 2, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(M.h_val16.p, M.d_val16.p, nnz * 2, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(M.h_ovf.p, M.d_ovf.p, 4, hipMemcpyDeviceToHost, st));
	} else {
		HIP_CHECK(hipMemcpyAsync(M.h_row.p, M.d_row.p, nnz * 4, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(M.h_val.p, M.d_val.p, nnz * 4, hipMemcpyDeviceToHost, st));
	}
}

// After the copies of matrix_copy_out have been waited for: the overflow lists of a form 1 / 2 matrix (short or empty).
void dropest_ctx::matrix_finish_overflow(MatrixResult &M, hipStream_t st) {
	M.n_ovf = M.n_rovf = 0;
	if (!M.narrow || !M.nnz) return;
	auto finish = [&](dropest::DevBuf<u32> &d, dropest::PinnedBuf<u32> &h, u32 cap, u32 &n_out, const char *what) {
		const u32 count = h.p[0];
		if (count > cap) throw UnsupportedError(std::string("more than ") + std::to_string(cap) + " matrix entries with " + what + ": take the 32-bit form of the count matrix (dropest_count_matrix_csc)");
		n_out = count;
		if (!count) return;
		HIP_CHECK(hipMemcpyAsync(h.p + 1, d.p + 1, size_t(count) * 4, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(h.p + 1 + cap, d.p + 1 + cap, size_t(count) * 4, hipMemcpyDeviceToHost, st));
		HIP_CHECK(stream_wait(st));
		// The list is filled in the order the atomics landed.  The 16-bit form promises it sorted by position (a handful of entries).  The
		// byte form does not: the small cells of cm_raw list 4e5 rows at C2 (25 ms of std::sort here, 0.9 ms of radix passes and their
		// waits on the device), and its decoder needs no order (dropest_matrix_bytes_widen).
		if (M.narrow == 2) return;
		std::vector<std::pair<u32, u32>> ov(count);
		for (u32 i = 0; i < count; ++i) ov[i] = {h.p[1 + i], h.p[1 + cap + i]};
		std::sort(ov.begin(), ov.end());
		for (u32 i = 0; i < count; ++i) { h.p[1 + i] = ov[i].first; h.p[1 + cap + i] = ov[i].second; }
	};
	finish(M.d_ovf, M.h_ovf, M.vcap, M.n_ovf, M.narrow == 2 ? "a count beyond 254" : "a count beyond 65534");
	if (M.narrow == 2) finish(M.d_rovf, M.h_rovf, M.rcap, M.n_rovf, "a row gap beyond 254");
}

// ---- 32-bit slots that cross PCIe as bytes (matrix_decode.h) ----
// dropest_count_matrix_csc hands out the dgCMatrix slots i / x as 32-bit arrays (ResultsPrinter.cpp:433-442).  As such they are 8 bytes
// per entry on a link of ~50 GB/s -- 6 ms for the 3.8e7 entries of C2, the longest single piece of a 10 ms pass.  So a large matrix is
// emitted in the byte form (2 bytes per entry), copied in chunks of whole columns with an event behind each, and widened into the slots by
// host threads while the next chunk is on the link.  Lists longer than their capacity (a matrix of very sparse columns): the slots are emitted
// directly instead, as before (wire_finish returns false).  DROPEST_MATRIX_DIRECT=1 switches the detour off.
bool dropest_ctx::wire_wanted(uint64_t nnz, int form, bool to_host) const {
	static const bool off = getenv("DROPEST_MATRIX_DIRECT") != nullptr;
	return form == 0 && to_host && !off && matrix_wire && nnz >= (1u << 18);
}

void dropest_ctx::wire_copy_and_decode(MatrixResult &M, uint64_t nnz, hipStream_t st, const WireTarget *target) {
	using namespace dropest;
	if (!target) { M.h_row.ensure(nnz); M.h_val.ensure(nnz); }
	if (target) hipLaunchKernelGGL(matrix_lists_out_global_kernel, dim3(64), dim3(256), 0, st, M.d_rovf.p, M.rcap, M.h_rovf.p, M.d_ovf.p, M.vcap, M.h_ovf.p, target->d_descr, u32(M.ncols));
	else hipLaunchKernelGGL(matrix_lists_out_kernel, dim3(64), dim3(256), 0, st, M.d_rovf.p, M.rcap, M.h_rovf.p, M.d_ovf.p, M.vcap, M.h_ovf.p);
	HIP_CHECK(hipGetLastError());
	auto job = std::make_shared<DecodeJob>();
	HIP_CHECK(hipGetDevice(&job->device));
	job->m.rd = M.h_drow8.p; job->m.vb = M.h_val8.p; job->m.colptr = M.colptr.data(); job->m.ncols = M.ncols; job->m.nnz = nnz;
	job->ro = M.h_row.p; job->vo = M.h_val.p;
	if (target) {   // a shard's columns: bytes by the local colptr, slots at each column's global place
		job->m.colptr = target->begin; job->m.colend = target->end; job->m.bytebeg = M.colptr.data(); job->m.nnz = target->global_nnz;
		job->cut = M.colptr.data();
		job->ro = target->rows; job->vo = target->vals;
	}
	job->r_count = M.h_rovf.p; job->r_pos = M.h_rovf.p + 1; job->r_val = M.h_rovf.p + 1 + M.rcap; job->rcap = M.rcap;
	job->v_count = M.h_ovf.p; job->v_pos = M.h_ovf.p + 1; job->v_val = M.h_ovf.p + 1 + M.vcap; job->vcap = M.vcap;
	static const uint64_t n_chunks = [] { const char *e = getenv("DROPEST_WIRE_CHUNKS"); return uint64_t(e ? std::max(1, atoi(e)) : 12); }();
	cut_columns(M.colptr.data(), 0, size_t(M.ncols), std::max<uint64_t>(nnz / n_chunks + 1, uint64_t(1) << 19), job->chunk_end);
	M.wire_chunk_end = job->chunk_end;   // (a rider on this matrix moves its bytes in the same chunks: emit_rider)
	// Arrival flags (matrix_decode.h): [0] the lists, [1 + j] chunk j; the value of this emit is a number no earlier emit of the slot used.
	const size_t K = job->chunk_end.size();
	M.h_flags.ensure(K + 2);
	for (size_t j = 0; j < K + 2; ++j) M.h_flags.p[j] = 0;   // (the slot's previous job is complete: nobody reads or writes these now)
	M.wire_epoch = M.wire_epoch + 1 ? M.wire_epoch + 1 : 1;
	job->flags = M.h_flags.p; job->epoch = M.wire_epoch;
	// Every chunk leaves by a kernel that writes the pinned buffers itself (k_misc.h: a device-to-host hipMemcpyAsync costs ~20 us of
	// copy-engine set-up each, two dozen of them per matrix) and starts by raising the flag of what came before it on the stream.
	u32 c0 = 0;
	for (size_t j = 0; j < K; ++j) {
		const u32 c1 = job->chunk_end[j];
		const size_t k0 = M.colptr[c0], k1 = M.colptr[c1];
		hipLaunchKernelGGL(matrix_chunk_to_host_kernel, dim3(u32(std::max<size_t>(1, std::min<size_t>(128, (k1 - k0 + 4095) / 4096)))), dim3(256), 0, st, M.d_drow8.p, M.d_val8.p,
		                   M.h_drow8.p, M.h_val8.p, k0, k1, M.h_flags.p + j, M.wire_epoch);
		c0 = c1;
	}
	hipLaunchKernelGGL(matrix_flag_kernel, dim3(1), dim3(1), 0, st, M.h_flags.p + K, M.wire_epoch);
	HIP_CHECK(hipGetLastError());
	static const bool trace = getenv("DROPEST_WIRE_TRACE") != nullptr;
	job->trace = trace;
	static const uint64_t slice_entries = [] { const char *e = getenv("DROPEST_DECODE_SLICE"); return e && atoll(e) >= 1024 ? uint64_t(atoll(e)) : uint64_t(1) << 16; }();
	static const uint32_t test_delay = [] { const char *e = getenv("DROPEST_DECODE_TEST_DELAY_US"); return e ? uint32_t(std::max(0, atoi(e))) : 0u; }();
	job->test_delay_us = test_delay;
	job->prepare(slice_entries);
	M.job = job; M.wire = true;
	M.job_t0 = std::chrono::steady_clock::now();
	DecodePool::get().prefer_node_of(target ? target->rows : M.h_row.p);
	DecodePool::get().submit(job);
}

// Sharded runs: the columns of a caller-given list of cells (local offsets col_start, nnz entries in all) leave the device as the byte form
// of a LOCAL matrix -- the emit of one context, the same chunked copies at the link's streaming rate -- and the pool's host threads widen
// them into the caller's slots at each column's GLOBAL place (the target).  The caller ends with wire_finish(mat[...]); false = the lists
// overflowed (very sparse columns): it then places the 32-bit form itself.
void dropest_ctx::ship_columns_to_slots(bool filtered_m, bool reads_output, const std::vector<u32> &col_cell, const std::vector<u32> &col_start, uint64_t nnz,
                                        const WireTarget &target, hipStream_t copy_st) {
	using namespace dropest;
	if (!filtered_m) invalidate_prefetch();   // (cm's columns use their own slot: cm_raw's may still be on their way, and stay so)
	MatrixResult &M = mat[filtered_m ? 0 : 1];
	M.settle();
	const u32 ncols = u32(col_cell.size());
	M.colptr.assign(col_start.begin(), col_start.end()); M.colptr.push_back(u32(nnz));
	M.nnz = nnz; M.ncols = ncols; M.narrow = 0; M.n_ovf = M.n_rovf = 0; M.wire = false;
	if (!ncols || !nnz) return;
	m_col_cell.ensure(ncols); m_col_start.ensure(ncols);
	DevBuf<u32> &d_cell = filtered_m ? m_col_cell : m2_col_cell, &d_start = filtered_m ? m_col_start : m2_col_start;
	d_cell.ensure(ncols); d_start.ensure(ncols);
	MatrixArgs a{};
	matrix_outputs(M, nnz, 2, true, a);
	HIP_CHECK(hipMemcpyAsync(d_cell.p, col_cell.data(), size_t(ncols) * 4, hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemcpyAsync(d_start.p, col_start.data(), size_t(ncols) * 4, hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemsetAsync(M.d_ovf.p, 0, 4, stream));
	HIP_CHECK(hipMemsetAsync(M.d_rovf.p, 0, 4, stream));
	a.col_cell = d_cell.p; a.col_start = d_start.p; a.cell_cg_begin = cell_cg_begin.p; a.cell_cg_count = cell_cg_count.p; a.cg_key = cg_key.p;
	a.value = filtered_m ? (reads_output ? cg_reads_req.p : cg_n_req.p) : (reads_output ? cg_reads_all.p : cg_n_all.p);
	a.gene_mask = layout.gene_none; a.skip_zero = filtered_m ? 1 : 0;
	timed(filtered_m ? "emit_matrix:cm" : "emit_matrix:cm_raw", double(nnz) * 14, [&] {
		std::vector<u32> rows(ncols);
		for (u32 j = 0; j < ncols; ++j) rows[j] = M.colptr[j + 1] - M.colptr[j];
		launch_emit_bytes(a, rows, filtered_m ? m_col_list : m2_col_list, filtered_m ? m_col_list_host : m2_col_list_host, stream);
	});
	if (copy_st && copy_st != stream) {
		if (!ev_ship) HIP_CHECK(hipEventCreateWithFlags(&ev_ship, hipEventDisableTiming));
		HIP_CHECK(hipEventRecord(ev_ship, stream));
		HIP_CHECK(hipStreamWaitEvent(copy_st, ev_ship, 0));
	} else copy_st = stream;
	wire_copy_and_decode(M, nnz, copy_st, &target);
}

bool dropest_ctx::wire_finish(MatrixResult &M) {
	using dropest::DecodeJob;
	if (!M.job) return true;
	HostStage hs(this, "matrix:decode_wait");
	const auto w0 = std::chrono::steady_clock::now();
	M.job->work(true);   // the caller takes part: what is unclaimed, then what a straggler holds
	const int st = M.job->finish_rider();   // (= wait(); a rider's listed values are put in place here)
	if (M.job->trace) {
		const auto now = std::chrono::steady_clock::now();
		{   // where the buffers live (NUMA node of a few pages each) and where this thread runs
			auto node_of = [](const void *p, size_t bytes) {
				void *pages[8]; int status[8] = {-9, -9, -9, -9, -9, -9, -9, -9};
				for (int i = 0; i < 8; ++i) pages[i] = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(p) + bytes / 8 * size_t(i)) & ~uintptr_t(4095));
				(void)syscall(SYS_move_pages, 0, 8ul, pages, nullptr, status, 0);
				std::string out;
				for (int i = 0; i < 8; ++i) out += std::to_string(status[i]) + (i < 7 ? "," : "");
				return out;
			};
			fprintf(stderr, "[wire] nodes: rows %s | values %s | delta bytes %s | caller on cpu %d\n", node_of(M.h_row.p, M.nnz * 4).c_str(),
			        node_of(M.h_val.p, M.nnz * 4).c_str(), node_of(M.h_drow8.p, M.nnz).c_str(), sched_getcpu());
		}
		fprintf(stderr, "[wire] nnz %llu: submit -> done %.3f ms, waited %.3f ms, slowest slice %.3f ms of %zu slices in %zu chunks\n", (unsigned long long)M.nnz,
		        std::chrono::duration<double, std::milli>(now - M.job_t0).count(), std::chrono::duration<double, std::milli>(now - w0).count(),
		        double(M.job->slowest_slice_ns.load()) * 1e-6, M.job->slice_end.size(), M.job->chunk_end.size());
	}
	M.n_rovf = M.job->n_r; M.n_ovf = M.job->n_v;
	M.late_job = std::move(M.job);   // complete; a straggler may still be inside (settle() before the buffers are touched again)
	M.job.reset();
	if (st == DecodeJob::DONE) return true;
	if (st == DecodeJob::OVERFLOW) return false;
	throw DeviceError(st == DecodeJob::FAILED ? "count matrix: a copy of the byte form failed" : "count matrix: a listed entry of the byte form lies outside the matrix");
}

// Byte form: long columns by the workgroup-per-column kernel, short ones (fewer than 256 (cell, gene) rows) by the wave-per-column kernel.
// `list` (device, >= ncols words; filled here) holds the long columns first, then the short ones.
void dropest_ctx::launch_emit_bytes(dropest::MatrixArgs a, const std::vector<u32> &col_cell_rows /* rows per column */, dropest::DevBuf<u32> &list,
                                    std::vector<u32> &host_list, hipStream_t st) {
	using namespace dropest;
	const u32 ncols = u32(col_cell_rows.size());
	host_list.resize(ncols);
	u32 n_long = 0;
	for (u32 j = 0; j < ncols; ++j) if (col_cell_rows[j] >= 256u) host_list[n_long++] = j;
	u32 at = n_long;
	for (u32 j = 0; j < ncols; ++j) if (col_cell_rows[j] < 256u) host_list[at++] = j;
	list.ensure(ncols);
	HIP_CHECK(hipMemcpyAsync(list.p, host_list.data(), size_t(ncols) * 4, hipMemcpyHostToDevice, st));
	if (n_long) { a.col_list = list.p; a.n_list = n_long; hipLaunchKernelGGL(emit_matrix_kernel<2>, dim3(n_long), dim3(256), 0, st, a); }
	if (ncols > n_long) {
		a.col_list = list.p + n_long; a.n_list = ncols - n_long;
		hipLaunchKernelGGL(emit_matrix_bytes_short_kernel, dim3(div_up(a.n_list, EMS_COLS)), dim3(256), 0, st, a);
	}
}

// cm_raw on a second stream: emit + device-to-host copy start now and run under whatever the caller does next (ordering
// the filtered cells, emitting cm); dropest_count_matrix_csc(filtered = 0) later only waits for the copy (and, for the 32-bit
// slots that travel as bytes, for the host threads that widen them).
void dropest_ctx::prefetch_raw_matrix(bool reads_output, int narrow, const dropest::CellRowPod *rows, const u32 *ids, u32 count) {
	if (raw_pf.valid && raw_pf.reads_output == reads_output && raw_pf.narrow == narrow) return;   // already under way (dropest_set_raw_matrix_prefetch)
	HostStage hs(this, "prefetch:cm_raw");
	invalidate_prefetch();
	if (narrow == 1 && !narrow_possible()) throw UnsupportedError("gene ids beyond 65535: the narrow matrix form is not available");
	MatrixResult &M = mat[1];
	uint64_t nnz = 0;
	if (rows) {   // straight from the rows fetch_real_cells just received (every one of them a column: n_genes >= min_genes_before_merge, nothing merged or excluded yet)
		raw_pf.col_cell.assign(ids, ids + count);
		M.colptr.resize(size_t(count) + 1);
		for (u32 i = 0; i < count; ++i) { M.colptr[i] = u32(nnz); nnz += rows[i].n_genes; }
		if (nnz > 0xFFFFFFF0ull) throw UnsupportedError("count matrix with more than 2^32 non-zeros");
		M.colptr[count] = u32(nnz);
	} else
		matrix_columns(false, raw_pf.col_cell, M.colptr, nnz);
	M.nnz = nnz; M.ncols = raw_pf.col_cell.size(); M.narrow = narrow; M.n_ovf = M.n_rovf = 0; M.wire = false;
	raw_pf.valid = true; raw_pf.reads_output = reads_output; raw_pf.narrow = narrow;
	if (nnz == 0) return;
	if (!stream2) {
		HIP_CHECK(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
		HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
		HIP_CHECK(hipEventCreateWithFlags(&ev_raw, hipEventDisableTiming));
	}
	const u32 ncols = u32(raw_pf.col_cell.size());
	m2_col_cell.ensure(ncols); m2_col_start.ensure(ncols);
	MatrixArgs a{};
	const bool wire = wire_wanted(nnz, narrow, true);
	const int dev_form = wire ? 2 : narrow;
	matrix_outputs(M, nnz, dev_form, true, a);
	HIP_CHECK(hipEventRecord(ev_fork, stream));                // everything enqueued so far (the tables) comes first
	HIP_CHECK(hipStreamWaitEvent(stream2, ev_fork, 0));
	HIP_CHECK(hipMemcpyAsync(m2_col_cell.p, raw_pf.col_cell.data(), size_t(ncols) * 4, hipMemcpyHostToDevice, stream2));
	HIP_CHECK(hipMemcpyAsync(m2_col_start.p, M.colptr.data(), size_t(ncols) * 4, hipMemcpyHostToDevice, stream2));
	if (!ev_raw_cols) HIP_CHECK(hipEventCreateWithFlags(&ev_raw_cols, hipEventDisableTiming));
	HIP_CHECK(hipEventRecord(ev_raw_cols, stream2));
	if (dev_form) HIP_CHECK(hipMemsetAsync(M.d_ovf.p, 0, 4, stream2));
	if (dev_form == 2) HIP_CHECK(hipMemsetAsync(M.d_rovf.p, 0, 4, stream2));
	a.col_cell = m2_col_cell.p; a.col_start = m2_col_start.p; a.cell_cg_begin = cell_cg_begin.p; a.cell_cg_count = cell_cg_count.p; a.cg_key = cg_key.p;
	a.value = reads_output ? cg_reads_all.p : cg_n_all.p;
	a.gene_mask = layout.gene_none; a.skip_zero = 0;
	if (dev_form == 2) {
		std::vector<u32> rows(ncols);   // (an upper bound of every column's entries: cm_raw keeps all of a cell's genes)
		for (u32 j = 0; j < ncols; ++j) rows[j] = M.colptr[j + 1] - M.colptr[j];
		launch_emit_bytes(a, rows, m2_col_list, m2_col_list_host, stream2);
	}
	else if (dev_form == 1) hipLaunchKernelGGL(emit_matrix_kernel<1>, dim3(ncols), dim3(256), 0, stream2, a);
	else hipLaunchKernelGGL(emit_matrix_kernel<0>, dim3(ncols), dim3(256), 0, stream2, a);
	HIP_CHECK(hipGetLastError());
	if (wire) wire_copy_and_decode(M, nnz, stream2); else matrix_copy_out(M, nnz, stream2);
	HIP_CHECK(hipEventRecord(ev_raw, stream2));
	raw_pf.in_flight = true;
}

void dropest_ctx::emit_matrix(bool filtered_m, bool reads_output, bool to_host, int narrow, bool direct) {
	HostStage hs(this, filtered_m ? "matrix:cm" : "matrix:cm_raw");
	if (narrow == 1 && !narrow_possible()) throw UnsupportedError("gene ids beyond 65535: the narrow matrix form is not available");
	MatrixResult &M = mat[filtered_m ? 0 : 1];
	std::vector<u32> col_cell;
	uint64_t nnz = 0;
	if (!direct && !filtered_m && raw_pf.valid && to_host && raw_pf.reads_output == reads_output && raw_pf.narrow == narrow) {
		// A prefetched cm_raw of the same value kind and form IS what this call would produce: whatever changes the container
		// (merges, mutators, a new pass) discards the prefetch on its way in (invalidate_prefetch), so a valid one is current.
		// (Round 2 rebuilt and compared the column lists here: two walks over 2.5 M cells, 10 ms of a C3 pass.)
		bool ok = true;
		if (raw_pf.in_flight) {
			HIP_CHECK(event_wait(ev_raw)); raw_pf.in_flight = false;
			if (M.wire) ok = wire_finish(M); else matrix_finish_overflow(M, stream2);
		}
		if (ok) return;
		direct = true;   // the lists of the byte form overflowed: the slots come directly
	}
	if (!filtered_m) invalidate_prefetch();
	M.settle();
	tail_mark(filtered_m ? "emit_matrix(cm) entered" : "emit_matrix(cm_raw) entered");
	matrix_columns(filtered_m, col_cell, M.colptr, nnz);
	tail_mark("columns known");
	M.nnz = nnz; M.ncols = col_cell.size(); M.narrow = narrow; M.n_ovf = M.n_rovf = 0; M.wire = false;
	if (nnz == 0) return;
	const u32 ncols = u32(col_cell.size());
	m_col_cell.ensure(ncols); m_col_start.ensure(ncols);
	MatrixArgs a{};
	const bool wire = !direct && wire_wanted(nnz, narrow, to_host);
	if (wire && filtered_m && emit_rider(reads_output, col_cell, nnz)) {   // cm's values ride on cm_raw's rows: half of cm's bytes stay off the link
		tail_mark("rider enqueued");
		HIP_CHECK(stream_wait(stream));
		const bool ok = wire_finish(M);
		tail_mark("matrix done");
		collect_timings();
		if (!ok) emit_matrix(filtered_m, reads_output, to_host, narrow, true);
		return;
	}
	const int dev_form = wire ? 2 : narrow;
	matrix_outputs(M, nnz, dev_form, to_host, a);
	HIP_CHECK(hipMemcpyAsync(m_col_cell.p, col_cell.data(), size_t(ncols) * 4, hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemcpyAsync(m_col_start.p, M.colptr.data(), size_t(ncols) * 4, hipMemcpyHostToDevice, stream));
	if (dev_form) HIP_CHECK(hipMemsetAsync(M.d_ovf.p, 0, 4, stream));
	if (dev_form == 2) HIP_CHECK(hipMemsetAsync(M.d_rovf.p, 0, 4, stream));
	a.col_cell = m_col_cell.p; a.col_start = m_col_start.p; a.cell_cg_begin = cell_cg_begin.p; a.cell_cg_count = cell_cg_count.p; a.cg_key = cg_key.p;
	a.value = filtered_m ? (reads_output ? cg_reads_req.p : cg_n_req.p) : (reads_output ? cg_reads_all.p : cg_n_all.p);
	a.gene_mask = layout.gene_none; a.skip_zero = filtered_m ? 1 : 0;
	timed(filtered_m ? "emit_matrix:cm" : "emit_matrix:cm_raw", double(nnz) * (dev_form == 2 ? 14 : dev_form ? 16 : 20), [&] {
		if (dev_form == 2) {
			std::vector<u32> rows(ncols);   // entries per column (cm: the requested genes; the kernel walks a few more rows and drops the zeros)
			for (u32 j = 0; j < ncols; ++j) rows[j] = M.colptr[j + 1] - M.colptr[j];
			launch_emit_bytes(a, rows, m_col_list, m_col_list_host, stream);
		}
		else if (dev_form == 1) hipLaunchKernelGGL(emit_matrix_kernel<1>, dim3(ncols), dim3(256), 0, stream, a);
		else hipLaunchKernelGGL(emit_matrix_kernel<0>, dim3(ncols), dim3(256), 0, stream, a);
	});
	tail_mark("emit enqueued");
	if (to_host) { if (wire) wire_copy_and_decode(M, nnz, stream); else matrix_copy_out(M, nnz, stream); }
	tail_mark("copies enqueued");
	HIP_CHECK(stream_wait(stream));   // col_cell (host vector) must outlive the H2D copy
	tail_mark("stream drained");
	bool ok = true;
	if (to_host) { if (wire) ok = wire_finish(M); else matrix_finish_overflow(M, stream); }
	tail_mark("matrix done");
	collect_timings();
	if (!ok) emit_matrix(filtered_m, reads_output, to_host, narrow, true);
}

// cm as a rider on cm_raw (k_misc.h: emit_values_on_rows_kernel, matrix_decode.h: widen_derived).  Taken when cm_raw of the
// same value kind is on its way (or there) in the byte form: one byte per entry of cm_raw -- its value in cm, 0 = not in cm -- crosses the link
// in cm_raw's chunks, and the host threads build cm's slots from cm_raw's row deltas.  mat[0].colptr / nnz are cm's own (the caller set them).
// false: not possible here (the caller emits cm's own byte form).
bool dropest_ctx::emit_rider(bool reads_output, const std::vector<u32> &col_cell, uint64_t nnz) {
	using namespace dropest;
	static const bool off = getenv("DROPEST_NO_RIDER") != nullptr;
	MatrixResult &M = mat[0], &R = mat[1];
	// Measured (NOTES_r06 section 5): at C3 (1.9e8 entries per matrix) the step is 0 to 6 ms shorter with the rider, box by box (0.18 of 0.69 GB off the link); at C2 (1.9e7
	// entries) the 0.35 ms of link time it saves are within the noise of its own dependencies (cm's columns wait for cm_raw's lists and
	// chunks): 9.53 against 9.32 ms over five pairs of runs.  Taken from 2^26 entries of cm_raw on; DROPEST_RIDER_MIN_NNZ moves the gate (tests: 0).
	const char *e_min = getenv("DROPEST_RIDER_MIN_NNZ");
	const uint64_t min_nnz = e_min ? uint64_t(std::max(0ll, atoll(e_min))) : (uint64_t(1) << 26);
	if (R.nnz < min_nnz) return false;
	if (off || !raw_pf.valid || !R.wire || R.narrow != 0 || raw_pf.reads_output != reads_output || !R.nnz || R.wire_chunk_end.empty() || !ev_raw_cols) return false;
	if (R.job == nullptr && R.late_job == nullptr) return false;   // (cm_raw's bytes have been let go)
	if (R.job && !R.job->running() && R.job->status.load() != DecodeJob::DONE) return false;   // cm_raw takes a wider form
	const u32 rcols = u32(R.ncols), ncols = u32(col_cell.size());
	if (raw_pf.col_cell.size() != rcols) return false;
	M.rider_out.assign(rcols, 0xFFFFFFFFu); M.rider_cnt.assign(rcols, 0u);
	for (u32 j = 0; j < ncols; ++j) {   // (cm_raw's columns ascend by cell id)
		auto it = std::lower_bound(raw_pf.col_cell.begin(), raw_pf.col_cell.end(), col_cell[j]);
		if (it == raw_pf.col_cell.end() || *it != col_cell[j]) return false;       // a filtered cell that is no column of cm_raw: not this way
		const size_t r = size_t(it - raw_pf.col_cell.begin());
		M.rider_out[r] = M.colptr[j]; M.rider_cnt[r] = M.colptr[j + 1] - M.colptr[j];
	}
	MatrixArgs a{};
	matrix_outputs(M, R.nnz, 2, true, a);          // (value bytes and lists sized for cm_raw's entries; settles the slot's previous job)
	M.h_row.ensure(nnz); M.h_val.ensure(nnz);
	M.d_rider_out.ensure(rcols);
	HIP_CHECK(hipMemcpyAsync(M.d_rider_out.p, M.rider_out.data(), size_t(rcols) * 4, hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemsetAsync(M.d_ovf.p, 0, 4, stream));
	HIP_CHECK(hipMemsetAsync(M.d_rovf.p, 0, 4, stream));
	HIP_CHECK(hipStreamWaitEvent(stream, ev_raw_cols, 0));   // cm_raw's column arrays (m2_col_cell / m2_col_start) are on the device
	timed("emit_matrix:cm", double(R.nnz) * 13, [&] {
		hipLaunchKernelGGL(emit_values_on_rows_kernel, dim3(std::min<u32>(div_up(rcols, 4u), 8192u)), dim3(256), 0, stream, m2_col_cell.p, m2_col_start.p, rcols, cell_cg_begin.p,
		                   cell_cg_count.p, cg_key.p, layout.gene_none, reads_output ? cg_reads_req.p : cg_n_req.p, M.d_rider_out.p, M.d_val8.p, a.ovf_count, a.ovf_pos, a.ovf_val, a.ovf_cap);
	});
	hipLaunchKernelGGL(matrix_lists_out_kernel, dim3(64), dim3(256), 0, stream, M.d_rovf.p, M.rcap, M.h_rovf.p, M.d_ovf.p, M.vcap, M.h_ovf.p);
	HIP_CHECK(hipGetLastError());
	auto job = std::make_shared<DecodeJob>();
	HIP_CHECK(hipGetDevice(&job->device));
	job->m.rd = R.h_drow8.p; job->m.vb = M.h_val8.p; job->m.colptr = R.colptr.data(); job->m.ncols = rcols; job->m.nnz = nnz;
	job->ro = M.h_row.p; job->vo = M.h_val.p;
	job->r_count = M.h_rovf.p; job->r_pos = M.h_rovf.p + 1; job->r_val = M.h_rovf.p + 1 + M.rcap; job->rcap = M.rcap;   // (always empty: a rider has no rows of its own)
	job->v_count = M.h_ovf.p; job->v_pos = M.h_ovf.p + 1; job->v_val = M.h_ovf.p + 1 + M.vcap; job->vcap = M.vcap;
	job->derived = true;
	job->dv.base_ro = R.h_row.p; job->dv.out_begin = M.rider_out.data(); job->dv.out_count = M.rider_cnt.data();
	if (R.job && R.job->running()) { job->base_job = R.job; job->flags2 = R.h_flags.p; job->epoch2 = R.wire_epoch; }
	job->chunk_end = R.wire_chunk_end;
	const size_t K = job->chunk_end.size();
	M.h_flags.ensure(K + 2);
	for (size_t j = 0; j < K + 2; ++j) M.h_flags.p[j] = 0;
	M.wire_epoch = M.wire_epoch + 1 ? M.wire_epoch + 1 : 1;
	job->flags = M.h_flags.p; job->epoch = M.wire_epoch;
	u32 c0 = 0;
	for (size_t j = 0; j < K; ++j) {
		const u32 c1 = job->chunk_end[j];
		const size_t k0 = R.colptr[c0], k1 = R.colptr[c1];
		hipLaunchKernelGGL(matrix_chunk_to_host_kernel, dim3(u32(std::max<size_t>(1, std::min<size_t>(128, (k1 - k0 + 4095) / 4096)))), dim3(256), 0, stream, M.d_val8.p,
		                   static_cast<const uint8_t *>(nullptr), M.h_val8.p, static_cast<uint8_t *>(nullptr), k0, k1, M.h_flags.p + j, M.wire_epoch);
		c0 = c1;
	}
	hipLaunchKernelGGL(matrix_flag_kernel, dim3(1), dim3(1), 0, stream, M.h_flags.p + K, M.wire_epoch);
	HIP_CHECK(hipGetLastError());
	static const bool trace = getenv("DROPEST_WIRE_TRACE") != nullptr;
	job->trace = trace;
	static const uint64_t slice_entries = [] { const char *e = getenv("DROPEST_DECODE_SLICE"); return e && atoll(e) >= 1024 ? uint64_t(atoll(e)) : uint64_t(1) << 16; }();
	job->prepare(slice_entries);
	M.narrow = 0; M.n_ovf = M.n_rovf = 0;
	M.job = job; M.wire = true;
	M.job_t0 = std::chrono::steady_clock::now();
	R.dependent = &M;
	DecodePool::get().prefer_node_of(M.h_row.p);
	DecodePool::get().submit(job);
	if (profiling) stats["count:cm_rides_on_cm_raw"].launches += 1;
	return true;
}

// Sharded runs: the columns of a caller-given list of cells, emitted compactly into the device staging of matrix slot
// (filtered: cm values and zero-skipping; else cm_raw) -- the caller places them in the global matrix.
void dropest_ctx::emit_columns_device(bool filtered_m, bool reads_output, const std::vector<u32> &col_cell, const std::vector<u32> &col_start, uint64_t nnz, bool wait) {
	invalidate_prefetch();
	MatrixResult &M = mat[filtered_m ? 0 : 1];
	const u32 ncols = u32(col_cell.size());
	if (!ncols || !nnz) return;
	m_col_cell.ensure(ncols); m_col_start.ensure(ncols);
	M.d_row.ensure(nnz); M.d_val.ensure(nnz); M.narrow = 0; M.n_ovf = 0;
	HIP_CHECK(hipMemcpyAsync(m_col_cell.p, col_cell.data(), size_t(ncols) * 4, hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemcpyAsync(m_col_start.p, col_start.data(), size_t(ncols) * 4, hipMemcpyHostToDevice, stream));
	MatrixArgs a{};
	a.col_cell = m_col_cell.p; a.col_start = m_col_start.p; a.cell_cg_begin = cell_cg_begin.p; a.cell_cg_count = cell_cg_count.p; a.cg_key = cg_key.p;
	a.value = filtered_m ? (reads_output ? cg_reads_req.p : cg_n_req.p) : (reads_output ? cg_reads_all.p : cg_n_all.p);
	a.gene_mask = layout.gene_none; a.skip_zero = filtered_m ? 1 : 0;
	a.t_gene = M.d_row.p; a.t_val = M.d_val.p;
	timed(filtered_m ? "emit_matrix:cm" : "emit_matrix:cm_raw", double(nnz) * 20, [&] {
		hipLaunchKernelGGL(emit_matrix_kernel<0>, dim3(ncols), dim3(256), 0, stream, a);
	});
	if (wait) HIP_CHECK(stream_wait(stream));   // the host vectors must outlive their copies (wait = false: the caller keeps them until the stream has drained)
}

// ResultsPrinter::get_count_matrix_filtered(container, query_marks) (ResultsPrinter.cpp:333-361) for a query other than
// the container's own: columns = the filtered cells, values = UMIs (reads) of each gene whose mark matches, zero
// entries dropped (Cell::requested_umis_per_gene, Cell.cpp:54-68).
void dropest_ctx::emit_matrix_levels(u32 mask, bool reads_output) {
	HostStage hs(this, "matrix:levels");
	MatrixResult &M = mat[2];
	M.colptr.assign(1, 0); M.nnz = 0; M.ncols = 0;
	filtered_cells();
	std::vector<u32> col_cell;
	for (u32 ri : filtered_ridx) col_cell.push_back(real[ri].id);
	const u32 ncols = u32(col_cell.size());
	M.ncols = ncols;
	M.colptr.assign(size_t(ncols) + 1, 0);
	if (!ncols || !n_cg) return;
	DevBuf<u32> d_value, d_count;
	d_value.alloc(n_cg); d_count.alloc(ncols);
	hipLaunchKernelGGL(cg_requested_by_mask_kernel, dim3(div_up(n_cg, 256)), dim3(256), 0, stream, cg_mol_begin.p, n_cg, mol_mark.p, mol_reads.p,
	                   mask, reads_output ? 1 : 0, d_value.p);
	HIP_CHECK(hipGetLastError());
	// groups rewritten by a UMI merge on the host: their molecules live in the override map
	for (auto const &kv : umi_overrides) {
		const u32 cell = u32(kv.first >> layout.gene_bits);
		u32 cgb = 0, cgc = 0;
		HIP_CHECK(hipMemcpy(&cgb, cell_cg_begin.p + cell, 4, hipMemcpyDeviceToHost));
		HIP_CHECK(hipMemcpy(&cgc, cell_cg_count.p + cell, 4, hipMemcpyDeviceToHost));
		std::vector<u64> keys(cgc);
		HIP_CHECK(hipMemcpy(keys.data(), cg_key.p + cgb, size_t(cgc) * 8, hipMemcpyDeviceToHost));
		for (u32 j = 0; j < cgc; ++j) {
			if (keys[j] != kv.first) continue;
			u32 v = 0;
			for (const UmiOverride &o : kv.second) if ((mask >> (o.mark & 7u)) & 1u) v += reads_output ? o.reads : 1u;
			HIP_CHECK(hipMemcpy(d_value.p + cgb + j, &v, 4, hipMemcpyHostToDevice));
		}
	}
	m_col_cell.ensure(ncols); m_col_start.ensure(ncols);
	HIP_CHECK(hipMemcpyAsync(m_col_cell.p, col_cell.data(), size_t(ncols) * 4, hipMemcpyHostToDevice, stream));
	hipLaunchKernelGGL(count_nonzero_rows_kernel, dim3(div_up(ncols, 256)), dim3(256), 0, stream, m_col_cell.p, ncols, cell_cg_begin.p,
	                   cell_cg_count.p, cg_key.p, layout.gene_none, d_value.p, d_count.p);
	HIP_CHECK(hipGetLastError());
	std::vector<u32> cnt(ncols);
	fetch(cnt.data(), d_count.p, size_t(ncols) * 4);
	uint64_t nnz = 0;
	for (u32 c = 0; c < ncols; ++c) { M.colptr[c] = u32(nnz); nnz += cnt[c]; }
	if (nnz > 0xFFFFFFF0ull) throw UnsupportedError("count matrix with more than 2^32 non-zeros");
	M.colptr[ncols] = u32(nnz);
	M.nnz = nnz;
	if (!nnz) return;
	M.d_row.ensure(nnz); M.d_val.ensure(nnz); M.h_row.ensure(nnz); M.h_val.ensure(nnz);
	HIP_CHECK(hipMemcpyAsync(m_col_start.p, M.colptr.data(), size_t(ncols) * 4, hipMemcpyHostToDevice, stream));
	MatrixArgs a{};
	a.col_cell = m_col_cell.p; a.col_start = m_col_start.p; a.cell_cg_begin = cell_cg_begin.p; a.cell_cg_count = cell_cg_count.p; a.cg_key = cg_key.p;
	a.value = d_value.p; a.gene_mask = layout.gene_none; a.skip_zero = 1;
	a.t_gene = M.d_row.p; a.t_val = M.d_val.p;
	hipLaunchKernelGGL(emit_matrix_kernel<0>, dim3(ncols), dim3(256), 0, stream, a);
	HIP_CHECK(hipGetLastError());
	HIP_CHECK(hipMemcpyAsync(M.h_row.p, M.d_row.p, nnz * 4, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipMemcpyAsync(M.h_val.p, M.d_val.p, nnz * 4, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(stream_wait(stream));
}

// Walks a fetched slice of the molecule table in order, skipping the pseudo rows of gene-less reads and replacing
// the groups that the N-UMI merge rewrote by their host-side contents.
// `emit` also receives the molecule row whose quality sums the molecule shows (first_row = row of k[0]).
template <class F>
static void for_each_molecule(dropest_ctx *ctx, const std::vector<u64> &k, const std::vector<u32> &r, const std::vector<u32> &m, F &&emit,
                              u32 first_row = 0) {
	const KeyLayout &L = ctx->layout;
	const u64 umask = L.umi_bits ? ((1ull << L.umi_bits) - 1ull) : 0ull;
	u64 done_group = ~0ull;
	for (size_t i = 0; i < k.size(); ++i) {
		const u64 cg = k[i] >> L.umi_bits;
		const u64 g = cg & L.gene_none;
		if (g == L.gene_none) continue;   // reads without a gene never form molecules
		const u32 c = u32(cg >> L.gene_bits);
		if (!ctx->umi_overrides.empty()) {
			auto it = ctx->umi_overrides.find(cg);
			if (it != ctx->umi_overrides.end()) {
				if (done_group != cg) { for (const UmiOverride &o : it->second) emit(c, u32(g), o.umi, o.reads, o.mark, o.src_row); done_group = cg; }
				continue;
			}
		}
		emit(c, u32(g), ctx->unmap_umi(k[i] & umask), r[i], uint8_t(m[i]), first_row + u32(i));
	}
}

// ================================================================================================
// C-ABI
// ================================================================================================
extern "C" {

const char *dropest_last_error(void) { return g_last_error.c_str(); }

void dropest_cfg_defaults(dropest_cfg *cfg) {
	std::memset(cfg, 0, sizeof(*cfg));
	cfg->device = 0;
	cfg->merge_kind = DROPEST_MERGE_NONE;
	cfg->barcodes_kind = DROPEST_BARCODES_INDROP;   // MergeStrategyFactory.cpp:32 default "indrop"
	cfg->barcodes_file = nullptr;
	cfg->min_genes_before_merge = 10;
	cfg->min_genes_after_merge = 10;
	cfg->min_merge_fraction = 0.2;
	cfg->max_cb_merge_edit_distance = 2;
	cfg->umi_merge_kind = DROPEST_UMI_MERGE_SIMPLE;
	cfg->umi_merge_multiplier = 2.0;
	cfg->max_merge_prob = 1e-4;
	cfg->max_real_merge_prob = 1e-7;
	cfg->max_umi_merge_edit_distance = 1;
	cfg->gene_match_levels = "eEBA";
	cfg->max_cells = -1;
	cfg->cb_table_capacity = 0;
}

dropest_status dropest_ctx_create(const dropest_cfg *cfg, dropest_ctx **out) {
	if (!cfg || !out) { g_last_error = "null argument"; return DROPEST_ERR_INVALID; }
	*out = nullptr;
	return guarded([&] {
		std::unique_ptr<dropest_ctx> c(new dropest_ctx());
		c->init_from_cfg(*cfg);
		*out = c.release();
	});
}

void dropest_ctx_destroy(dropest_ctx *ctx) {
	if (!ctx) return;
	(void)hipSetDevice(ctx->cfg.device);
	delete ctx;
}

dropest_status dropest_set_side_strings(dropest_ctx *ctx, const char *const *strings, uint64_t n) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		if (n < ctx->side.size()) throw InvalidError("the side-string table may only grow");
		for (uint64_t i = ctx->side.size(); i < n; ++i) ctx->side.emplace_back(strings[i]);
	});
}

static void push_common(dropest_ctx *ctx, uint64_t n) {
	if (!ctx) throw InvalidError("null context");
	if (ctx->initialized) throw InvalidError("Container is already initialized");   // CellsDataContainer.cpp:61-62
	if (ctx->d_cb) throw InvalidError("reads were already frozen by a previous set_initialized");
	if (ctx->n_reads + n >= 0xFFFFFFFEull) throw UnsupportedError("more than 2^32-2 reads in one context (shard across GPUs)");
	HIP_CHECK(hipSetDevice(ctx->cfg.device));
}

dropest_status dropest_push_reads(dropest_ctx *ctx, const uint64_t *cb, const uint64_t *umi, const uint32_t *gene,
                                  const uint32_t *aux, uint64_t n) {
	return guarded([&] {
		push_common(ctx, n);
		if (n == 0) return;
		if (!cb || !umi || !gene || !aux) throw InvalidError("null read array");
		// one growing device buffer, fed on the copy stream: no allocation per batch, nothing to concatenate afterwards.  A
		// batch pushed between two adopted device chunks would have to keep its place in the stream: such mixes take a
		// chunk of their own.
		if (ctx->store_chunk >= 0 && size_t(ctx->store_chunk) + 1 != ctx->chunks.size()) {
			ReadChunk c;
			c.cb.alloc(n); c.umi.alloc(n); c.gene.alloc(n); c.aux.alloc(n);
			c.cb.mark_persistent(); c.umi.mark_persistent(); c.gene.mark_persistent(); c.aux.mark_persistent();
			HIP_CHECK(hipMemcpyAsync(c.cb.p, cb, n * 8, hipMemcpyHostToDevice, ctx->stream));
			HIP_CHECK(hipMemcpyAsync(c.umi.p, umi, n * 8, hipMemcpyHostToDevice, ctx->stream));
			HIP_CHECK(hipMemcpyAsync(c.gene.p, gene, n * 4, hipMemcpyHostToDevice, ctx->stream));
			HIP_CHECK(hipMemcpyAsync(c.aux.p, aux, n * 4, hipMemcpyHostToDevice, ctx->stream));
			HIP_CHECK(stream_wait(ctx->stream));
			c.p_cb = c.cb.p; c.p_umi = c.umi.p; c.p_gene = c.gene.p; c.p_aux = c.aux.p; c.n = n;
			ctx->chunks.push_back(std::move(c));
			ctx->n_reads += n;
			return;
		}
		if (ctx->store_chunk < 0) { ctx->store_chunk = long(ctx->chunks.size()); ctx->chunks.emplace_back(); }
		ctx->store.push(cb, umi, gene, aux, n);
		ctx->chunks[size_t(ctx->store_chunk)].n = ctx->store.n;   // the pointers are set when the reads are frozen (the buffer may still move)
		ctx->n_reads += n;
	});
}

dropest_status dropest_push_reads_gather(dropest_ctx *ctx, uint64_t n_segments, const uint64_t *const *cb, const uint64_t *const *umi,
                                         const uint32_t *const *gene, const uint32_t *const *aux, const uint64_t *counts) {
	return guarded([&] {
		if (n_segments && (!cb || !umi || !gene || !aux || !counts)) throw InvalidError("null segment array");
		uint64_t n = 0;
		for (uint64_t k = 0; k < n_segments; ++k) { n += counts[k]; if (counts[k] && (!cb[k] || !umi[k] || !gene[k] || !aux[k])) throw InvalidError("null read array"); }
		push_common(ctx, n);
		if (n == 0) return;
		if (ctx->store_chunk >= 0 && size_t(ctx->store_chunk) + 1 != ctx->chunks.size()) throw InvalidError("dropest_push_reads_gather after an adopted device chunk: push the segments one by one");
		if (ctx->store_chunk < 0) { ctx->store_chunk = long(ctx->chunks.size()); ctx->chunks.emplace_back(); }
		ctx->store.push_segments(size_t(n_segments), cb, umi, gene, aux, counts);
		ctx->chunks[size_t(ctx->store_chunk)].n = ctx->store.n;
		ctx->n_reads += n;
	});
}

dropest_status dropest_reserve_reads(dropest_ctx *ctx, uint64_t n_total) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		if (n_total >= 0xFFFFFFFEull) throw UnsupportedError("more than 2^32-2 reads in one context (shard across GPUs)");
		HIP_CHECK(hipSetDevice(ctx->cfg.device));
		ctx->store.reserve(size_t(n_total));
	});
}

dropest_status dropest_push_reads_device(dropest_ctx *ctx, const uint64_t *d_cb, const uint64_t *d_umi,
                                         const uint32_t *d_gene, const uint32_t *d_aux, uint64_t n, int adopt) {
	return guarded([&] {
		push_common(ctx, n);
		if (n == 0) return;
		if (!d_cb || !d_umi || !d_gene || !d_aux) throw InvalidError("null read array");
		// a copied chunk joins the growing store while that is the tail of the stream (every chunk so far was a copy): the reads
		// stay one block, which the facade's preview of an uninitialised container needs (dropest_resident_reads)
		if (!adopt && (ctx->store_chunk < 0 || size_t(ctx->store_chunk) + 1 == ctx->chunks.size())) {
			if (ctx->store_chunk < 0) { ctx->store_chunk = long(ctx->chunks.size()); ctx->chunks.emplace_back(); }
			ctx->store.push_device(d_cb, d_umi, d_gene, d_aux, n, ctx->stream);
			ctx->chunks[size_t(ctx->store_chunk)].n = ctx->store.n;
			ctx->n_reads += n;
			return;
		}
		ReadChunk c;
		c.n = n;
		if (adopt) {
			c.p_cb = reinterpret_cast<const u64 *>(d_cb); c.p_umi = reinterpret_cast<const u64 *>(d_umi);
			c.p_gene = d_gene; c.p_aux = d_aux;
		} else {
			c.cb.alloc(n); c.umi.alloc(n); c.gene.alloc(n); c.aux.alloc(n);
			c.cb.mark_persistent(); c.umi.mark_persistent(); c.gene.mark_persistent(); c.aux.mark_persistent();
			HIP_CHECK(hipMemcpyAsync(c.cb.p, d_cb, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
			HIP_CHECK(hipMemcpyAsync(c.umi.p, d_umi, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
			HIP_CHECK(hipMemcpyAsync(c.gene.p, d_gene, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
			HIP_CHECK(hipMemcpyAsync(c.aux.p, d_aux, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
			HIP_CHECK(stream_wait(ctx->stream));
			c.p_cb = c.cb.p; c.p_umi = c.umi.p; c.p_gene = c.gene.p; c.p_aux = c.aux.p;
		}
		ctx->chunks.push_back(std::move(c));
		ctx->n_reads += n;
	});
}

dropest_status dropest_set_initialized(dropest_ctx *ctx) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		HIP_CHECK(hipSetDevice(ctx->cfg.device));
		ctx->run_set_initialized();
	});
}

dropest_status dropest_ingest(dropest_ctx *ctx) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		HIP_CHECK(hipSetDevice(ctx->cfg.device));
		ctx->run_ingest();
		ctx->collect_timings();
	});
}

dropest_status dropest_ingest_summary_get(dropest_ctx *ctx, dropest_ingest_summary *out) {
	return guarded([&] {
		if (!ctx || !out) throw InvalidError("null argument");
		if (!ctx->ingested) throw InvalidError("dropest_ingest was not run");
		const IngestStats &g = ctx->ingest;
		out->umi_clean_min = g.umi_clean_min; out->umi_clean_max = g.umi_clean_max;
		out->umi_escape_max_plus1 = g.umi_escape_max_plus1; out->gene_max_plus1 = g.gene_max_plus1;
		out->chr_max_plus1 = g.chr_max_plus1; out->gene_chr_conflict = g.gene_chr_conflict; out->reserved = 0;
	});
}

dropest_status dropest_ingest_summary_set(dropest_ctx *ctx, const dropest_ingest_summary *in) {
	return guarded([&] {
		if (!ctx || !in) throw InvalidError("null argument");
		if (!ctx->ingested || ctx->initialized) throw InvalidError("the ingest summary is set between dropest_ingest and dropest_set_initialized");
		IngestStats &g = ctx->ingest;
		// widening only: every local value must stay representable
		if (in->umi_clean_min > g.umi_clean_min || in->umi_clean_max < g.umi_clean_max || in->umi_escape_max_plus1 < g.umi_escape_max_plus1 ||
		    in->gene_max_plus1 < g.gene_max_plus1 || in->chr_max_plus1 < g.chr_max_plus1 || (g.gene_chr_conflict && !in->gene_chr_conflict))
			throw InvalidError("the global ingest summary does not cover this shard's values");
		g.umi_clean_min = in->umi_clean_min; g.umi_clean_max = in->umi_clean_max; g.umi_escape_max_plus1 = in->umi_escape_max_plus1;
		g.gene_max_plus1 = in->gene_max_plus1; g.chr_max_plus1 = in->chr_max_plus1; g.gene_chr_conflict = in->gene_chr_conflict;
	});
}

dropest_status dropest_gene_chr_table(dropest_ctx *ctx, uint32_t **d_table, uint64_t *n) {
	return guarded([&] {
		if (!ctx || !d_table || !n) throw InvalidError("null argument");
		if (!ctx->ingested) throw InvalidError("dropest_ingest was not run");
		HIP_CHECK(hipSetDevice(ctx->cfg.device));
		if (!ctx->gene_chr.p) {   // a shard without reads: an all-unset table
			ctx->gene_chr.ensure(GENE_CHR_CAP);
			HIP_CHECK(hipMemsetAsync(ctx->gene_chr.p, 0xFF, size_t(GENE_CHR_CAP) * 4, ctx->stream));
			HIP_CHECK(stream_wait(ctx->stream));
		}
		*d_table = ctx->gene_chr.p; *n = GENE_CHR_CAP;
	});
}

dropest_status dropest_shard_merge_search(dropest_ctx *ctx, uint64_t n_global, const uint64_t *g_barcode, const uint32_t *g_n_genes,
                                          const int32_t *g_total_umis, uint64_t n_bases, const uint32_t *base_global,
                                          const uint32_t *base_local, uint64_t *n_pairs) {
	return guarded([&] {
		if (!ctx || !n_pairs) throw InvalidError("null argument");
		HIP_CHECK(hipSetDevice(ctx->cfg.device));
		ctx->shard_merge_search(n_global, g_barcode, g_n_genes, g_total_umis, n_bases, base_global, base_local, n_pairs);
	});
}

dropest_status dropest_shard_merge_pairs(dropest_ctx *ctx, uint32_t *pair_base_global, uint32_t *pair_cand_global) {
	return guarded([&] {
		if (!ctx || !ctx->shard) throw InvalidError("dropest_shard_merge_search was not run");
		const MergeSearch &S = ctx->shard->S;
		for (size_t p = 0; p < S.pair_base.size(); ++p) {
			pair_base_global[p] = ctx->shard->base_g[S.pair_base[p]];
			pair_cand_global[p] = S.pair_cand[p];
		}
	});
}

dropest_status dropest_shard_merge_export(dropest_ctx *ctx, uint64_t *n_listed, uint32_t *listed_global, uint64_t *row_offset,
                                          const uint64_t **d_low, const uint32_t *d_cols[4]) {
	return guarded([&] {
		if (!ctx || !ctx->shard || !n_listed) throw InvalidError("dropest_shard_merge_search was not run");
		dropest_ctx::ShardMerge &M = *ctx->shard;
		*n_listed = M.listed_f.size();
		if (listed_global) for (size_t i = 0; i < M.listed_f.size(); ++i) listed_global[i] = M.base_g[M.listed_f[i]];
		if (row_offset) for (size_t i = 0; i < M.row_offset.size(); ++i) row_offset[i] = M.row_offset[i];
		if (d_low) *d_low = reinterpret_cast<const uint64_t *>(M.x_low.p);
		if (d_cols) for (int k = 0; k < 4; ++k) d_cols[k] = M.x_col[k].p;
	});
}

dropest_status dropest_shard_merge_intersect(dropest_ctx *ctx, uint64_t n_pairs, const uint32_t *cand_local, const uint64_t *base_begin,
                                             const uint64_t *base_end, const uint64_t *d_base_low, uint32_t *inter) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		HIP_CHECK(hipSetDevice(ctx->cfg.device));
		ctx->shard_merge_intersect(n_pairs, cand_local, base_begin, base_end, d_base_low, inter);
	});
}

dropest_status dropest_shard_merge_decide(dropest_ctx *ctx, const uint32_t *inter, int64_t *target_global) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		HIP_CHECK(hipSetDevice(ctx->cfg.device));
		ctx->shard_merge_decide(inter, target_global);
	});
}

dropest_status dropest_merge_apply(uint64_t n_cells, uint64_t n_order, const uint32_t *order, const int64_t *target, int32_t *total_reads,
                                   int32_t *total_umis, uint32_t *final_target, uint8_t *excluded) {
	return guarded([&] {
		if (n_cells >= 0xFFFFFFF0ull) throw UnsupportedError("too many cells");
		apply_merge_order(u32(n_cells), size_t(n_order), order, target, total_reads, total_umis, final_target, excluded);
	});
}

dropest_status dropest_shard_merge_finish(dropest_ctx *ctx, uint64_t n_local, const uint32_t *local_id, const uint8_t *excluded,
                                          const uint8_t *merged_away, const int32_t *total_reads, const int32_t *total_umis,
                                          uint64_t n_moves, const uint32_t *move_src, const uint32_t *move_tgt, uint64_t n_import,
                                          const uint32_t *d_cell, const uint64_t *d_low, const uint32_t *const d_cols[4]) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		HIP_CHECK(hipSetDevice(ctx->cfg.device));
		ctx->shard_merge_finish(n_local, local_id, excluded, merged_away, total_reads, total_umis, n_moves, move_src, move_tgt, n_import,
		                        d_cell, d_low, d_cols);
	});
}

dropest_status dropest_merge_and_filter(dropest_ctx *ctx) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		HIP_CHECK(hipSetDevice(ctx->cfg.device));
		ctx->run_merge_and_filter();
	});
}

dropest_status dropest_reset_results(dropest_ctx *ctx) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		ctx->free_results();
	});
}

static void need_init(dropest_ctx *ctx) {
	if (!ctx) throw InvalidError("null context");
	if (!ctx->initialized) throw InvalidError("You must initialize container");
	HIP_CHECK(hipSetDevice(ctx->cfg.device));
}

dropest_status dropest_total_cells(dropest_ctx *ctx, uint64_t *n) {
	return guarded([&] { need_init(ctx); *n = ctx->n_cells; });
}
dropest_status dropest_real_cells(dropest_ctx *ctx, uint64_t *n) {
	return guarded([&] { need_init(ctx); *n = ctx->n_real_now; });
}

dropest_status dropest_cell_rows(dropest_ctx *ctx, uint64_t first, uint64_t count, dropest_cell_row *out) {
	static_assert(sizeof(dropest_cell_row) == sizeof(CellRowPod), "row layout");
	return guarded([&] {
		need_init(ctx);
		if (first + count > ctx->n_cells) throw RangeError("cell index out of range");
		if (count == 0) return;
		DevBuf<CellRowPod> rows; rows.alloc(count);
		CellArrays a{ctx->cell_cb.p, ctx->cell_first.p, ctx->cell_n_genes.p, ctx->cell_req_genes.p, ctx->cell_req_umis.p,
		             ctx->cell_total_umis.p, ctx->cell_total_reads.p};
		hipLaunchKernelGGL(gather_cell_rows_kernel, dim3(div_up(count, 256)), dim3(256), 0, ctx->stream, a,
		                   static_cast<const u32 *>(nullptr), u32(first), u32(count), rows.p);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(hipMemcpyAsync(out, rows.p, count * sizeof(CellRowPod), hipMemcpyDeviceToHost, ctx->stream));
		HIP_CHECK(stream_wait(ctx->stream));
		for (uint64_t j = 0; j < count; ++j) {   // overlay the host-tracked state of real-candidate cells
			const long ri = ctx->real_find(u32(first + j));
			if (ri < 0) continue;
			const HostCell &h = ctx->real[size_t(ri)];
			dropest_cell_row &r = out[j];
			r.n_genes = h.row.n_genes; r.requested_genes = h.row.requested_genes; r.requested_umis = h.row.requested_umis;
			r.total_reads = h.row.total_reads; r.total_umis = h.row.total_umis;
			r.is_merged = h.merged; r.is_excluded = h.excluded;
			r.is_real = !h.merged && !h.excluded && h.row.n_genes >= ctx->min_before;
		}
		for (u32 c : ctx->extra_excluded) if (c >= first && c < first + count) out[c - first].is_excluded = 1;
	});
}

dropest_status dropest_cell_id_by_cb(dropest_ctx *ctx, uint64_t barcode, int64_t *id) {
	return guarded([&] {
		need_init(ctx);
		*id = -1;
		if (ctx->n_cells == 0 || barcode == 0) return;
		// No device round trip per question (round 4 probed the device table with one synchronous 16-byte hipMemcpy per slot): the real
		// cells' barcodes are on the host already; the first question about any other barcode fetches the pass's barcode list once.
		dropest_ctx::CbMirror &m = ctx->cb_mirror;
		if (m.level == 0) {
			m.build(ctx->real.size(), [&](size_t i, u64 &k, u32 &v) { k = ctx->real[i].row.barcode; v = ctx->real[i].id; });
			m.level = 1;
		}
		long got = m.find(barcode);
		if (got < 0 && m.level == 1) {
			std::vector<u64> all(ctx->n_cells);
			ctx->fetch(all.data(), ctx->cell_cb.p, size_t(ctx->n_cells) * 8);
			m.build(all.size(), [&](size_t i, u64 &k, u32 &v) { k = all[i]; v = u32(i); });
			m.level = 2;
			got = m.find(barcode);
		}
		*id = got;
	});
}

dropest_status dropest_filtered_cells(dropest_ctx *ctx, uint64_t *n, uint64_t *ids) {
	return guarded([&] {
		need_init(ctx);
		const std::vector<uint64_t> &f = ctx->filtered_cells();
		*n = f.size();
		if (ids) std::copy(f.begin(), f.end(), ids);
	});
}

dropest_status dropest_merge_targets(dropest_ctx *ctx, uint64_t *n, uint64_t *src, uint64_t *tgt) {
	return guarded([&] {
		need_init(ctx);
		if (!ctx->merged) throw InvalidError("merge_and_filter has not run");
		// (pairs merged through dropest_merge_cells are not the strategy's: CellsDataContainer::_merge_targets is what
		// MergeStrategyAbstract::merge returned, CellsDataContainer.cpp:44)
		size_t k = 0;
		for (auto const &pr : ctx->merge_pairs) {
			if (ctx->explicit_sources.count(u32(pr.first))) continue;
			if (src && tgt) { src[k] = pr.first; tgt[k] = pr.second; }
			++k;
		}
		*n = k;
	});
}

dropest_status dropest_sort_layout(dropest_ctx *ctx, uint32_t out[7]) {
	return guarded([&] {
		need_init(ctx);
		const KeyLayout &L = ctx->layout;
		out[0] = L.cell_bits; out[1] = L.gene_bits; out[2] = L.umi_bits; out[3] = L.mark_shift; out[4] = L.val_bytes;
		out[5] = ctx->main_sort_passes; out[6] = ctx->main_sort_kind;
	});
}

dropest_status dropest_global_counters(dropest_ctx *ctx, uint64_t out[4]) {
	return guarded([&] {
		need_init(ctx);
		out[0] = ctx->counters.intergenic; out[1] = ctx->counters.exon; out[2] = ctx->counters.intron;
		out[3] = ctx->counters.not_annotated;
	});
}

static void fetch_molecule_range(dropest_ctx *ctx, u32 mb, u32 me, std::vector<u64> &k, std::vector<u32> &r, std::vector<u32> &m) {
	const u32 cnt = me - mb;
	k.resize(cnt); r.resize(cnt); m.resize(cnt);
	if (!cnt) return;
	HIP_CHECK(hipMemcpyAsync(k.data(), ctx->mol_key.p + mb, size_t(cnt) * 8, hipMemcpyDeviceToHost, ctx->stream));
	HIP_CHECK(hipMemcpyAsync(r.data(), ctx->mol_reads.p + mb, size_t(cnt) * 4, hipMemcpyDeviceToHost, ctx->stream));
	HIP_CHECK(hipMemcpyAsync(m.data(), ctx->mol_mark.p + mb, size_t(cnt) * 4, hipMemcpyDeviceToHost, ctx->stream));
	HIP_CHECK(stream_wait(ctx->stream));
}

dropest_status dropest_molecules(dropest_ctx *ctx, uint64_t *n, uint32_t *cell, uint32_t *gene, uint64_t *umi,
                                 uint32_t *reads, uint8_t *mark) {
	return guarded([&] {
		need_init(ctx);
		std::vector<u64> k; std::vector<u32> r, m;
		fetch_molecule_range(ctx, 0, ctx->n_mol, k, r, m);
		const KeyLayout &L = ctx->layout;
		const u64 umask = L.umi_bits ? ((1ull << L.umi_bits) - 1ull) : 0ull;
		uint64_t cnt = 0;
		for_each_molecule(ctx, k, r, m, [&](u32 c, u32 g, u64 u, u32 rd, uint8_t mk, u32) {
			if (cell) { cell[cnt] = c; gene[cnt] = g; umi[cnt] = u; reads[cnt] = rd; mark[cnt] = mk; }
			++cnt;
		});
		(void)umask; (void)L;
		*n = cnt;
	});
}

dropest_status dropest_cell_molecules(dropest_ctx *ctx, uint64_t cell_id, uint64_t *n, uint32_t *gene, uint64_t *umi,
                                      uint32_t *reads, uint8_t *mark) {
	return guarded([&] {
		need_init(ctx);
		if (cell_id >= ctx->n_cells) throw RangeError("cell index out of range");
		u32 cgb = 0, cgc = 0, mb = 0, me = 0;
		HIP_CHECK(hipMemcpy(&cgb, ctx->cell_cg_begin.p + cell_id, 4, hipMemcpyDeviceToHost));
		HIP_CHECK(hipMemcpy(&cgc, ctx->cell_cg_count.p + cell_id, 4, hipMemcpyDeviceToHost));
		if (cgc) {
			HIP_CHECK(hipMemcpy(&mb, ctx->cg_mol_begin.p + cgb, 4, hipMemcpyDeviceToHost));
			HIP_CHECK(hipMemcpy(&me, ctx->cg_mol_begin.p + cgb + cgc, 4, hipMemcpyDeviceToHost));
		}
		std::vector<u64> k; std::vector<u32> r, m;
		fetch_molecule_range(ctx, mb, me, k, r, m);
		const KeyLayout &L = ctx->layout;
		const u64 umask = L.umi_bits ? ((1ull << L.umi_bits) - 1ull) : 0ull;
		uint64_t cnt = 0;
		for_each_molecule(ctx, k, r, m, [&](u32, u32 g, u64 u, u32 rd, uint8_t mk, u32) {
			if (gene) { gene[cnt] = g; umi[cnt] = u; reads[cnt] = rd; mark[cnt] = mk; }
			++cnt;
		});
		(void)umask; (void)L;
		*n = cnt;
	});
}

dropest_status dropest_exclude_cell(dropest_ctx *ctx, uint64_t cell) {
	return guarded([&] {
		need_init(ctx);
		if (cell >= ctx->n_cells) throw RangeError("cell index out of range");
		ctx->mutate_exclude_cell(u32(cell));
	});
}

dropest_status dropest_merge_cells(dropest_ctx *ctx, uint64_t source_cell, uint64_t target_cell) {
	return guarded([&] {
		need_init(ctx);
		if (source_cell >= ctx->n_cells || target_cell >= ctx->n_cells) throw RangeError("cell index out of range");
		ctx->mutate_merge_cells(u32(source_cell), u32(target_cell));
	});
}

dropest_status dropest_merge_umis(dropest_ctx *ctx, uint64_t cell, uint32_t gene, uint64_t n, const uint64_t *source_umis,
                                  const uint64_t *target_umis) {
	return guarded([&] {
		need_init(ctx);
		if (cell >= ctx->n_cells) throw RangeError("cell index out of range");
		if (n && (!source_umis || !target_umis)) throw InvalidError("null UMI array");
		ctx->mutate_merge_umis(u32(cell), gene, n, source_umis, target_umis);
	});
}

dropest_status dropest_set_umi_qualities(dropest_ctx *ctx, const uint8_t *qualities, uint32_t quality_length, uint64_t n_reads) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		if (ctx->initialized || ctx->ingested) throw InvalidError("UMI qualities must be set before set_initialized");
		if (n_reads != ctx->n_reads) throw InvalidError("UMI qualities must cover every pushed read (" + std::to_string(ctx->n_reads) + ")");
		if (quality_length > 255) throw UnsupportedError("UMI quality strings longer than 255");
		if (n_reads && quality_length && !qualities) throw InvalidError("null quality array");
		HIP_CHECK(hipSetDevice(ctx->cfg.device));
		ctx->qual_len = quality_length; ctx->qual_reads = n_reads; ctx->have_qual = true; ctx->qual_var = false;
		const size_t bytes = size_t(n_reads) * quality_length;
		if (bytes) {
			ctx->umi_qual.alloc(bytes); ctx->umi_qual.mark_persistent();
			HIP_CHECK(hipMemcpyAsync(ctx->umi_qual.p, qualities, bytes, hipMemcpyHostToDevice, ctx->stream));
			HIP_CHECK(stream_wait(ctx->stream));
		}
	});
}

dropest_status dropest_set_umi_qualities_var(dropest_ctx *ctx, const uint8_t *qualities, uint32_t row_bytes, const uint8_t *lengths, uint64_t n_reads) {
	const dropest_status st = dropest_set_umi_qualities(ctx, qualities, row_bytes, n_reads);
	if (st != DROPEST_OK || !lengths) return st;
	return guarded([&] {
		for (uint64_t i = 0; i < n_reads; ++i)
			if (lengths[i] > row_bytes) throw InvalidError("a quality length beyond the row width (" + std::to_string(row_bytes) + ")");
		if (n_reads) {
			ctx->umi_qual_lens.alloc(n_reads); ctx->umi_qual_lens.mark_persistent();
			HIP_CHECK(hipMemcpyAsync(ctx->umi_qual_lens.p, lengths, n_reads, hipMemcpyHostToDevice, ctx->stream));
			HIP_CHECK(stream_wait(ctx->stream));
		}
		ctx->qual_var = true;
	});
}

dropest_status dropest_umi_quality_length(dropest_ctx *ctx, uint32_t *quality_length) {
	return guarded([&] {
		if (!ctx || !quality_length) throw InvalidError("null argument");
		*quality_length = ctx->have_qual ? ctx->qual_len : 0u;
	});
}

static void cell_molecule_quality_fetch(dropest_ctx *ctx, uint64_t cell_id, uint64_t n, uint32_t *quality_sums, uint32_t *lengths);
dropest_status dropest_cell_molecule_qualities(dropest_ctx *ctx, uint64_t cell_id, uint64_t n, uint32_t *quality_sums) {
	return guarded([&] {
		if (!quality_sums && ctx && ctx->have_qual && ctx->qual_len) throw InvalidError("null output array");
		cell_molecule_quality_fetch(ctx, cell_id, n, quality_sums, nullptr);
	});
}
dropest_status dropest_cell_molecule_quality_lengths(dropest_ctx *ctx, uint64_t cell_id, uint64_t n, uint32_t *lengths) {
	return guarded([&] {
		if (!lengths) throw InvalidError("null output array");
		for (uint64_t i = 0; i < n; ++i) lengths[i] = 0;
		cell_molecule_quality_fetch(ctx, cell_id, n, nullptr, lengths);
	});
}
static void cell_molecule_quality_fetch(dropest_ctx *ctx, uint64_t cell_id, uint64_t n, uint32_t *quality_sums, uint32_t *lengths) {
	{
		need_init(ctx);
		if (cell_id >= ctx->n_cells) throw RangeError("cell index out of range");
		if (!ctx->have_qual || ctx->qual_len == 0) { if (n) { /* nothing to write: quality length 0 */ } return; }
		u32 cgb = 0, cgc = 0, mb = 0, me = 0;
		HIP_CHECK(hipMemcpy(&cgb, ctx->cell_cg_begin.p + cell_id, 4, hipMemcpyDeviceToHost));
		HIP_CHECK(hipMemcpy(&cgc, ctx->cell_cg_count.p + cell_id, 4, hipMemcpyDeviceToHost));
		if (cgc) {
			HIP_CHECK(hipMemcpy(&mb, ctx->cg_mol_begin.p + cgb, 4, hipMemcpyDeviceToHost));
			HIP_CHECK(hipMemcpy(&me, ctx->cg_mol_begin.p + cgb + cgc, 4, hipMemcpyDeviceToHost));
		}
		std::vector<u64> k; std::vector<u32> r, m, rows;
		fetch_molecule_range(ctx, mb, me, k, r, m);
		for_each_molecule(ctx, k, r, m, [&](u32, u32, u64, u32, uint8_t, u32 row) { rows.push_back(row); }, mb);
		if (rows.size() != n) throw InvalidError("the cell has " + std::to_string(rows.size()) + " molecules, not " + std::to_string(n));
		ctx->fetch_quality_rows(rows, quality_sums, lengths);
	}
}

dropest_status dropest_count_matrix_csc(dropest_ctx *ctx, int filtered, int reads_output, uint64_t *ncols, uint64_t *nnz,
                                        const uint32_t **colptr, const uint32_t **rowidx, const uint32_t **values) {
	return guarded([&] {
		need_init(ctx);
		ctx->emit_matrix(filtered != 0, reads_output != 0);
		const dropest_ctx::MatrixResult &M = ctx->mat[filtered ? 0 : 1];
		*ncols = M.ncols; *nnz = M.nnz;
		*colptr = M.colptr.data(); *rowidx = M.h_row.p; *values = M.h_val.p;
	});
}

dropest_status dropest_prefetch_raw_matrix(dropest_ctx *ctx, int reads_output) {
	return guarded([&] {
		need_init(ctx);
		ctx->prefetch_raw_matrix(reads_output != 0, 0);
	});
}

dropest_status dropest_prefetch_raw_matrix_narrow(dropest_ctx *ctx, int reads_output) {
	return guarded([&] {
		need_init(ctx);
		ctx->prefetch_raw_matrix(reads_output != 0, 1);
	});
}

dropest_status dropest_narrow_matrix_possible(dropest_ctx *ctx, int *possible) {
	return guarded([&] { need_init(ctx); if (!possible) throw InvalidError("null argument"); *possible = ctx->narrow_possible() ? 1 : 0; });
}

dropest_status dropest_count_matrix_csc_narrow(dropest_ctx *ctx, int filtered, int reads_output, uint64_t *ncols, uint64_t *nnz,
                                               const uint32_t **colptr, const uint16_t **rowidx, const uint16_t **values,
                                               uint64_t *n_overflow, const uint32_t **overflow_pos, const uint32_t **overflow_val) {
	return guarded([&] {
		need_init(ctx);
		if (!ncols || !nnz || !colptr || !rowidx || !values || !n_overflow || !overflow_pos || !overflow_val) throw InvalidError("null argument");
		ctx->emit_matrix(filtered != 0, reads_output != 0, true, 1);
		const dropest_ctx::MatrixResult &M = ctx->mat[filtered ? 0 : 1];
		*ncols = M.ncols; *nnz = M.nnz;
		*colptr = M.colptr.data(); *rowidx = M.h_row16.p; *values = M.h_val16.p;
		*n_overflow = M.n_ovf; *overflow_pos = M.h_ovf.p ? M.h_ovf.p + 1 : nullptr; *overflow_val = M.h_ovf.p ? M.h_ovf.p + 1 + M.vcap : nullptr;
	});
}

dropest_status dropest_set_raw_matrix_prefetch(dropest_ctx *ctx, int form, int reads_output) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		if (form < -1 || form > 2) throw InvalidError("form: -1 (off), 0 (32-bit), 1 (16-bit), 2 (bytes)");
		ctx->auto_pf_form = form; ctx->auto_pf_reads = reads_output != 0;
	});
}

dropest_status dropest_set_matrix_wire(dropest_ctx *ctx, int enabled) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		ctx->invalidate_prefetch();
		ctx->matrix_wire = enabled != 0;
	});
}

dropest_status dropest_set_umi_dictionary(dropest_ctx *ctx, int mode) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		if (mode < 0 || mode > 2) throw InvalidError("UMI dictionary mode out of range");
		if (ctx->initialized) throw InvalidError("Container is already initialized");
		ctx->umi_dict_mode = mode;
	});
}

dropest_status dropest_prefetch_raw_matrix_bytes(dropest_ctx *ctx, int reads_output) {
	return guarded([&] {
		need_init(ctx);
		ctx->prefetch_raw_matrix(reads_output != 0, 2);
	});
}

dropest_status dropest_count_matrix_csc_bytes(dropest_ctx *ctx, int filtered, int reads_output, dropest_matrix_bytes *out) {
	return guarded([&] {
		need_init(ctx);
		if (!out) throw InvalidError("null argument");
		ctx->emit_matrix(filtered != 0, reads_output != 0, true, 2);
		const dropest_ctx::MatrixResult &M = ctx->mat[filtered ? 0 : 1];
		out->ncols = M.ncols; out->nnz = M.nnz; out->colptr = M.colptr.data();
		out->row_delta = M.h_drow8.p; out->value = M.h_val8.p;
		out->n_row_listed = M.n_rovf; out->row_listed_pos = M.h_rovf.p ? M.h_rovf.p + 1 : nullptr; out->row_listed_row = M.h_rovf.p ? M.h_rovf.p + 1 + M.rcap : nullptr;
		out->n_value_listed = M.n_ovf; out->value_listed_pos = M.h_ovf.p ? M.h_ovf.p + 1 : nullptr; out->value_listed_value = M.h_ovf.p ? M.h_ovf.p + 1 + M.vcap : nullptr;
	});
}

// dgCMatrix slots i / x from the byte form.  The listed entries come in no particular order: they are written to their places first, then
// every column is walked once (a column's first delta counts from row -1; a 255 takes what the first phase put there).  The walk is the one
// dropest_count_matrix_csc runs under its copies (matrix_decode.h), here with everything already on the host; the calling thread takes part.
dropest_status dropest_matrix_bytes_widen(const dropest_matrix_bytes *m, uint32_t *rowidx, uint32_t *values) {
	return guarded([&] {
		if (!m || (m->nnz && (!rowidx || !values))) throw InvalidError("null argument");
		if (!m->nnz) return;
		if (m->n_row_listed > 0xFFFFFFFFull || m->n_value_listed > 0xFFFFFFFFull) throw InvalidError("byte matrix: a list longer than the matrix");
		auto job = std::make_shared<dropest::DecodeJob>();
		(void)hipGetDevice(&job->device);
		(void)hipGetLastError();   // (no device: the walk needs none)
		job->m.rd = m->row_delta; job->m.vb = m->value; job->m.colptr = m->colptr; job->m.ncols = m->ncols; job->m.nnz = m->nnz;
		job->ro = rowidx; job->vo = values;
		const uint32_t nr = uint32_t(m->n_row_listed), nv = uint32_t(m->n_value_listed);
		job->r_count = &nr; job->r_pos = m->row_listed_pos; job->r_val = m->row_listed_row; job->rcap = nr;
		job->v_count = &nv; job->v_pos = m->value_listed_pos; job->v_val = m->value_listed_value; job->vcap = nv;
		job->check_marks = true;
		// (read at every call: tests run small slices with workers that nap between claiming a slice and walking it)
		const char *e_slice = getenv("DROPEST_DECODE_SLICE"), *e_delay = getenv("DROPEST_DECODE_TEST_DELAY_US");
		if (e_delay) job->test_delay_us = uint32_t(std::max(0, atoi(e_delay)));
		job->prepare(e_slice && atoll(e_slice) >= 64 ? uint64_t(atoll(e_slice)) : uint64_t(1) << 16);
		dropest::DecodePool::get().submit(job);
		job->work(true);
		const int st = job->wait();
		job->quiesce();   // the caller's arrays are its own again when this returns
		// every slice is counted exactly once, by whoever finished it first: a second count would have ended the job one slice early
		if (st == dropest::DecodeJob::DONE && job->slice_done.load() != uint32_t(job->slice_end.size()))
			throw DeviceError("byte matrix: internal: " + std::to_string(job->slice_done.load()) + " slices counted, " + std::to_string(job->slice_end.size()) + " exist");
		if (st == dropest::DecodeJob::BAD_ROW) throw InvalidError("byte matrix: a listed row does not stand on a 255");
		if (st == dropest::DecodeJob::BAD_VALUE) throw InvalidError("byte matrix: a listed value does not stand on a 255");
		if (st != dropest::DecodeJob::DONE) throw InvalidError("byte matrix: the decode failed");
	});
}

dropest_status dropest_matrix_rider_widen(const dropest_matrix_bytes *base, const uint32_t *base_rows, const uint8_t *value, const uint32_t *out_begin,
                                          const uint32_t *out_count, uint64_t rider_nnz, uint64_t n_listed, const uint32_t *listed_pos,
                                          const uint32_t *listed_value, uint32_t *rowidx, uint32_t *values) {
	return guarded([&] {
		if (!base || (base->nnz && (!base_rows || !value)) || (base->ncols && (!out_begin || !out_count)) || (rider_nnz && (!rowidx || !values)) ||
		    (n_listed && (!listed_pos || !listed_value))) throw InvalidError("null argument");
		if (n_listed > 0xFFFFFFFFull || rider_nnz > 0xFFFFFFF0ull) throw InvalidError("rider matrix: too many entries");
		if (!base->nnz || !rider_nnz) return;
		auto job = std::make_shared<dropest::DecodeJob>();
		(void)hipGetDevice(&job->device);
		(void)hipGetLastError();   // (no device: the walk needs none)
		job->m.rd = base->row_delta; job->m.vb = value; job->m.colptr = base->colptr; job->m.ncols = base->ncols; job->m.nnz = rider_nnz;
		job->ro = rowidx; job->vo = values;
		const uint32_t nv = uint32_t(n_listed);
		job->v_count = &nv; job->v_pos = listed_pos; job->v_val = listed_value; job->vcap = nv;
		job->derived = true;
		job->dv.base_ro = base_rows; job->dv.out_begin = out_begin; job->dv.out_count = out_count;
		const char *e_slice = getenv("DROPEST_DECODE_SLICE"), *e_delay = getenv("DROPEST_DECODE_TEST_DELAY_US");
		if (e_delay) job->test_delay_us = uint32_t(std::max(0, atoi(e_delay)));
		job->prepare(e_slice && atoll(e_slice) >= 64 ? uint64_t(atoll(e_slice)) : uint64_t(1) << 16);
		dropest::DecodePool::get().submit(job);
		job->work(true);
		const int st = job->finish_rider();
		job->quiesce();
		if (st == dropest::DecodeJob::BAD_ROW) throw InvalidError("rider matrix: a column keeps another number of entries than announced");
		if (st == dropest::DecodeJob::BAD_VALUE) throw InvalidError("rider matrix: a listed value lies outside the matrix");
		if (st != dropest::DecodeJob::DONE) throw InvalidError("rider matrix: the decode failed");
	});
}

dropest_status dropest_count_matrix_csc_levels(dropest_ctx *ctx, const char *gene_match_levels, int reads_output, uint64_t *ncols,
                                               uint64_t *nnz, const uint32_t **colptr, const uint32_t **rowidx, const uint32_t **values) {
	return guarded([&] {
		need_init(ctx);
		if (!gene_match_levels) throw InvalidError("null gene_match_levels");
		ctx->emit_matrix_levels(query_mask_from_code(gene_match_levels), reads_output != 0);
		const dropest_ctx::MatrixResult &M = ctx->mat[2];
		*ncols = M.ncols; *nnz = M.nnz;
		*colptr = M.colptr.data(); *rowidx = M.h_row.p; *values = M.h_val.p;
	});
}

dropest_status dropest_count_matrix(dropest_ctx *ctx, int filtered, int reads_output, uint64_t *nnz, uint32_t *gene,
                                    uint32_t *col, uint32_t *val) {
	return guarded([&] {
		need_init(ctx);
		ctx->emit_matrix(filtered != 0, reads_output != 0);
		const dropest_ctx::MatrixResult &M = ctx->mat[filtered ? 0 : 1];
		*nnz = M.nnz;
		if (gene && col && val) {
			for (uint64_t c = 0; c < M.ncols; ++c)
				for (u32 k = M.colptr[c]; k < M.colptr[c + 1]; ++k) { gene[k] = M.h_row.p[k]; col[k] = u32(c); val[k] = M.h_val.p[k]; }
		}
	});
}

dropest_status dropest_chr_stats(dropest_ctx *ctx, uint64_t *n, uint32_t *cell, uint32_t *kind, uint32_t *chr, int32_t *count) {
	return guarded([&] {
		need_init(ctx);
		*n = 0;
		// real cells now, and the cell -> real index map (merged sources fold into their targets)
		std::vector<u32> real_ids;
		std::vector<u32> map(ctx->n_cells, 0xFFFFFFFFu);
		for (const HostCell &h : ctx->real)
			if (!h.merged && !h.excluded && h.row.n_genes >= ctx->min_before) { map[h.id] = u32(real_ids.size()); real_ids.push_back(h.id); }
		for (auto &p : ctx->merge_pairs) map[p.first] = map[p.second];
		if (real_ids.empty()) return;
		if (!ctx->chr_from_gene && ctx->n_chr_rows == 0) return;
		u32 n_chr = 0;
		if (ctx->chr_from_gene) {
			n_chr = ctx->ingest.chr_max_plus1;
		} else {   // number of chromosomes = 1 + max chr id seen in the partial rows
			std::vector<u64> keys(ctx->n_chr_rows);
			HIP_CHECK(hipMemcpy(keys.data(), ctx->chr_row_key.p, size_t(ctx->n_chr_rows) * 8, hipMemcpyDeviceToHost));
			for (u64 k : keys) n_chr = std::max(n_chr, u32(k & 0xFFFF) + 1);
		}
		if (n_chr == 0) return;
		const size_t cells_n = real_ids.size(), tab = cells_n * 3 * n_chr;
		if (tab > (size_t(1) << 30)) throw UnsupportedError("per-chromosome table too large for the dense path");
		DevBuf<u32> d_map, d_tab;
		d_map.alloc(ctx->n_cells); d_tab.alloc(tab);
		HIP_CHECK(hipMemcpyAsync(d_map.p, map.data(), size_t(ctx->n_cells) * 4, hipMemcpyHostToDevice, ctx->stream));
		HIP_CHECK(hipMemsetAsync(d_tab.p, 0, tab * 4, ctx->stream));
		if (ctx->chr_from_gene) {
			ChrFromGeneArgs a{};
			a.cg_key = ctx->cg_key.p; a.cg_exon = ctx->cg_exon.p; a.cg_intron = ctx->cg_intron.p; a.cg_mol_begin = ctx->cg_mol_begin.p;
			a.n_cg = ctx->n_cg; a.mol_key = ctx->mol_key.p; a.mol_reads = ctx->mol_reads.p;
			a.gene_bits = ctx->layout.gene_bits; a.umi_bits = ctx->layout.umi_bits; a.gene_none = ctx->layout.gene_none;
			a.gene_chr = ctx->gene_chr.p; a.real_index = d_map.p; a.n_chr = n_chr; a.table = d_tab.p;
			hipLaunchKernelGGL(chr_from_gene_kernel, dim3(div_up(ctx->n_cg, 256)), dim3(256), 0, ctx->stream, a);
		} else {
			hipLaunchKernelGGL(chr_accumulate_kernel, dim3(div_up(ctx->n_chr_rows, 256)), dim3(256), 0, ctx->stream,
			                   ctx->chr_row_key.p, ctx->chr_exon.p, ctx->chr_intron.p, ctx->chr_inter.p, ctx->n_chr_rows, d_map.p,
			                   n_chr, d_tab.p);
		}
		HIP_CHECK(hipGetLastError());
		std::vector<u32> t(tab);
		HIP_CHECK(hipMemcpyAsync(t.data(), d_tab.p, tab * 4, hipMemcpyDeviceToHost, ctx->stream));
		HIP_CHECK(stream_wait(ctx->stream));
		uint64_t cnt = 0;
		for (size_t ci = 0; ci < cells_n; ++ci)
			for (u32 k = 0; k < 3; ++k)
				for (u32 ch = 0; ch < n_chr; ++ch) {
					u32 v = t[(ci * 3 + k) * n_chr + ch];
					if (!v) continue;
					if (cell) { cell[cnt] = real_ids[ci]; kind[cnt] = k; chr[cnt] = ch; count[cnt] = int32_t(v); }
					++cnt;
				}
		*n = cnt;
	});
}

dropest_status dropest_umi_distribution(dropest_ctx *ctx, uint64_t *n, uint64_t *umi, uint64_t *count) {
	return guarded([&] {
		need_init(ctx);
		*n = 0;
		if (ctx->n_mol == 0) return;
		// flags of the filtered cells
		std::vector<u32> flags(ctx->n_cells, 0);
		for (uint64_t id : ctx->filtered_cells()) flags[id] = 1;
		ctx->remap.ensure(ctx->n_cells);
		HIP_CHECK(hipMemcpyAsync(ctx->remap.p, flags.data(), size_t(ctx->n_cells) * 4, hipMemcpyHostToDevice, ctx->stream));
		const u32 nm = ctx->n_mol;
		ctx->keys_a.ensure(nm); ctx->keys_b.ensure(nm); ctx->vals_a.ensure(nm); ctx->vals_b.ensure(nm);
		ctx->scalars.ensure(16);
		HIP_CHECK(hipMemsetAsync(ctx->scalars.p, 0, 4, ctx->stream));
		const KeyLayout &L = ctx->layout;
		hipLaunchKernelGGL(emit_filtered_umis_kernel, dim3(div_up(nm, 256)), dim3(256), 0, ctx->stream, ctx->mol_key.p, nm, L.umi_bits,
		                   L.gene_bits, L.gene_none, ctx->remap.p, ctx->keys_a.p, ctx->scalars.p);
		HIP_CHECK(hipGetLastError());
		u32 kept = 0;
		ctx->fetch(&kept, ctx->scalars.p, 4);
		std::map<u64, uint64_t> dist;   // API code -> molecules
		if (kept) {
			u64 *k = ctx->keys_a.p, *k_alt = ctx->keys_b.p;
			u32 *v = ctx->vals_a.p, *v_alt = ctx->vals_b.p;
			ctx->radix_sort(k, v, k_alt, v_alt, kept, L.umi_bits >= 64 ? ~0ull : ((1ull << L.umi_bits) - 1ull));
			UmiRuns p{};
			p.keys = k;
			DevBuf<u64> run_key; DevBuf<u32> run_cnt;
			const u32 runs = run_segmented_reduce(*ctx, "umi_runs", p, kept, 8, [&](u32 total) {
				run_key.alloc(total + 1); run_cnt.alloc(total + 1);
				zero_async(*ctx, run_cnt.p, size_t(total + 1) * 4);
				p.run_key = run_key.p; p.out[0] = run_cnt.p;
			});
			std::vector<u64> hk(runs); std::vector<u32> hc(runs);
			ctx->fetch(hk.data(), run_key.p, size_t(runs) * 8);
			ctx->fetch(hc.data(), run_cnt.p, size_t(runs) * 4);
			for (u32 i = 0; i < runs; ++i) dist[ctx->unmap_umi(hk[i])] += hc[i];
		}
		// groups rewritten by the N-UMI merge: replace their device molecules by the host-side contents
		if (!ctx->umi_overrides.empty()) {
			const u64 umask = L.umi_bits ? ((1ull << L.umi_bits) - 1ull) : 0ull;
			for (auto const &kv : ctx->umi_overrides) {
				const u32 cell = u32(kv.first >> L.gene_bits);
				if (!flags[cell]) continue;
				// device molecules of the group
				u32 cgb = 0, cgc = 0;
				HIP_CHECK(hipMemcpy(&cgb, ctx->cell_cg_begin.p + cell, 4, hipMemcpyDeviceToHost));
				HIP_CHECK(hipMemcpy(&cgc, ctx->cell_cg_count.p + cell, 4, hipMemcpyDeviceToHost));
				std::vector<u64> cgk(cgc); std::vector<u32> mb(cgc + 1);
				HIP_CHECK(hipMemcpy(cgk.data(), ctx->cg_key.p + cgb, size_t(cgc) * 8, hipMemcpyDeviceToHost));
				HIP_CHECK(hipMemcpy(mb.data(), ctx->cg_mol_begin.p + cgb, size_t(cgc + 1) * 4, hipMemcpyDeviceToHost));
				for (u32 j = 0; j < cgc; ++j) {
					if (cgk[j] != kv.first) continue;
					std::vector<u64> mk(mb[j + 1] - mb[j]);
					HIP_CHECK(hipMemcpy(mk.data(), ctx->mol_key.p + mb[j], mk.size() * 8, hipMemcpyDeviceToHost));
					for (u64 k2 : mk) { auto it = dist.find(ctx->unmap_umi(k2 & umask)); if (it != dist.end() && --it->second == 0) dist.erase(it); }
				}
				for (const UmiOverride &o : kv.second) dist[o.umi]++;
			}
		}
		*n = dist.size();
		if (umi && count) { size_t i = 0; for (auto const &kv : dist) { umi[i] = kv.first; count[i] = kv.second; ++i; } }
	});
}

// CellsDataContainer::umi_indexer() (CellsDataContainer.h:118): the UMIs in index order = order of first appearance among the
// gene-bearing reads (Gene::add_umi -> StringIndexer::add, Gene.cpp:19), followed by the UMIs only merges brought in (random
// fills of N-UMIs, explicit merge_umis targets: Gene.cpp:47) -- those in (cell id, gene id, UMI code) order of their groups.
dropest_status dropest_umi_first_seen(dropest_ctx *ctx, uint64_t *n_out, uint64_t *umi_codes) {
	return guarded([&] {
		need_init(ctx);
		if (!n_out) throw InvalidError("null argument");
		*n_out = 0;
		dropest_ctx &c = *ctx;
		std::vector<u64> codes;
		if (c.n_reads && c.ingest.gene_max_plus1) {
			const u32 n = u32(c.n_reads);
			// distinct UMI codes are bounded by the UMI field of the key layout
			const uint64_t field = c.umi_clean_bits >= 40 ? c.n_reads : std::min<uint64_t>(c.n_reads, (uint64_t(1) << (c.umi_clean_bits + (c.umi_sentinel_stripped ? 0 : 1))) + c.ingest.umi_escape_max_plus1);
			uint64_t cap = 1024; while (cap < field * 2) cap <<= 1;
			for (int attempt = 0;; ++attempt) {
				if (cap > (1ull << 32)) throw UnsupportedError("UMI table would exceed 2^32 slots");
				DevBuf<CbSlot> slots; slots.alloc(cap);
				CbTable t{slots.p, cap - 1};
				HIP_CHECK(hipMemsetAsync(slots.p, 0, cap * sizeof(CbSlot), c.stream));
				c.scalars.ensure(16);
				HIP_CHECK(hipMemsetAsync(c.scalars.p, 0, 8, c.stream));
				c.need_columns();
				hipLaunchKernelGGL(umi_insert_kernel, dim3(std::min<u32>(div_up(n, 256), 8192u)), dim3(256), 0, c.stream, c.d_umi, c.d_gene, n, t, c.scalars.p + 1);
				c.keys_a.ensure(std::max<size_t>(n, 1)); c.keys_b.ensure(std::max<size_t>(n, 1)); c.vals_a.ensure(1); c.vals_b.ensure(1);
				hipLaunchKernelGGL(cb_compact_slots_kernel, dim3(u32(std::min<uint64_t>((cap + 4095) / 4096, 4096))), dim3(256), 0, c.stream, t, c.keys_a.p, c.scalars.p);
				HIP_CHECK(hipGetLastError());
				u32 head[2] = {0, 0};
				c.fetch(head, c.scalars.p, 8);
				if (head[1]) { if (attempt >= 4) throw DeviceError("UMI table overflow"); cap <<= 2; continue; }
				const u32 distinct = head[0];
				u64 *k = c.keys_a.p, *k_alt = c.keys_b.p;
				u32 *v = c.vals_a.p, *v_alt = c.vals_b.p;
				const int ord_bits = std::max(1, bit_length(uint64_t(n ? n - 1 : 0)));
				c.radix_sort(k, v, k_alt, v_alt, distinct, ((1ull << ord_bits) - 1ull) << 32, 0, "umi_index:");
				DevBuf<u64> out; out.alloc(std::max<u32>(distinct, 1));
				hipLaunchKernelGGL(gather_slot_keys_kernel, dim3(div_up(std::max<u32>(distinct, 1), 256)), dim3(256), 0, c.stream, k, distinct, t, out.p);
				HIP_CHECK(hipGetLastError());
				codes.resize(distinct);
				c.fetch(codes.data(), out.p, size_t(distinct) * 8);
				break;
			}
		}
		if (!c.umi_overrides.empty()) {   // UMIs that exist only because a merge created them
			std::unordered_set<u64> known(codes.begin(), codes.end());
			std::vector<u64> groups;
			for (auto const &kv : c.umi_overrides) groups.push_back(kv.first);
			std::sort(groups.begin(), groups.end());
			for (u64 g : groups) for (const UmiOverride &o : c.umi_overrides.at(g)) if (known.insert(o.umi).second) codes.push_back(o.umi);
		}
		*n_out = codes.size();
		if (umi_codes) std::copy(codes.begin(), codes.end(), umi_codes);
	});
}

dropest_status dropest_add_umi_to_cell(dropest_ctx *ctx, uint64_t cell, uint32_t gene, uint64_t umi_code, uint32_t mark,
                                       const uint8_t *umi_quality, uint32_t quality_length) {
	return guarded([&] {
		need_init(ctx);
		if (cell >= ctx->n_cells) throw RangeError("cell index out of range");
		ctx->mutate_add_umi_to_cell(u32(cell), gene, umi_code, mark, umi_quality, quality_length);
	});
}

dropest_status dropest_collisions_adjusted_sizes(int device, const double *umi_probabilities, uint64_t n, uint64_t max_expression,
                                                 uint64_t *adjusted_sizes) {
	return guarded([&] {
		if (!umi_probabilities || !adjusted_sizes) throw InvalidError("null argument");
		HIP_CHECK(hipSetDevice(device));
		if (max_expression == 0) return;
		DevBuf<double> d_p, d_np, d_partial; DevBuf<CollisionState> d_st; DevBuf<u64> d_adj;
		d_p.alloc(n); d_np.alloc(n); d_partial.alloc(CA_BLOCKS); d_st.alloc(1); d_adj.alloc(max_expression);
		std::vector<double> ones(n, 1.0);
		HIP_CHECK(hipMemcpy(d_p.p, umi_probabilities, n * 8, hipMemcpyHostToDevice));
		HIP_CHECK(hipMemcpy(d_np.p, ones.data(), n * 8, hipMemcpyHostToDevice));
		CollisionState st{0.0, 0ull, 0ull};
		HIP_CHECK(hipMemcpy(d_st.p, &st, sizeof(st), hipMemcpyHostToDevice));
		hipLaunchKernelGGL(collisions_finish_kernel, dim3(1), dim3(1), 0, nullptr, d_partial.p, d_st.p, 0ull, d_adj.p);   // exponent of s = 1
		for (uint64_t s = 1; s <= max_expression; ++s) {
			hipLaunchKernelGGL(collisions_step_kernel, dim3(CA_BLOCKS), dim3(CA_THREADS), 0, nullptr, d_p.p, d_np.p, n, d_st.p, d_partial.p);
			hipLaunchKernelGGL(collisions_finish_kernel, dim3(1), dim3(1), 0, nullptr, d_partial.p, d_st.p, s, d_adj.p);
		}
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(hipMemcpy(adjusted_sizes, d_adj.p, max_expression * 8, hipMemcpyDeviceToHost));
	});
}

dropest_status dropest_poisson_intersection_prob(dropest_ctx *ctx, uint64_t cell1, uint64_t cell2, uint64_t *intersection_size,
                                                 double *expected_intersection_size, double *merge_probability) {
	return guarded([&] {
		need_init(ctx);
		if (ctx->merged) throw InvalidError("the estimator is defined on the un-merged state (call before merge_and_filter)");
		if (cell1 >= ctx->n_cells || cell2 >= ctx->n_cells) throw RangeError("cell index out of range");
		const std::vector<u32> b{u32(cell1)}, c{u32(cell2)};
		const u32 inter = ctx->pair_intersections(b, c)[0];
		double expected = -1, prob = 1;
		if (inter) {
			expected = ctx->poisson_expected_intersections(b, c)[0];
			prob = poisson_upper_tail(long(inter), expected);
		}
		if (intersection_size) *intersection_size = inter;
		if (expected_intersection_size) *expected_intersection_size = expected;
		if (merge_probability) *merge_probability = prob;
	});
}

dropest_status dropest_merge_target(dropest_ctx *ctx, uint64_t cell, int64_t *target) {
	return guarded([&] {
		need_init(ctx);
		if (ctx->merged) throw InvalidError("merge targets are defined on the un-merged state (call before merge_and_filter)");
		if (ctx->cfg.merge_kind != DROPEST_MERGE_REAL_BARCODES && ctx->cfg.merge_kind != DROPEST_MERGE_POISSON_REAL) {
			*target = int64_t(cell);   // DummyMergeStrategy
			return;
		}
		if (cell >= ctx->n_cells) throw RangeError("cell index out of range");
		*target = ctx->compute_merge_targets(std::vector<u32>{u32(cell)}, std::vector<u32>{ctx->real_at(u32(cell))})[0];
	});
}

// ---- multi-GPU building blocks ---------------------------------------------------------------------------------
uint32_t dropest_owner_of(uint64_t barcode, uint32_t n_parts) { return n_parts ? uint32_t(mix64(barcode) % n_parts) : 0u; }

static void partition_plan(u32 n, u32 &nblocks, u32 &tpb, size_t &off_k1, size_t &off_hist, size_t &off_row, size_t &off_base, size_t &total) {
	const u32 n_tiles = div_up(n, OP_TILE);
	nblocks = std::min<u32>(std::max<u32>(n_tiles, 1u), 1024);
	tpb = div_up(std::max<u32>(n_tiles, 1u), nblocks);
	nblocks = div_up(std::max<u32>(n_tiles, 1u), tpb);
	auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
	off_k1 = 0;
	off_hist = 0;
	off_row = off_hist + up(size_t(RS_RADIX) * nblocks * 4);
	off_base = off_row + up(size_t(RS_RADIX) * 4);
	total = off_base + up(size_t(RS_RADIX) * 4);
}

dropest_status dropest_partition_scratch_bytes(uint64_t n, uint64_t *bytes) {
	return guarded([&] {
		if (n >= 0xFFFFFFFEull) throw UnsupportedError("more than 2^32-2 reads per GPU");
		u32 nb, tpb; size_t a, b, c, d, total;
		partition_plan(u32(n), nb, tpb, a, b, c, d, total);
		*bytes = total;
	});
}

dropest_status dropest_partition_by_owner(int device, const uint64_t *d_cb, const uint64_t *d_umi, const uint32_t *d_gene,
                                          const uint32_t *d_aux, uint64_t n64, uint32_t n_parts, uint64_t *d_out_cb,
                                          uint64_t *d_out_umi, uint32_t *d_out_gene, uint32_t *d_out_aux, uint32_t *d_out_idx,
                                          uint64_t *counts, void *d_scratch, uint64_t scratch_bytes) {
	return guarded([&] {
		partition_by_owner_on(device, nullptr, reinterpret_cast<const u64 *>(d_cb), reinterpret_cast<const u64 *>(d_umi), d_gene, d_aux, n64, n_parts,
		                      reinterpret_cast<u64 *>(d_out_cb), reinterpret_cast<u64 *>(d_out_umi), d_out_gene, d_out_aux, d_out_idx, counts, d_scratch, scratch_bytes);
	});
}

}  // extern "C"

static void partition_by_owner_on(int device, hipStream_t st, const u64 *d_cb, const u64 *d_umi, const u32 *d_gene, const u32 *d_aux, uint64_t n64,
                                  u32 n_parts, u64 *d_out_cb, u64 *d_out_umi, u32 *d_out_gene, u32 *d_out_aux, u32 *d_out_idx, uint64_t *counts,
                                  void *d_scratch, uint64_t scratch_bytes) {
	{
		if (n_parts == 0 || n_parts > 256) throw InvalidError("n_parts must be in 1..256");
		if (n64 >= 0xFFFFFFFEull) throw UnsupportedError("more than 2^32-2 reads per GPU");
		HIP_CHECK(hipSetDevice(device));
		const u32 n = u32(n64);
		for (u32 p = 0; p < n_parts; ++p) counts[p] = 0;
		if (n == 0) return;
		// owner histogram per workgroup, scans, then ONE pass that writes every read to its place (k_misc.h)
		u32 nblocks, tpb; size_t off_k1, off_hist, off_row, off_base, total;
		partition_plan(n, nblocks, tpb, off_k1, off_hist, off_row, off_base, total);
		if (!d_scratch || scratch_bytes < total) throw InvalidError("partition scratch too small (dropest_partition_scratch_bytes)");
		char *base = static_cast<char *>(d_scratch);
		u32 *hist = reinterpret_cast<u32 *>(base + off_hist), *row_total = reinterpret_cast<u32 *>(base + off_row);
		u32 *digit_base = reinterpret_cast<u32 *>(base + off_base);
		const int owner_bits = std::max(1, bit_length(uint64_t(n_parts - 1)));
		hipLaunchKernelGGL(owner_hist_kernel, dim3(nblocks), dim3(OP_T), 0, st, d_cb, n, n_parts, tpb, hist);
		hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(RS_RADIX), dim3(256), 0, st, hist, nblocks, row_total);
		hipLaunchKernelGGL(rs_scan_totals_kernel<256>, dim3(1), dim3(256), 0, st, row_total, digit_base);
		hipLaunchKernelGGL(owner_scatter_kernel<false>, dim3(nblocks), dim3(OP_T), 0, st, d_cb, d_umi, d_gene, d_aux, n, n_parts, owner_bits, tpb, hist, digit_base,
		                   d_out_cb, d_out_umi, d_out_gene, d_out_aux, d_out_idx, ExchangePack{});
		HIP_CHECK(hipGetLastError());
		std::vector<u32> totals(RS_RADIX);
		HIP_CHECK(hipMemcpyAsync(totals.data(), row_total, RS_RADIX * 4, hipMemcpyDeviceToHost, st));
		HIP_CHECK(stream_wait(st));
		for (u32 p = 0; p < n_parts; ++p) counts[p] = totals[p];
	}
}

extern "C" {

dropest_status dropest_real_candidate_rows(dropest_ctx *ctx, uint64_t *n, uint64_t *ids, dropest_cell_row *rows) {
	return guarded([&] {
		need_init(ctx);
		*n = ctx->real.size();
		if (!ids || !rows) return;
		for (size_t i = 0; i < ctx->real.size(); ++i) {
			const HostCell &h = ctx->real[i];
			ids[i] = h.id;
			dropest_cell_row &r = rows[i];
			std::memcpy(&r, &h.row, sizeof(r));
			r.is_merged = h.merged; r.is_excluded = h.excluded;
			r.is_real = !h.merged && !h.excluded && h.row.n_genes >= ctx->min_before;
		}
	});
}

dropest_status dropest_dev_copy_device(int device, void *d_dst, const void *d_src, uint64_t bytes) {
	return guarded([&] {
		HIP_CHECK(hipSetDevice(device));
		if (bytes) HIP_CHECK(hipMemcpy(d_dst, d_src, bytes, hipMemcpyDeviceToDevice));
	});
}

// The device arrays of the reads pushed so far (dropest_push_reads), for a second context that looks at the same reads in
// place (dropest_push_reads_device(..., adopt = 1)): the facade's view of a container that is not initialised yet.  Valid
// until the next push.
dropest_status dropest_resident_reads(dropest_ctx *ctx, const uint64_t **d_cb, const uint64_t **d_umi, const uint32_t **d_gene,
                                      const uint32_t **d_aux, uint64_t *n) {
	return guarded([&] {
		if (!ctx || !d_cb || !d_umi || !d_gene || !d_aux || !n) throw InvalidError("null argument");
		HIP_CHECK(hipSetDevice(ctx->cfg.device));
		*n = ctx->n_reads;
		*d_cb = *d_umi = nullptr; *d_gene = *d_aux = nullptr;
		if (ctx->n_reads == 0) return;
		ctx->need_columns();
		if (ctx->d_cb) { *d_cb = reinterpret_cast<const uint64_t *>(ctx->d_cb); *d_umi = reinterpret_cast<const uint64_t *>(ctx->d_umi); *d_gene = ctx->d_gene; *d_aux = ctx->d_aux; return; }
		if (ctx->chunks.size() != 1 || ctx->store_chunk != 0) throw UnsupportedError("the reads are not one pushed block (device chunks were adopted in between)");
		ctx->store.wait();
		*d_cb = reinterpret_cast<const uint64_t *>(ctx->store.cb.p); *d_umi = reinterpret_cast<const uint64_t *>(ctx->store.umi.p);
		*d_gene = ctx->store.gene.p; *d_aux = ctx->store.aux.p;
	});
}

dropest_status dropest_clear_reads(dropest_ctx *ctx) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		ctx->free_results();
		ctx->chunks.clear();
		ctx->store.clear(); ctx->store_chunk = -1;
		ctx->n_reads = 0;
		ctx->d_cb = ctx->d_umi = nullptr; ctx->d_gene = ctx->d_aux = nullptr;
		ctx->have_qual = ctx->qual_var = false; ctx->qual_len = 0; ctx->qual_reads = 0;   // the qualities belonged to those reads
	});
}

dropest_status dropest_count_matrix_device(dropest_ctx *ctx, int filtered, int reads_output, uint64_t *ncols, uint64_t *nnz,
                                           const uint32_t **colptr, const uint32_t **d_rowidx, const uint32_t **d_values) {
	return guarded([&] {
		need_init(ctx);
		ctx->emit_matrix(filtered != 0, reads_output != 0, /*to_host=*/false);
		const dropest_ctx::MatrixResult &M = ctx->mat[filtered ? 0 : 1];
		*ncols = M.ncols; *nnz = M.nnz;
		*colptr = M.colptr.data(); *d_rowidx = M.d_row.p; *d_values = M.d_val.p;
	});
}

dropest_status dropest_cell_first_reads_device(dropest_ctx *ctx, uint64_t *n_cells, const uint32_t **d_first) {
	return guarded([&] { need_init(ctx); *n_cells = ctx->n_cells; *d_first = ctx->cell_first.p; });
}

static void assemble_columns_launch(int device, int slot, uint64_t n_cols, const uint64_t *src_start, const uint64_t *dst_start,
                                    const uint64_t *len, const uint32_t *d_src_rows, const uint32_t *d_src_vals,
                                    uint32_t *d_dst_rows, uint32_t *d_dst_vals) {
	HIP_CHECK(hipSetDevice(device));
	if (n_cols == 0) return;
	// descriptor scratch: two slots per device, kept for the life of the library (this runs once per matrix and step;
	// a slot must not be reused before the launch that reads it has been waited for)
	static DevBuf<u64> desc_cache[64][2];
	static PinnedBuf<u64> desc_host[64][2];
	if (device < 0 || device >= 64) throw InvalidError("device index out of range");
	if (slot < 0 || slot > 1) throw InvalidError("descriptor slot must be 0 or 1");
	DevBuf<u64> &d_desc = desc_cache[device][slot];
	PinnedBuf<u64> &desc = desc_host[device][slot];
	d_desc.ensure(n_cols * 3 + n_cols / 2); desc.ensure(n_cols * 3);
	for (uint64_t c = 0; c < n_cols; ++c) { desc.p[3 * c] = src_start[c]; desc.p[3 * c + 1] = dst_start[c]; desc.p[3 * c + 2] = len[c]; }
	HIP_CHECK(hipMemcpyAsync(d_desc.p, desc.p, n_cols * 3 * 8, hipMemcpyHostToDevice, nullptr));
	hipLaunchKernelGGL(assemble_columns_kernel, dim3(u32(n_cols)), dim3(256), 0, nullptr, d_desc.p, d_src_rows, d_src_vals,
	                   d_dst_rows, d_dst_vals);
	HIP_CHECK(hipGetLastError());
}

dropest_status dropest_assemble_columns(int device, uint64_t n_cols, const uint64_t *src_start, const uint64_t *dst_start,
                                        const uint64_t *len, const uint32_t *d_src_rows, const uint32_t *d_src_vals,
                                        uint32_t *d_dst_rows, uint32_t *d_dst_vals) {
	return guarded([&] {
		assemble_columns_launch(device, 0, n_cols, src_start, dst_start, len, d_src_rows, d_src_vals, d_dst_rows, d_dst_vals);
		HIP_CHECK(hipDeviceSynchronize());
	});
}

dropest_status dropest_assemble_columns_async(int device, int slot, uint64_t n_cols, const uint64_t *src_start, const uint64_t *dst_start,
                                              const uint64_t *len, const uint32_t *d_src_rows, const uint32_t *d_src_vals,
                                              uint32_t *d_dst_rows, uint32_t *d_dst_vals) {
	return guarded([&] {
		assemble_columns_launch(device, slot, n_cols, src_start, dst_start, len, d_src_rows, d_src_vals, d_dst_rows, d_dst_vals);
	});
}

dropest_status dropest_host_register(int device, void *host, uint64_t bytes, void **d_ptr) {
	return guarded([&] {
		if (!host || !d_ptr) throw InvalidError("null argument");
		HIP_CHECK(hipSetDevice(device));
		HIP_CHECK(hipHostRegister(host, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
		HIP_CHECK(hipHostGetDevicePointer(d_ptr, host, 0));
	});
}

dropest_status dropest_host_unregister(int device, void *host) {
	return guarded([&] {
		HIP_CHECK(hipSetDevice(device));
		HIP_CHECK(hipHostUnregister(host));
	});
}

dropest_status dropest_kernel_stats(dropest_ctx *ctx, uint32_t *n, dropest_kernel_stat *out) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		ctx->collect_timings();
		u32 i = 0;
		for (auto &kv : ctx->stats) {
			if (out) { out[i].name = kv.first.c_str(); out[i].launches = kv.second.launches; out[i].ms = kv.second.ms; out[i].bytes = kv.second.bytes; }
			++i;
		}
		*n = i;
	});
}

dropest_status dropest_set_profiling(dropest_ctx *ctx, int enabled) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		ctx->collect_timings();
		ctx->profiling = enabled != 0;
		if (enabled) ctx->stats.clear();
	});
}

dropest_status dropest_radix_plan(uint64_t varying_mask, uint32_t *n_passes, int32_t *shifts, int32_t *bits) {
	return guarded([&] {
		if (!n_passes) throw InvalidError("null argument");
		const std::vector<RadixPass> plan = plan_radix_passes(varying_mask);
		*n_passes = u32(plan.size());
		for (size_t i = 0; i < plan.size(); ++i) { if (shifts) shifts[i] = plan[i].shift; if (bits) bits[i] = plan[i].bits; }
	});
}

dropest_status dropest_table_sizes(dropest_ctx *ctx, uint64_t out[4]) {
	return guarded([&] { need_init(ctx); out[0] = ctx->n_reads; out[1] = ctx->n_cells; out[2] = ctx->n_mol; out[3] = ctx->n_cg; });
}

dropest_status dropest_rand_sequence(uint32_t seed, uint64_t n, int32_t *out) {
	return guarded([&] {
		if (n && !out) throw InvalidError("null argument");
		GlibcRand r(seed);
		for (uint64_t i = 0; i < n; ++i) out[i] = r.next();
	});
}

dropest_status dropest_debug_poison_scratch(uint64_t seed, uint64_t *n_blocks) {
	return guarded([&] {
		HIP_CHECK(hipDeviceSynchronize());
		const size_t d = DevRegistry::get().poison_all(seed), h = PinnedRegistry::get().poison_all(seed);
		if (n_blocks) *n_blocks = d + h;
	});
}

dropest_status dropest_debug_alloc_ordinal(uint64_t *next_ordinal) {
	return guarded([&] { auto &r = DevRegistry::get(); std::lock_guard<std::mutex> lk(r.mu); *next_ordinal = r.next; });
}

dropest_status dropest_debug_alloc_site(uint64_t ordinal, char *out, uint64_t out_bytes) {
	return guarded([&] {
		auto &r = DevRegistry::get();
		std::lock_guard<std::mutex> lk(r.mu);
		std::string text = "unknown (set DROPEST_ALLOC_TRACE=1)";
		for (auto const &s : r.sites) if (s.ordinal == ordinal) text = std::string(s.file) + ":" + std::to_string(s.line) + " " + std::to_string(s.bytes) + " B" + (s.recycled ? " recycled" : " fresh");
		if (out && out_bytes) { std::snprintf(out, size_t(out_bytes), "%s", text.c_str()); }
	});
}

dropest_status dropest_debug_refresh(void) {
	return guarded([&] { dropest::DevDebug::refresh(); });
}

dropest_status dropest_debug_trim_pool(void) {
	return guarded([&] { DevRegistry::get().trim_pool(); });
}

dropest_status dropest_set_profiling_filter(dropest_ctx *ctx, const char *name_prefix) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		ctx->collect_timings();
		ctx->profile_only = name_prefix ? name_prefix : "";
	});
}

void *dropest_stream(dropest_ctx *ctx) { return ctx ? ctx->stream : nullptr; }

}  // extern "C"

#include "shard_run.h"
