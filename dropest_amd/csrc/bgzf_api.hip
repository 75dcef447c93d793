// bgzf_api.hip -- include/dropest_bgzf.h: BGZF block scan on the host, DEFLATE on the device (k_inflate.h).
#include "../../include/dropest_bgzf.h"
#include "../../include/dropest_annotation.h"
#include "k_inflate.h"
#include "k_inflate_par.h"
#include "k_bamparse.h"
#include "util.h"

#include <atomic>
#include <chrono>
#include <future>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

using namespace dropest;

namespace {
thread_local std::string g_bgzf_error;
inline uint32_t le16(const uint8_t *p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8); }
inline uint32_t le32(const uint8_t *p) { return le16(p) | (le16(p + 2) << 16); }
template <class F> int bgzf_guarded(F &&f) {
	try { f(); return 0; }
	catch (const std::exception &e) { g_bgzf_error = e.what(); return 1; }
}
}  // namespace

extern "C" const char *dropest_bgzf_last_error(void) { return g_bgzf_error.c_str(); }

extern "C" int dropest_bgzf_scan(const uint8_t *data, uint64_t len, uint64_t cap, uint64_t *in_off, uint32_t *in_len, uint64_t *out_off,
                                 uint32_t *out_len, uint32_t *crc32, uint64_t *n_blocks, uint64_t *bytes_used, uint64_t *out_total) {
	return bgzf_guarded([&] {
		if (!data || !in_off || !in_len || !out_off || !out_len || !n_blocks) throw InvalidError("null argument");
		uint64_t at = 0, n = 0, total = 0;
		while (n < cap && at + 18 <= len) {
			const uint8_t *h = data + at;
			// gzip member with FEXTRA (SAMv1 4.1): 1f 8b 08 04, XLEN at 10, subfields of (SI1, SI2, SLEN, data); 'B' 'C' 2 holds BSIZE
			if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) throw InvalidError("not a BGZF block header at offset " + std::to_string(at));
			const uint32_t xlen = le16(h + 10);
			if (at + 12 + xlen > len) break;
			uint32_t bsize = 0;
			for (uint32_t x = 0; x + 4 <= xlen;) {
				const uint8_t *sf = h + 12 + x;
				const uint32_t slen = le16(sf + 2);
				if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = le16(sf + 4) + 1;
				x += 4 + slen;
			}
			if (!bsize || bsize < 12 + xlen + 8) throw InvalidError("BGZF block without a BC subfield at offset " + std::to_string(at));
			if (at + bsize > len) break;
			in_off[n] = at + 12 + xlen; in_len[n] = bsize - 12 - xlen - 8;
			out_off[n] = total; out_len[n] = le32(h + bsize - 4);
			if (out_len[n] > 65536u) throw InvalidError("BGZF block with ISIZE beyond 64 KB at offset " + std::to_string(at));
			if (crc32) crc32[n] = le32(h + bsize - 8);
			total += out_len[n];
			at += bsize; ++n;
		}
		*n_blocks = n;
		if (bytes_used) *bytes_used = at;
		if (out_total) *out_total = total;
	});
}

extern "C" int dropest_bgzf_inflate_device(int device, void *stream, const uint8_t *d_in, uint64_t in_total, const uint64_t *d_in_off,
                                           const uint32_t *d_in_len, const uint64_t *d_out_off, const uint32_t *d_out_len, uint32_t n_blocks,
                                           uint8_t *d_out, uint32_t *d_status, const uint32_t *d_crc32) {
	return bgzf_guarded([&] {
		if (!n_blocks) return;
		if (!d_in || !d_in_off || !d_in_len || !d_out_off || !d_out_len || !d_out || !d_status) throw InvalidError("null argument");
		if (uintptr_t(d_in) & 7u) throw InvalidError("the compressed bytes must be 8-byte aligned");
		HIP_CHECK(hipSetDevice(device));
		// DROPEST_INFLATE_PAR=0: one chain of symbols per block (k_inflate.h: 131 GB/s on a file that deflates 10.8 x, 52 on one that deflates 3.2 x like
		// a real 10x BAM); default: the lanes of a wave on different chunks of the block's symbols (k_inflate_par.h: 123 / 94 GB/s, and a block lasts
		// 0.7 ms instead of 7-11, which is what a window of the BAM path waits for)
		static const int par = [] { const char *e = getenv("DROPEST_INFLATE_PAR"); return e ? atoi(e) : 1; }();
		if (par) {
			// the lanes of a wave on different parts of a block's symbol stream (k_inflate_par.h); the waves take blocks from a counter and keep their
			// match lists in a scratch buffer of the (device, stream) they run on
			struct Scratch { DevBuf<InfpMatch> list; DevBuf<uint32_t> next; uint32_t grid = 0; };
			static std::mutex mu;
			static std::map<std::pair<int, void *>, std::unique_ptr<Scratch>> pool;
			Scratch *sc = nullptr;
			{
				std::lock_guard<std::mutex> lk(mu);
				auto &slot = pool[{device, stream}];
				if (!slot) {
					slot.reset(new Scratch());
					hipDeviceProp_t prop{};
					HIP_CHECK(hipGetDeviceProperties(&prop, device));
					slot->grid = uint32_t(prop.multiProcessorCount) * 4u;
					slot->list.alloc(size_t(slot->grid) * INFP_WAVES * INFP_MATCH_CAP);
					slot->next.alloc(1);
				}
				sc = slot.get();
			}
			HIP_CHECK(hipMemsetAsync(sc->next.p, 0, 4, hipStream_t(stream)));
			// (DROPEST_INFLATE_PAR_WGS_PER_CU=1..3: fewer workgroups than the CUs hold, so that kernels of other streams find wave slots and LDS beside this one)
			static const uint32_t wgs_per_cu = [] { const char *e = getenv("DROPEST_INFLATE_PAR_WGS_PER_CU"); const int v = e ? atoi(e) : 4; return uint32_t(v < 1 ? 1 : v > 4 ? 4 : v); }();
			const uint32_t grid = std::min<uint32_t>(sc->grid / 4u * wgs_per_cu, (n_blocks + INFP_WAVES - 1) / INFP_WAVES);
			hipLaunchKernelGGL(bgzf_inflate_par_kernel, dim3(grid), dim3(INFP_WAVES * 64), 0, hipStream_t(stream), d_in, in_total, d_in_off, d_in_len, d_out_off, d_out_len,
			                   n_blocks, d_out, d_status, d_crc32, sc->list.p, sc->next.p, uint32_t(getenv("DROPEST_INFLATE_PAR_DBG") ? atoi(getenv("DROPEST_INFLATE_PAR_DBG")) : 0));
			HIP_CHECK(hipGetLastError());
			return;
		}
		hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((n_blocks + INF_WAVES - 1) / INF_WAVES), dim3(INF_WAVES * 64), 0, hipStream_t(stream), d_in, in_total,
		                   d_in_off, d_in_len, d_out_off, d_out_len, n_blocks, d_out, d_status, d_crc32);
		HIP_CHECK(hipGetLastError());
	});
}

#ifdef INFP_PROFILE
// (variant builds of scripts/experiments/inflate_variants only: the kernel's own account of its phases, read and cleared)
extern "C" int dropest_bgzf_inflate_profile(unsigned long long *out24) {
	if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(dropest::infp_prof), 24 * sizeof(unsigned long long)) != hipSuccess) return 1;
	unsigned long long zero[24] = {};
	return hipMemcpyToSymbol(HIP_SYMBOL(dropest::infp_prof), zero, sizeof(zero)) != hipSuccess;
}
#endif

extern "C" int dropest_bgzf_inflate_buffer(int device, const uint8_t *data, uint64_t len, uint8_t *out, uint64_t out_cap, uint64_t *out_len,
                                           uint32_t *status, uint64_t status_cap, uint64_t *n_blocks, double *kernel_ms, int repeats) {
	const bool check_crc = repeats >= 0;     // (repeats < 0: |repeats| runs without the CRC-32 check -- what the check costs)
	if (repeats < 0) repeats = -repeats;
	return bgzf_guarded([&] {
		if (!data || !out_len || !n_blocks) throw InvalidError("null argument");
		int n_dev = 0;
		if (hipGetDeviceCount(&n_dev) != hipSuccess || device >= n_dev) throw DeviceError("no such GPU: BGZF blocks are inflated on the device only here");
		HIP_CHECK(hipSetDevice(device));
		const uint64_t cap = len / 26 + 1;
		std::vector<uint64_t> in_off(cap), out_off(cap);
		std::vector<uint32_t> in_len(cap), o_len(cap), crc(cap);
		uint64_t n = 0, used = 0, total = 0;
		if (dropest_bgzf_scan(data, len, cap, in_off.data(), in_len.data(), out_off.data(), o_len.data(), crc.data(), &n, &used, &total)) throw InvalidError(g_bgzf_error);
		*n_blocks = n; *out_len = total;
		if (total > out_cap) throw InvalidError("output buffer too small: " + std::to_string(total) + " bytes needed");
		if (!n) return;
		if (n > 0xFFFFFFFFull) throw UnsupportedError("more than 2^32 blocks in one call");
		DevBuf<uint8_t> d_in, d_out;
		DevBuf<uint64_t> d_in_off, d_out_off;
		DevBuf<uint32_t> d_in_len, d_out_len, d_status, d_crc;
		d_in.alloc(used + 8); d_out.alloc(total + 8); d_in_off.alloc(n); d_out_off.alloc(n); d_in_len.alloc(n); d_out_len.alloc(n); d_status.alloc(n); d_crc.alloc(n);
		HIP_CHECK(hipMemcpy(d_crc.p, crc.data(), n * 4, hipMemcpyHostToDevice));
		HIP_CHECK(hipMemcpy(d_in.p, data, used, hipMemcpyHostToDevice));
		HIP_CHECK(hipMemcpy(d_in_off.p, in_off.data(), n * 8, hipMemcpyHostToDevice));
		HIP_CHECK(hipMemcpy(d_out_off.p, out_off.data(), n * 8, hipMemcpyHostToDevice));
		HIP_CHECK(hipMemcpy(d_in_len.p, in_len.data(), n * 4, hipMemcpyHostToDevice));
		HIP_CHECK(hipMemcpy(d_out_len.p, o_len.data(), n * 4, hipMemcpyHostToDevice));
		HIP_CHECK(hipMemset(d_status.p, 0xFF, n * 4));
		HIP_CHECK(hipMemset(d_out.p, 0, total + 8));
		hipEvent_t e0, e1;
		HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
		double ms_sum = 0;
		const int reps = repeats > 0 ? repeats : 1;
		for (int r = 0; r < reps; ++r) {
			HIP_CHECK(hipEventRecord(e0, nullptr));
			if (dropest_bgzf_inflate_device(device, nullptr, d_in.p, used, d_in_off.p, d_in_len.p, d_out_off.p, d_out_len.p, uint32_t(n), d_out.p, d_status.p, check_crc ? d_crc.p : nullptr))
				throw DeviceError(g_bgzf_error);
			HIP_CHECK(hipEventRecord(e1, nullptr));
			HIP_CHECK(hipEventSynchronize(e1));
			float ms = 0;
			HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
			ms_sum += ms;
		}
		(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
		if (kernel_ms) *kernel_ms = ms_sum / reps;
		if (out) HIP_CHECK(hipMemcpy(out, d_out.p, total, hipMemcpyDeviceToHost));
		if (status) HIP_CHECK(hipMemcpy(status, d_status.p, std::min<uint64_t>(n, status_cap) * 4, hipMemcpyDeviceToHost));
	});
}

// ---- BAM records of a window, on the device ------------------------------------------------------------------------------------
// tests: wrong guesses on purpose (every third segment one byte late, every seventh none at all) -- the host's check must find the true chain anyway
__global__ __launch_bounds__(256) void bam_spoil_guesses_kernel(uint64_t *__restrict__ seg_start, uint32_t n_segs) {
	const uint32_t k = blockIdx.x * 256 + threadIdx.x;
	if (k == 0 || k >= n_segs) return;
	if (k % 7u == 3u) seg_start[k] = BAM_NONE;
	else if (k % 3u == 1u && seg_start[k] != BAM_NONE) seg_start[k] += 1;
}

// What the first half of a window produces (copy in, inflate, the chain of records) and the second half (fields, dense columns) consumes.  Two of
// them: the caller may run the first half of window k + 1 on another thread while the second half of window k and its own work go on.
struct BamFront {
	hipStream_t stream = nullptr;
	DevBuf<uint8_t> d_in, d_out;
	DevBuf<uint64_t> d_in_off, d_out_off, seg_start, seg_exit;
	DevBuf<uint32_t> d_in_len, d_out_len, d_status, d_crc, seg_count, seg_base, d_bad, d_list;
	PinnedBuf<uint64_t> h_seg_start, h_seg_exit;
	PinnedBuf<uint32_t> h_count, h_block_status, h_base;
	PinnedBuf<uint64_t> h_tab;      // the window's block table on its way to the device: in_off | out_off | in_len, out_len, crc (28 bytes a block)
	std::vector<uint64_t> in_off, out_off;
	std::vector<uint32_t> in_len, out_len, crc;
	uint64_t data_len = 0, tail_start = 0, n_rec = 0;
	uint32_t n_segs = 0, n_blocks = 0, refused = 0, repaired = 0;
	double ms_copy = 0, ms_inflate = 0, ms_boundaries = 0;
	bool begun = false;
	// the window's data stand at d_out + base_off: [the record the window before cut off | the inflated blocks].  The blocks are inflated to
	// d_out + reserve BEFORE the window before has said how long its cut-off record is (dropest_bam_decoder_window_inflate: the inflate of window
	// k + 1 runs beside the chain / parse / host work of window k); the record is put in front of them afterwards (.._window_chain).
	uint64_t base_off = 0, reserve = 0, total = 0, comp_len = 0;
	const uint8_t *comp = nullptr;
	bool inflating = false;
	uint8_t *data() { return d_out.p + base_off; }
};

struct dropest_bam_decoder {
	int device = 0;
	hipStream_t stream = nullptr;
	BamParseCfg cfg{};
	BamFront front[2];
	int next_front = 0, last_front = 0;
	bool traced_first = false;
	bool halves_in_sequence = false;   // dropest_bam_decoder_window is running: its first half uses `stream`
	DevBuf<uint8_t> d_tail, d_gather;
	DevBuf<uint64_t> rec_off, d_goff;
	DevBuf<uint32_t> d_gidx, d_gsize;
	DevBuf<unsigned long long> o_cb, o_umi, dn_cb, dn_umi, p_cb, p_umi, g_keys, o_qoff, dn_qoff;
	DevBuf<uint8_t> dn_qual;
	PinnedBuf<uint8_t> h_qual;
	DevBuf<uint32_t> o_gene, o_aux, dn_gene, dn_aux, tile_ok, tile_need, d_totals, nd_rec, nd_pos, nd_size, p_pos, p_gene, p_aux, g_vals;
	DevBuf<int32_t> d_chr, d_ann_chr, d_ann_id, a_chr, a_mark;
	DevBuf<uint32_t> a_pos, a_end, a_gene;
	dropest_annotation *annotation = nullptr;   // -g: not owned
	uint32_t n_ann_genes = 0;
	DevBuf<uint16_t> o_uql;
	DevBuf<uint8_t> o_status, o_need;
	DevBuf<BamWindowCounts> d_wc;
	PinnedBuf<uint8_t> h_stage[2];
	PinnedBuf<uint32_t> h_need_rec, h_need_pos, h_need_size, h_gsize, h_gidx, h_result;
	PinnedBuf<uint64_t> h_goff, h_patch;
	PinnedBuf<uint8_t> h_gather, h_dict;      // h_result: a window's counters, totals and flag (13 words)
	uint32_t g_mask = 0;
	DevBuf<uint32_t> g_name_off;      // the dictionary's gene names by index (dropest_bam_decoder_set_gene_names), 0 names: hashes alone
	DevBuf<uint8_t> g_name_pool;
	uint32_t n_gene_names = 0;
	unsigned long long hash_mask = ~0ull;
	uint64_t tail_len = 0, last_n_rec = 0, last_n_ok = 0;
	// the compressed bytes of a staging buffer on their way to the device ahead of the window call (dropest_bam_decoder_upload)
	// Two streams: `stream` for the kernels and the small copies, `up_stream` for the compressed bytes on their way in under the kernels of the
	// window before.  A stream is 8-20 ms to create (and its first dispatch as much again), and every other HIP call of the process waits
	// meanwhile -- measured three ways in round 6 (created up front: 16 ms; on a helper thread beside the pinned allocation: the allocation runs,
	// hipMalloc and the reader's first copies wait, nothing gained; NOTES_r06 section 7).  So the decoder creates none unless it has to:
	// `stream` is LENT by the caller for a file (dropest_bam_decoder_use_stream: the container's own, which exists and has run kernels; a
	// decoder that is lent none makes one when first needed), and the uploads take the device's null stream, which the non-blocking streams
	// do not wait for.
	hipStream_t up_stream = nullptr;
	bool up_begun[2] = {false, false};
	hipStream_t own_stream = nullptr;      // `stream` is this one, or the caller's
	bool tables_fresh = true;              // the empty dictionaries' memsets have not been queued yet (they wait for `stream`)
	std::mutex ready_mutex;
	// pinned pieces (dropest_bam_decoder_pieces): the compressed bytes of a window go to the device piece by piece, so that the pinned memory
	// (0.15-0.4 ms per MB to allocate) does not grow with the window
	PinnedBuf<uint8_t> piece_mem;      // one allocation (each has a fixed cost of a few ms), cut into the pieces
	uint32_t n_pieces = 0;
	uint64_t piece_bytes = 0;
	std::vector<hipEvent_t> piece_done;
	std::atomic<const uint8_t *> up_host[2] = {{nullptr}, {nullptr}};   // the host bytes that up_in[w] holds (or will, once up_done[w] has passed)
	hipEvent_t up_done[2] = {nullptr, nullptr};
	DevBuf<uint8_t> up_in[2];
	uint64_t up_len[2] = {0, 0};
	std::atomic<bool> up_ready[2] = {{false}, {false}};   // written by the reader thread (dropest_bam_decoder_upload), read by the window call: release / acquire around up_len / up_blocks
	// ... and their block table, made by the same caller (a walk from header to header is one cache miss per block: ~2 ms per 64 MB window)
	struct UpBlocks { std::vector<uint64_t> in_off, out_off; std::vector<uint32_t> in_len, out_len, crc; uint64_t n = 0, used = 0, total = 0; bool ok = false; } up_blocks[2];
};

// The decoder's streams exist from here on (the helper thread of dropest_bam_decoder_create is joined by the first call that needs them)
static void bam_ready(dropest_bam_decoder *d) {
	std::lock_guard<std::mutex> lk(d->ready_mutex);
	if (!d->stream) {
		if (!d->own_stream) HIP_CHECK(hipStreamCreateWithFlags(&d->own_stream, hipStreamNonBlocking));
		d->stream = d->own_stream;
	}
	if (d->tables_fresh) {
		HIP_CHECK(hipMemsetAsync(d->g_vals.p, 0, size_t(d->g_mask + 1) * 4, d->stream));
		HIP_CHECK(hipMemsetAsync(d->d_chr.p, 0xFF, d->d_chr.n * 4, d->stream));
		d->tables_fresh = false;
	}
}

// comp[0 .. len) walked from block header to block header: the arrays the inflate kernel and the host fall-back take.  false: not whole, sound blocks
// (the caller says what is wrong with dropest_bgzf_scan's message)
static bool bgzf_block_table(const uint8_t *comp, uint64_t len, std::vector<uint64_t> &in_off, std::vector<uint64_t> &out_off, std::vector<uint32_t> &in_len,
                             std::vector<uint32_t> &out_len, std::vector<uint32_t> &crc, uint64_t *n, uint64_t *used, uint64_t *total) {
	// counted first: arrays for the smallest possible block, 26 bytes, would be 36 MB of page faults per 32 MB window
	uint64_t cap = 1;
	for (uint64_t at = 0; at + 18 <= len; ++cap) {
		const uint8_t *h = comp + at;
		if (h[0] != 0x1f || h[1] != 0x8b) break;                      // (dropest_bgzf_scan below says what is wrong)
		const uint32_t xlen = le16(h + 10);
		uint32_t bsize = 0;
		for (uint32_t x = 0; x + 4 <= xlen && at + 12 + x + 6 <= len;) { const uint8_t *sf = h + 12 + x; const uint32_t sl = le16(sf + 2); if (sf[0] == 'B' && sf[1] == 'C' && sl == 2) bsize = le16(sf + 4) + 1; x += 4 + sl; }
		if (!bsize) break;
		at += bsize;
	}
	if (in_off.size() < cap) { const uint64_t c2 = cap + cap / 2; in_off.resize(c2); out_off.resize(c2); in_len.resize(c2); out_len.resize(c2); crc.resize(c2); }
	*n = *used = *total = 0;
	return !len || dropest_bgzf_scan(comp, len, cap, in_off.data(), in_len.data(), out_off.data(), out_len.data(), crc.data(), n, used, total) == 0;
}

// This translation unit's kernels are loaded onto the device by the first launch of one of them (~10 ms): a context does it when it is created
// (dropest_amd.hip), with the other start-up costs, so that it does not fall into the first window of the first BAM file.
extern "C" void dropest_bgzf_warm_up(void *stream) {
	hipLaunchKernelGGL(bam_spoil_guesses_kernel, dim3(1), dim3(256), 0, hipStream_t(stream), (uint64_t *)nullptr, 0u);      // (no segment: nothing is touched)
	(void)hipGetLastError();
}

extern "C" int dropest_bam_decoder_create(int device, const dropest_bam_parse_cfg *cfg, dropest_bam_decoder **out) {
	return bgzf_guarded([&] {
		if (!cfg || !out) throw InvalidError("null argument");
		int n_dev = 0;
		if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) throw DeviceError("no such GPU: the BAM decoder has no CPU implementation");
		HIP_CHECK(hipSetDevice(device));
		static_assert(sizeof(dropest_bam_parse_cfg) == sizeof(BamParseCfg), "the C struct and the kernels' struct are one layout");
		if (cfg->intronic_len > 24 || cfg->intergenic_len > 24) throw InvalidError("read-type values longer than 24 characters");
		if (cfg->n_refs < 0) throw InvalidError("negative number of references");
		auto *d = new dropest_bam_decoder();
		d->device = device;
		std::memcpy(&d->cfg, cfg, sizeof(BamParseCfg));
		if (const char *e = getenv("DROPEST_BAM_TEST_GENE_HASH_BITS")) { const int b = atoi(e); if (b > 0 && b < 64) d->hash_mask = (1ull << b) - 1ull; }   // (tests: names that collide)
		try {
			using clk = std::chrono::steady_clock;
			const bool trace = getenv("DROPEST_BAM_TRACE") != nullptr;
			auto t0 = clk::now();
			auto lap = [&](const char *what) { if (trace) { std::fprintf(stderr, "[bam] decoder: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(clk::now() - t0).count()); t0 = clk::now(); } };
			// empty dictionaries: every gene and chromosome is new
			// (room for 32 000 genes and their names from the start: growing these tables between two windows is a hipFree, a hipMalloc and a pinned
			// reallocation -- ~10 ms of "dictionaries to the device" in the window that first knows a few thousand genes)
			d->g_mask = 1023; d->g_keys.alloc(size_t(1) << 16); d->g_vals.alloc(size_t(1) << 16); d->d_chr.alloc(size_t(std::max(1, cfg->n_refs)));
			d->g_name_off.ensure(size_t(1) << 15); d->g_name_pool.ensure(size_t(1) << 20); d->h_dict.ensure(size_t(2) << 20);
			lap("dictionary buffers");
			for (hipEvent_t &e : d->up_done) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
			lap("events");
		} catch (...) { delete d; throw; }
		*out = d;
	});
}

// A decoder taken up again for another file (its buffers, streams and pinned memory stay): the parse configuration of that file, empty
// dictionaries, no annotation, no record carried over.
extern "C" int dropest_bam_decoder_reset(dropest_bam_decoder *d, const dropest_bam_parse_cfg *cfg) {
	return bgzf_guarded([&] {
		if (!d || !cfg) throw InvalidError("null argument");
		if (cfg->intronic_len > 24 || cfg->intergenic_len > 24) throw InvalidError("read-type values longer than 24 characters");
		if (cfg->n_refs < 0) throw InvalidError("negative number of references");
		HIP_CHECK(hipSetDevice(d->device));
		bam_ready(d);
		HIP_CHECK(hipStreamSynchronize(d->stream));
		for (BamFront &f : d->front) { if (f.stream) HIP_CHECK(hipStreamSynchronize(f.stream)); f.begun = false; f.inflating = false; }
		HIP_CHECK(hipStreamSynchronize(d->up_stream));
		d->up_ready[0].store(false, std::memory_order_relaxed); d->up_ready[1].store(false, std::memory_order_relaxed);
		d->up_host[0].store(nullptr, std::memory_order_relaxed); d->up_host[1].store(nullptr, std::memory_order_relaxed);
		std::memcpy(&d->cfg, cfg, sizeof(BamParseCfg));
		d->tail_len = 0; d->last_n_rec = 0; d->last_n_ok = 0; d->next_front = 0; d->last_front = 0;
		d->annotation = nullptr; d->n_ann_genes = 0; d->n_gene_names = 0;
		d->d_chr.ensure(size_t(std::max(1, cfg->n_refs)));
		HIP_CHECK(hipMemsetAsync(d->g_vals.p, 0, size_t(d->g_mask + 1) * 4, d->stream));
		HIP_CHECK(hipMemsetAsync(d->d_chr.p, 0xFF, d->d_chr.n * 4, d->stream));
		HIP_CHECK(hipStreamSynchronize(d->stream));
	});
}

// BGZF blocks the device inflates at once (a wave each, INF_WAVES_PER_EU per SIMD): a window of that many blocks takes as long as one of fewer
extern "C" uint32_t dropest_bam_decoder_wave_slots(const dropest_bam_decoder *d) {
	if (!d) return 0;
	hipDeviceProp_t prop{};
	if (hipGetDeviceProperties(&prop, d->device) != hipSuccess) return 0;
	return uint32_t(prop.multiProcessorCount) * 4u * uint32_t(INF_WAVES_PER_EU);
}

// Room on the device for windows of `bytes` compressed bytes in front `which`, up front (a BAM inflates ~4-12 x, a record is >= ~120 bytes), so that
// the first windows do not grow every buffer step by step (each growth is a free + an allocation that wait for the device)
static void bam_reserve(dropest_bam_decoder *d, int which, uint64_t bytes, uint64_t out_hint = 0) {
	BamFront &F = d->front[which];
	// (14 x when the caller does not know better: 2 x 1.8 GB for windows of 128 MB, 1-2 ms on most boxes and 130 ms on some)
	const uint64_t out_bytes = out_hint ? out_hint + out_hint / 8 + (uint64_t(1) << 20) : bytes * 14, n_rec = out_bytes / 120, n_blk = bytes / 2048 + 1024, n_seg = out_bytes / BAM_SEG + 16;
	F.d_in.ensure(bytes + 8); F.d_out.ensure(out_bytes);
	d->up_in[which].ensure(bytes + 8);
	F.d_in_off.ensure(n_blk); F.d_out_off.ensure(n_blk); F.d_in_len.ensure(n_blk); F.d_out_len.ensure(n_blk); F.d_status.ensure(n_blk); F.d_crc.ensure(n_blk); F.h_block_status.ensure(n_blk);
	F.seg_start.ensure(n_seg); F.seg_exit.ensure(n_seg); F.seg_count.ensure(n_seg); F.seg_base.ensure(n_seg);
	F.h_seg_start.ensure(n_seg); F.h_seg_exit.ensure(n_seg); F.h_count.ensure(n_seg); F.h_base.ensure(n_seg); F.h_tab.ensure(n_blk * 4 + 8);
	if (which == 0) {
		d->rec_off.ensure(n_rec); d->o_cb.ensure(n_rec); d->o_umi.ensure(n_rec); d->o_gene.ensure(n_rec); d->o_aux.ensure(n_rec); d->o_uql.ensure(n_rec);
		d->o_status.ensure(n_rec); d->o_need.ensure(n_rec); d->dn_cb.ensure(n_rec); d->dn_umi.ensure(n_rec); d->dn_gene.ensure(n_rec); d->dn_aux.ensure(n_rec);
		d->nd_rec.ensure(n_rec); d->nd_pos.ensure(n_rec); d->nd_size.ensure(n_rec);
		d->o_qoff.ensure(n_rec); d->dn_qoff.ensure(n_rec);
		const uint64_t tiles = n_rec / BAM_FIN_TILE + 2;
		d->tile_ok.ensure(tiles); d->tile_need.ensure(tiles); d->d_totals.ensure(2); d->d_wc.ensure(1);
		if (d->annotation) { d->a_chr.ensure(n_rec); d->a_pos.ensure(n_rec); d->a_end.ensure(n_rec); d->a_gene.ensure(n_rec); d->a_mark.ensure(n_rec); }
	}
}

extern "C" int dropest_bam_decoder_staging(dropest_bam_decoder *d, int which, uint64_t bytes, uint8_t **out) {
	return bgzf_guarded([&] {
		if (!d || !out || which < 0 || which > 1) throw InvalidError("bad argument");
		HIP_CHECK(hipSetDevice(d->device));
		using clk = std::chrono::steady_clock;
		const bool trace = getenv("DROPEST_BAM_TRACE") != nullptr;
		auto t0 = clk::now();
		auto lap = [&](const char *what) { if (trace) { std::fprintf(stderr, "[bam] staging %d: %s %.1f ms\n", which, what, std::chrono::duration<double, std::milli>(clk::now() - t0).count()); t0 = clk::now(); } };
		d->h_stage[which].ensure(bytes);
		lap("pinned buffer");
		*out = d->h_stage[which].p;
		bam_reserve(d, which, bytes);
		lap("device buffers");
	});
}

// The same without the pinned buffer: for a caller that sends the compressed bytes in pieces (below)
extern "C" int dropest_bam_decoder_reserve(dropest_bam_decoder *d, int which, uint64_t bytes, uint64_t inflated_bytes) {
	return bgzf_guarded([&] {
		if (!d || which < 0 || which > 1) throw InvalidError("bad argument");
		HIP_CHECK(hipSetDevice(d->device));
		bam_reserve(d, which, bytes, inflated_bytes);
	});
}

// n pinned pieces of `bytes` each (kept by the decoder; a second call with other sizes replaces them): out[k] = piece k
extern "C" int dropest_bam_decoder_pieces(dropest_bam_decoder *d, uint32_t n, uint64_t bytes, uint8_t **out) {
	return bgzf_guarded([&] {
		if (!d || !out || !n || n > 64 || !bytes) throw InvalidError("bad argument");
		HIP_CHECK(hipSetDevice(d->device));
		if (d->n_pieces) HIP_CHECK(hipStreamSynchronize(d->up_stream));      // (copies out of the pieces there are)
		if (d->n_pieces != n) {
			for (hipEvent_t e : d->piece_done) if (e) (void)hipEventDestroy(e);
			d->piece_done.clear();
			d->n_pieces = 0;
			d->piece_done.assign(n, nullptr);
			for (hipEvent_t &e : d->piece_done) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		}
		const uint64_t each = (bytes + 4095u) & ~uint64_t(4095);
		// (DROPEST_BAM_PINNED_COHERENT=1: the default kind of pinned memory, fine-grained, instead of memory the CPU caches)
		static const bool coherent = getenv("DROPEST_BAM_PINNED_COHERENT") != nullptr;
		d->piece_mem.ensure_exact(each * n, coherent ? hipHostMallocDefault : hipHostMallocNonCoherent);
		d->n_pieces = n; d->piece_bytes = each;
		for (uint32_t k = 0; k < n; ++k) out[k] = d->piece_mem.p + each * k;
	});
}

// Piece `piece` is free again: the copy that read it last has finished (at once when there was none)
extern "C" int dropest_bam_decoder_piece_wait(dropest_bam_decoder *d, uint32_t piece) {
	if (!d || piece >= d->n_pieces) return 1;
	if (hipSetDevice(d->device) != hipSuccess) return 1;
	return hipEventSynchronize(d->piece_done[piece]) == hipSuccess ? 0 : 1;
}

// A window's pieces are about to be sent to up-buffer `which` (one thread; the .._upload_piece calls that follow may come from several)
extern "C" int dropest_bam_decoder_upload_begin(dropest_bam_decoder *d, int which) {
	if (!d || which < 0 || which > 1 || !d->n_pieces) return 1;
	d->up_ready[which].store(false, std::memory_order_relaxed);
	d->up_begun[which] = true;
	return 0;
}

// The first `len` bytes of piece `piece` to up-buffer `which` at byte `dst_off`, on the upload stream; does not wait.  May be called from several
// threads (different pieces).  .._upload_done(which, host bytes, total length) then says what the buffer holds.
extern "C" int dropest_bam_decoder_upload_piece(dropest_bam_decoder *d, int which, uint32_t piece, uint64_t dst_off, uint64_t len) {
	if (!d || which < 0 || which > 1 || piece >= d->n_pieces || len > d->piece_bytes || dst_off + len + 8 > d->up_in[which].n) return 1;
	if (!len) return 0;
	if (hipSetDevice(d->device) != hipSuccess) return 1;
	if (!d->up_begun[which]) return 1;      // (.._upload_begin was not called)
	if (hipMemcpyAsync(d->up_in[which].p + dst_off, d->piece_mem.p + d->piece_bytes * piece, len, hipMemcpyHostToDevice, d->up_stream) != hipSuccess) return 1;
	return hipEventRecord(d->piece_done[piece], d->up_stream) == hipSuccess ? 0 : 1;
}

// Every piece of a window has been given to .._upload_piece: up-buffer `which` holds (will hold, when those copies are through) host[0 .. len).
// The block table is made from `host` (which must stay readable until the window call that is given `host` and `len` has returned -- the host
// fall-back for a refused block reads it).
extern "C" int dropest_bam_decoder_upload_done(dropest_bam_decoder *d, int which, const uint8_t *host, uint64_t len, const dropest_bgzf_blocks *blocks) {
	if (!d || which < 0 || which > 1) return 1;
	d->up_ready[which].store(false, std::memory_order_relaxed);
	if (!len || !host || len + 8 > d->up_in[which].n) return 0;
	if (hipSetDevice(d->device) != hipSuccess) return 1;
	if (!d->up_begun[which] || hipEventRecord(d->up_done[which], d->up_stream) != hipSuccess) return 1;
	d->up_begun[which] = false;
	auto &b = d->up_blocks[which];
	b.ok = false;
	try {
		if (blocks && blocks->n && blocks->in_off && blocks->in_len && blocks->out_len && blocks->crc32) {
			// the caller's table (made while it read the file: `host` is then not touched at all unless a block is refused), checked for what the kernel relies on
			const uint64_t n = blocks->n;
			if (b.in_off.size() < n) { const uint64_t c2 = n + n / 2; b.in_off.resize(c2); b.out_off.resize(c2); b.in_len.resize(c2); b.out_len.resize(c2); b.crc.resize(c2); }
			uint64_t total = 0, at = 0;
			bool sound = true;
			for (uint64_t k = 0; k < n && sound; ++k) {
				sound = blocks->in_off[k] >= at + 18 && blocks->in_off[k] + blocks->in_len[k] + 8 <= len && blocks->out_len[k] <= 65536u;
				b.in_off[k] = blocks->in_off[k]; b.in_len[k] = blocks->in_len[k]; b.out_off[k] = total; b.out_len[k] = blocks->out_len[k]; b.crc[k] = blocks->crc32[k];
				total += blocks->out_len[k]; at = blocks->in_off[k] + blocks->in_len[k] + 8;
			}
			if (sound && at == len) { b.n = n; b.used = len; b.total = total; b.ok = true; }
		}
		if (!b.ok) b.ok = bgzf_block_table(host, len, b.in_off, b.out_off, b.in_len, b.out_len, b.crc, &b.n, &b.used, &b.total);
	} catch (...) { b.ok = false; }
	d->up_len[which] = len;
	d->up_host[which].store(host, std::memory_order_relaxed);
	d->up_ready[which].store(true, std::memory_order_release);
	return 0;
}

// The first `len` bytes of staging buffer `which` start their way to the device now, on a stream of their own: the window call that is then given
// exactly that buffer and length waits for this copy instead of making one -- with a reader thread that calls this when its read is done, the
// copy of window k + 1 runs under the kernels of window k.  Allocates nothing; the one call of the decoder that may run beside a window call
// (from another thread).  A length the buffers were not sized for is not an error: the window call copies as before.
extern "C" int dropest_bam_decoder_upload(dropest_bam_decoder *d, int which, uint64_t len) {
	if (!d || which < 0 || which > 1) return 1;
	d->up_ready[which].store(false, std::memory_order_relaxed);
	if (!len || !d->h_stage[which].p || len > d->h_stage[which].n || len + 8 > d->up_in[which].n) return 0;
	if (hipSetDevice(d->device) != hipSuccess) return 1;
	if (hipMemcpyAsync(d->up_in[which].p, d->h_stage[which].p, len, hipMemcpyHostToDevice, d->up_stream) != hipSuccess) return 1;
	if (hipEventRecord(d->up_done[which], d->up_stream) != hipSuccess) return 1;
	auto &b = d->up_blocks[which];      // while the copy runs: the block table of these bytes (a failure here is the window call's to report)
	b.ok = false;
	try { b.ok = bgzf_block_table(d->h_stage[which].p, len, b.in_off, b.out_off, b.in_len, b.out_len, b.crc, &b.n, &b.used, &b.total); } catch (...) { b.ok = false; }
	d->up_len[which] = len;
	d->up_host[which].store(d->h_stage[which].p, std::memory_order_relaxed);
	d->up_ready[which].store(true, std::memory_order_release);
	return 0;
}

// The decoder's kernels and small copies run on `stream` (a hipStream_t of the decoder's device) from now on -- the caller's own, which exists and
// has run kernels, instead of one the decoder would have to create (~16 ms with its first dispatch).  NULL: back to a stream of the decoder's own
// (made when first needed); call that before the lent stream goes away.  Not while a window is in flight.
extern "C" int dropest_bam_decoder_use_stream(dropest_bam_decoder *d, void *stream) {
	return bgzf_guarded([&] {
		if (!d) throw InvalidError("null argument");
		HIP_CHECK(hipSetDevice(d->device));
		std::lock_guard<std::mutex> lk(d->ready_mutex);
		if (getenv("DROPEST_BAM_TRACE_SYNC")) {      // (development: which stream still has work when a file is done)
			using clk = std::chrono::steady_clock;
			auto t0 = clk::now();
			(void)hipStreamSynchronize(d->up_stream);
			std::fprintf(stderr, "[bam] use_stream: the upload stream waited for %.1f ms\n", std::chrono::duration<double, std::milli>(clk::now() - t0).count());
			t0 = clk::now();
			if (d->stream) (void)hipStreamSynchronize(d->stream);
			std::fprintf(stderr, "[bam] use_stream: the kernels' stream waited for %.1f ms\n", std::chrono::duration<double, std::milli>(clk::now() - t0).count());
		}
		if (d->stream) HIP_CHECK(hipStreamSynchronize(d->stream));
		d->stream = stream ? hipStream_t(stream) : d->own_stream;      // (null: bam_ready makes the decoder's own)
	});
}

extern "C" void dropest_bam_decoder_destroy(dropest_bam_decoder *d) {
	if (!d) return;
	(void)hipSetDevice(d->device);
	(void)hipStreamSynchronize(d->up_stream);
	for (hipEvent_t e : d->up_done) if (e) (void)hipEventDestroy(e);
	for (hipEvent_t e : d->piece_done) if (e) (void)hipEventDestroy(e);
	if (d->own_stream) { (void)hipStreamSynchronize(d->own_stream); (void)hipStreamDestroy(d->own_stream); }
	for (BamFront &f : d->front) if (f.stream) { (void)hipStreamSynchronize(f.stream); (void)hipStreamDestroy(f.stream); }
	delete d;
}

// Host bytes to a device array without a staged copy: into the decoder's pinned staging (at `*at`, which moves on), then a kernel's loads.  The
// dictionaries' tables are 17 KB to a few hundred KB -- sizes at which hipMemcpyAsync out of pageable memory took 8-12 ms now and then (measured:
// "dictionaries to the device" 0.7 or 12 ms per file).  The caller waits for the stream before the staging is used again.
__global__ __launch_bounds__(256) void bam_bytes_from_host_kernel(const uint8_t *h, uint8_t *__restrict__ d, uint64_t n) {
	const uint64_t k = (uint64_t(blockIdx.x) * 256 + threadIdx.x) * 16;
	if (k + 16 <= n) { uint4 v; __builtin_memcpy(&v, h + k, 16); __builtin_memcpy(d + k, &v, 16); }
	else for (uint64_t i = k; i < n; ++i) d[i] = h[i];
}
static void bam_to_device(dropest_bam_decoder *d, void *dst, const void *src, size_t bytes, size_t &at) {
	if (!bytes) return;
	const size_t from = (at + 15) & ~size_t(15);
	if (from + bytes > d->h_dict.n) throw InvalidError("internal: the dictionaries' staging is too small");
	std::memcpy(d->h_dict.p + from, src, bytes);
	hipLaunchKernelGGL(bam_bytes_from_host_kernel, dim3(uint32_t((bytes + 4095) / 4096)), dim3(256), 0, d->stream, d->h_dict.p + from, static_cast<uint8_t *>(dst), uint64_t(bytes));
	HIP_CHECK(hipGetLastError());
	at = from + bytes;
}

extern "C" int dropest_bam_decoder_set_annotation(dropest_bam_decoder *d, dropest_annotation *a, const int32_t *ann_chr_of_ref, uint32_t n_refs) {
	return bgzf_guarded([&] {
		if (!d || !a || (n_refs && !ann_chr_of_ref)) throw InvalidError("null argument");
		if (n_refs != uint32_t(d->cfg.n_refs)) throw InvalidError("one annotation chromosome per reference is expected");
		if (dropest_annotation_device(a) != d->device) throw InvalidError("the annotation lives on another GPU");
		HIP_CHECK(hipSetDevice(d->device));
		bam_ready(d);
		d->annotation = a;
		d->n_ann_genes = dropest_annotation_genes(a);
		d->d_ann_chr.alloc(std::max<uint32_t>(n_refs, 1u));
		size_t at = 0;
		d->h_dict.ensure(size_t(n_refs) * 4 + 64);
		if (n_refs) bam_to_device(d, d->d_ann_chr.p, ann_chr_of_ref, size_t(n_refs) * 4, at);
		d->d_ann_id.alloc(std::max<uint32_t>(d->n_ann_genes, 1u));
		HIP_CHECK(hipMemsetAsync(d->d_ann_id.p, 0xFF, size_t(std::max<uint32_t>(d->n_ann_genes, 1u)) * 4, d->stream));   // no gene of the annotation is in the dictionary yet
		HIP_CHECK(hipStreamSynchronize(d->stream));
	});
}

extern "C" int dropest_bam_decoder_set_annotation_genes(dropest_bam_decoder *d, const int32_t *id_of_ann_gene, uint32_t n) {
	return bgzf_guarded([&] {
		if (!d || !d->annotation || (n && !id_of_ann_gene)) throw InvalidError("no annotation was given to this decoder");
		if (n != d->n_ann_genes) throw InvalidError("one entry per gene of the annotation is expected");
		HIP_CHECK(hipSetDevice(d->device));
		bam_ready(d);
		HIP_CHECK(hipStreamSynchronize(d->stream));
		if (n) { size_t at = 0; d->h_dict.ensure(size_t(n) * 4 + 64); bam_to_device(d, d->d_ann_id.p, id_of_ann_gene, size_t(n) * 4, at); HIP_CHECK(hipStreamSynchronize(d->stream)); }
	});
}

// The names of the genes in the dictionary, by index: name k = pool[off[k] .. off[k + 1]).  With them a record's gene is accepted on the device
// only if its bytes ARE the name the hash points at; a name that merely shares the FNV-1a value of another goes to the host (which interns it
// by its bytes, CellsDataContainer::intern_gene).  Call after dropest_bam_decoder_set_dictionaries, with every gene the dictionary holds.
extern "C" int dropest_bam_decoder_set_gene_names(dropest_bam_decoder *d, const uint32_t *off, const uint8_t *pool, uint32_t n_names) {
	return bgzf_guarded([&] {
		if (!d || (n_names && (!off || !pool))) throw InvalidError("null argument");
		HIP_CHECK(hipSetDevice(d->device));
		bam_ready(d);
		HIP_CHECK(hipStreamSynchronize(d->stream));
		d->n_gene_names = 0;
		if (!n_names) return;
		d->g_name_off.ensure(size_t(n_names) + 1 + n_names / 4); d->g_name_pool.ensure(size_t(off[n_names]) + off[n_names] / 4 + 16);
		size_t at = 0;
		d->h_dict.ensure((size_t(n_names) + 1) * 4 + size_t(off[n_names]) + 128);
		bam_to_device(d, d->g_name_off.p, off, (size_t(n_names) + 1) * 4, at);
		if (off[n_names]) bam_to_device(d, d->g_name_pool.p, pool, off[n_names], at);
		HIP_CHECK(hipStreamSynchronize(d->stream));
		d->n_gene_names = n_names;
	});
}

extern "C" int dropest_bam_decoder_set_dictionaries(dropest_bam_decoder *d, const uint64_t *gene_hash, const uint32_t *gene_id, uint32_t n_genes,
                                                    const int32_t *chr_of_ref, uint32_t n_refs) {
	return bgzf_guarded([&] {
		if (!d || (n_genes && (!gene_hash || !gene_id)) || (n_refs && !chr_of_ref)) throw InvalidError("null argument");
		if (n_refs != uint32_t(d->cfg.n_refs)) throw InvalidError("the chromosome table does not have one entry per reference");
		HIP_CHECK(hipSetDevice(d->device));
		bam_ready(d);
		uint32_t cap = 1024;
		while (cap < n_genes * 2u + 16u) cap <<= 1;
		std::vector<unsigned long long> keys(cap, 0);
		std::vector<uint32_t> vals(cap, 0);
		for (uint32_t k = 0; k < n_genes; ++k) {
			uint32_t sl = bam_dict_slot(gene_hash[k], cap - 1);
			while (vals[sl] && keys[sl] != gene_hash[k]) sl = (sl + 1) & (cap - 1);
			if (!vals[sl]) { keys[sl] = gene_hash[k]; vals[sl] = gene_id[k] + 1; }      // (the first name with a hash keeps it, as the host's map does)
		}
		HIP_CHECK(hipStreamSynchronize(d->stream));
		d->g_keys.ensure(cap); d->g_vals.ensure(cap);
		size_t at = 0;
		d->h_dict.ensure(size_t(cap) * 12 + size_t(n_refs) * 4 + 128);
		bam_to_device(d, d->g_keys.p, keys.data(), size_t(cap) * 8, at);
		bam_to_device(d, d->g_vals.p, vals.data(), size_t(cap) * 4, at);
		if (n_refs) bam_to_device(d, d->d_chr.p, chr_of_ref, size_t(n_refs) * 4, at);
		HIP_CHECK(hipStreamSynchronize(d->stream));
		d->g_mask = cap - 1;
	});
}

__global__ __launch_bounds__(256) void bam_chain_to_host_kernel(const uint64_t *__restrict__ seg_start, const uint64_t *__restrict__ seg_exit, const uint32_t *__restrict__ seg_count, uint32_t n,
                                                                uint64_t *h_start, uint64_t *h_exit, uint32_t *h_count) {
	const uint32_t k = blockIdx.x * 256 + threadIdx.x;
	if (k < n) { h_start[k] = seg_start[k]; h_exit[k] = seg_exit[k]; h_count[k] = seg_count[k]; }
}

// a window's block table from pinned host memory to the device's arrays, in one launch (five hipMemcpyAsync calls out of pageable vectors are staged by
// the runtime; one of them took 8 ms whenever its staging had to grow)
__global__ __launch_bounds__(256) void bam_tables_from_host_kernel(const uint64_t *h_in_off, const uint64_t *h_out_off, const uint32_t *h_in_len, const uint32_t *h_out_len, const uint32_t *h_crc,
                                                                   uint64_t *__restrict__ in_off, uint64_t *__restrict__ out_off, uint32_t *__restrict__ in_len, uint32_t *__restrict__ out_len,
                                                                   uint32_t *__restrict__ crc, uint32_t n) {
	const uint32_t k = blockIdx.x * 256 + threadIdx.x;
	if (k < n) { in_off[k] = h_in_off[k]; out_off[k] = h_out_off[k]; in_len[k] = h_in_len[k]; out_len[k] = h_out_len[k]; crc[k] = h_crc[k]; }
}

// words of a pinned host array to the device by a kernel's loads (a hipMemcpyAsync of ~50 KB took 4-7 ms the first time a window was large enough to need one)
__global__ __launch_bounds__(256) void bam_words_from_host_kernel(const uint32_t *h, uint32_t *__restrict__ d, uint32_t n) {
	const uint32_t k = blockIdx.x * 256 + threadIdx.x;
	if (k < n) d[k] = h[k];
}

// ... and words of a device array to a pinned host array by a kernel's stores (the blocks' verdicts)
__global__ __launch_bounds__(256) void bam_words_to_host_kernel(const uint32_t *__restrict__ d, uint32_t *h, uint32_t n) {
	const uint32_t k = blockIdx.x * 256 + threadIdx.x;
	if (k < n) h[k] = d[k];
}

// the records the host must see, to its pinned arrays by a kernel's stores (three copies of ~50 KB took 7.5 ms the first time a process made them)
__global__ __launch_bounds__(256) void bam_need_to_host_kernel(const uint32_t *__restrict__ rec, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ size, uint32_t n,
                                                               uint32_t *h_rec, uint32_t *h_pos, uint32_t *h_size) {
	const uint32_t k = blockIdx.x * 256 + threadIdx.x;
	if (k < n) { h_rec[k] = rec[k]; h_pos[k] = pos[k]; h_size[k] = size[k]; }
}

// First half of a window, part A: the block table, the tables and (unless dropest_bam_decoder_upload sent them ahead) the compressed bytes to the
// device, the inflate kernel enqueued -- and back, without waiting for any of it.  The blocks land at a fixed distance from the start of the
// front's buffer, so this may be called for window k + 1 while window k is still in its chain / fields / host work: the device then always has
// the next window's blocks to fill the wave slots that window k's finished blocks leave (a window's inflate takes as long as its slowest
// block, ~7-11 ms on a file that deflates 3 x, however few blocks are left running).  comp must stay as it is until .._window_chain returns.
constexpr uint64_t BAM_TAIL_RESERVE = uint64_t(1) << 20;
extern "C" int dropest_bam_decoder_window_inflate(dropest_bam_decoder *dec, const uint8_t *comp, uint64_t len, int *slot) {
	return bgzf_guarded([&] {
		if (!dec || !slot || (len && !comp)) throw InvalidError("null argument");
		BamFront *const d = &dec->front[dec->next_front];
		if (d->inflating) throw InvalidError("this front still holds a window whose blocks are being inflated (call .._window_chain first)");
		d->begun = false;
		*slot = dec->next_front;
		dec->next_front ^= 1;
		using clk = std::chrono::steady_clock;
		auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
		HIP_CHECK(hipSetDevice(dec->device));
		bam_ready(dec);
		// the first half of a window on a stream of its own, for a caller that runs it beside the second half of the window before; the two
		// halves one after the other (dropest_bam_decoder_window) share the decoder's stream
		// (Beside window k's chain / fields kernels the inflate of window k + 1 holds every wave slot of the device for 7-11 ms: whatever else is
		// launched meanwhile waits for slots.  Measured on the 3.2 x file: the chain 1.3 -> 52 ms, the dictionaries' copies 1.2 -> 31 ms; with a CU
		// mask on this stream that leaves 32 CUs alone the chain is back at 2 ms but the fields kernels run on those 32 CUs, 28 -> 49 ms, and a
		// masked stream is a blocking one: every null-stream copy waits for the inflate.  NOTES_r06 section 3.)
		if (!dec->halves_in_sequence && !d->stream) HIP_CHECK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
		hipStream_t st = dec->halves_in_sequence ? dec->stream : d->stream;
		int up = -1;      // the bytes went ahead (dropest_bam_decoder_upload)
		for (int w = 0; w < 2; ++w)
			// (the buffer first: only the slot these bytes stand in is looked at -- the reader thread may be filling the other one right now)
			if (comp && comp == dec->up_host[w].load(std::memory_order_relaxed) && dec->up_ready[w].load(std::memory_order_acquire) && len == dec->up_len[w]) { up = w; dec->up_ready[w].store(false, std::memory_order_relaxed); }
		uint64_t n = 0, used = 0, total = 0;
		if (up >= 0 && dec->up_blocks[up].ok) {
			auto &b = dec->up_blocks[up];
			d->in_off.swap(b.in_off); d->out_off.swap(b.out_off); d->in_len.swap(b.in_len); d->out_len.swap(b.out_len); d->crc.swap(b.crc);
			n = b.n; used = b.used; total = b.total; b.ok = false;
		} else if (!bgzf_block_table(comp, len, d->in_off, d->out_off, d->in_len, d->out_len, d->crc, &n, &used, &total)) throw InvalidError(g_bgzf_error);
		if (used != len) throw InvalidError("a window must hold whole BGZF blocks");
		if (n > 0xFFFFFFFFull) throw UnsupportedError("more than 2^32 blocks in one window");
		if (total >= (uint64_t(1) << 40)) throw UnsupportedError("window too large");
		static const uint64_t reserve_min = [] { const char *e = getenv("DROPEST_BAM_TEST_TAIL_RESERVE"); return e ? uint64_t(std::max(0ll, atoll(e))) : BAM_TAIL_RESERVE; }();
		d->reserve = (std::max(reserve_min, dec->tail_len) + 255u) & ~uint64_t(255);   // (the last cut-off record seen is a hint; a longer one moves the data: .._window_chain)
		auto t0 = clk::now();
		d->d_out.ensure(d->reserve + total + total / 4 + 64);
		d->ms_copy = d->ms_inflate = d->ms_boundaries = 0;
		if (n) {
			const uint8_t *uploaded = nullptr;
			if (up >= 0) { uploaded = dec->up_in[up].p; HIP_CHECK(hipStreamWaitEvent(st, dec->up_done[up], 0)); }
			else d->d_in.ensure(len + len / 4 + 8);
			d->d_in_off.ensure(n + n / 4); d->d_out_off.ensure(n + n / 4); d->d_in_len.ensure(n + n / 4); d->d_out_len.ensure(n + n / 4);
			d->d_status.ensure(n + n / 4); d->d_crc.ensure(n + n / 4); d->h_block_status.ensure(n);
			if (!uploaded) HIP_CHECK(hipMemcpyAsync(d->d_in.p, comp, len, hipMemcpyHostToDevice, st));
			{
				d->h_tab.ensure(size_t(n) * 4 + 8);      // (words of 8 bytes: 2 n for the offsets, 1.5 n for the three 32-bit arrays)
				uint64_t *const t_in = d->h_tab.p, *const t_out = t_in + n;
				uint32_t *const t_ilen = reinterpret_cast<uint32_t *>(t_out + n), *const t_olen = t_ilen + n, *const t_crc = t_olen + n;
				std::memcpy(t_in, d->in_off.data(), n * 8); std::memcpy(t_out, d->out_off.data(), n * 8);
				std::memcpy(t_ilen, d->in_len.data(), n * 4); std::memcpy(t_olen, d->out_len.data(), n * 4); std::memcpy(t_crc, d->crc.data(), n * 4);
				hipLaunchKernelGGL(bam_tables_from_host_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, st, t_in, t_out, t_ilen, t_olen, t_crc, d->d_in_off.p, d->d_out_off.p, d->d_in_len.p,
				                   d->d_out_len.p, d->d_crc.p, uint32_t(n));
				HIP_CHECK(hipGetLastError());
			}
			d->ms_copy = ms_since(t0);
			if (dropest_bgzf_inflate_device(dec->device, st, uploaded ? uploaded : d->d_in.p, len, d->d_in_off.p, d->d_in_len.p, d->d_out_off.p, d->d_out_len.p, uint32_t(n), d->d_out.p + d->reserve, d->d_status.p, d->d_crc.p))
				throw DeviceError(g_bgzf_error);
			hipLaunchKernelGGL(bam_words_to_host_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, st, d->d_status.p, d->h_block_status.p, uint32_t(n));
			HIP_CHECK(hipGetLastError());
		}
		d->n_blocks = uint32_t(n); d->total = total; d->comp = comp; d->comp_len = len;
		d->inflating = true;
	});
}

// First half of a window, part B: waits for the inflate of `slot`, has refused blocks inflated on the host, puts the record the window before cut
// off in front of the data, and finds the chain of records (guesses per segment, walks, the host's check).  Windows take this call in file order.
extern "C" int dropest_bam_decoder_window_chain(dropest_bam_decoder *dec, int slot, uint32_t first_skip, int final, dropest_bgzf_host_inflate inflate_fallback, void *user) {
	return bgzf_guarded([&] {
		if (!dec || slot < 0 || slot > 1) throw InvalidError("bad argument");
		BamFront *const d = &dec->front[slot];
		if (!d->inflating) throw InvalidError("this window's inflate was not started (or failed)");
		d->inflating = false;
		struct { uint32_t refused_blocks = 0, guesses_repaired = 0; double ms_boundaries = 0; } o_, *out = &o_;
		using clk = std::chrono::steady_clock;
		auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
		HIP_CHECK(hipSetDevice(dec->device));
		bam_ready(dec);
		// (the blocks were inflated on the front's stream, which may leave CUs alone -- see _inflate; everything from here on runs on the
		// decoder's own stream, which may use them all: it must not queue behind the next window's inflate)
		hipStream_t st_inflate = dec->halves_in_sequence ? dec->stream : d->stream;
		hipStream_t st = dec->stream;
		const uint64_t n = d->n_blocks, total = d->total;
		const uint8_t *comp = d->comp;
		const uint64_t tail = dec->tail_len, data_len = tail + total;
		if (tail && first_skip) throw InvalidError("first_skip belongs to the first window");
		if (data_len >= (uint64_t(1) << 40)) throw UnsupportedError("window too large");
		auto t0 = clk::now();
		HIP_CHECK(hipStreamSynchronize(st_inflate));      // the blocks are inflated (and their verdicts on the host)
		if (n) {
			std::vector<uint8_t> tmp;
			for (uint64_t k = 0; k < n; ++k) {
				if (!d->h_block_status.p[k]) continue;
				++out->refused_blocks;
				if (!inflate_fallback) throw InvalidError("the device refused BGZF block " + std::to_string(k) + " of the window (status " + std::to_string(d->h_block_status.p[k]) + ") and no host inflate was given");
				tmp.resize(d->out_len[k] + 1);
				if (inflate_fallback(comp + d->in_off[k], d->in_len[k], tmp.data(), d->out_len[k], user)) throw InvalidError("damaged BGZF block (neither the device nor the host inflates it)");
				HIP_CHECK(hipMemcpyAsync(d->d_out.p + d->reserve + d->out_off[k], tmp.data(), d->out_len[k], hipMemcpyHostToDevice, st));
				HIP_CHECK(hipStreamSynchronize(st));
			}
		}
		d->ms_inflate = ms_since(t0);
		// the cut-off record of the window before, in front of the blocks
		if (tail > d->reserve) {      // longer than the room left for it: the window moves behind it in a buffer of its own (a record beyond 1 MB: rare)
			DevBuf<uint8_t> moved;
			moved.alloc(data_len + data_len / 4 + 64);
			if (total) HIP_CHECK(hipMemcpyAsync(moved.p + tail, d->d_out.p + d->reserve, total, hipMemcpyDeviceToDevice, st));
			HIP_CHECK(hipStreamSynchronize(st));
			d->d_out = std::move(moved);
			d->reserve = tail;
		}
		d->base_off = d->reserve - tail;
		uint8_t *const data = d->data();
		if (tail) HIP_CHECK(hipMemcpyAsync(data, dec->d_tail.p, tail, hipMemcpyDeviceToDevice, st));
		// 2. the chain of records: guesses per segment, walks, the host's check
		t0 = clk::now();
		const uint32_t n_segs = uint32_t((data_len + BAM_SEG - 1) / BAM_SEG);
		uint64_t expect = tail ? 0 : first_skip;
		uint64_t n_rec = 0;
		if (n_segs) {
			const size_t sc = size_t(n_segs) + n_segs / 4;
			d->seg_start.ensure(sc); d->seg_exit.ensure(sc); d->seg_count.ensure(sc); d->seg_base.ensure(sc); d->d_bad.ensure(2); d->d_list.ensure(1);
			d->h_seg_start.ensure(n_segs); d->h_seg_exit.ensure(n_segs); d->h_count.ensure(n_segs);
			HIP_CHECK(hipMemsetAsync(d->d_bad.p, 0, 8, st));
			HIP_CHECK(hipMemcpyAsync(d->seg_start.p, &expect, 8, hipMemcpyHostToDevice, st));
			if (n_segs > 1) hipLaunchKernelGGL(bam_seg_guess_kernel, dim3((n_segs - 1 + 3) / 4), dim3(256), 0, st, data, data_len, dec->cfg.n_refs, n_segs, d->seg_start.p);
			if (getenv("DROPEST_BAM_TEST_SPOIL_GUESSES")) hipLaunchKernelGGL(bam_spoil_guesses_kernel, dim3((n_segs + 255) / 256), dim3(256), 0, st, d->seg_start.p, n_segs);
			hipLaunchKernelGGL(bam_seg_walk_kernel, dim3((n_segs + 255) / 256), dim3(256), 0, st, data, data_len, (const uint32_t *)nullptr, n_segs, d->seg_start.p,
			                   d->seg_count.p, d->seg_exit.p, (const uint32_t *)nullptr, (uint64_t *)nullptr, d->d_bad.p + 1);
			// (a walk from a guess that is not on the chain reads anything as a length: what these walks flag is not looked at -- the walk that
			// writes the record offsets, from the checked starts, is the one whose flag counts: dropest_bam_decoder_window_finish)
			// (the three arrays go to the pinned host buffers by a kernel's stores, not by copies: a copy queues behind the next window's 80 MB on
			// their way in -- dropest_bam_decoder_upload -- and the chain took 0.75 instead of 0.25 ms per window)
			hipLaunchKernelGGL(bam_chain_to_host_kernel, dim3((n_segs + 255) / 256), dim3(256), 0, st, d->seg_start.p, d->seg_exit.p, d->seg_count.p, n_segs, d->h_seg_start.p, d->h_seg_exit.p, d->h_count.p);
			HIP_CHECK(hipGetLastError());
			HIP_CHECK(hipStreamSynchronize(st));
			d->h_base.ensure(n_segs + n_segs / 4);
			for (uint32_t k = 0; k < n_segs; ++k) {
				const uint64_t seg_end = uint64_t(k + 1) * BAM_SEG;
				const uint64_t want = expect < seg_end ? expect : BAM_NONE;      // no record starts in this segment: the chain is already past it
				if (d->h_seg_start.p[k] != want) {                                 // the guess was not on the chain: walk again from the true place
					if (k) ++out->guesses_repaired;
					d->h_seg_start.p[k] = want;
					HIP_CHECK(hipMemcpyAsync(d->seg_start.p + k, &d->h_seg_start.p[k], 8, hipMemcpyHostToDevice, st));
					HIP_CHECK(hipMemcpyAsync(d->d_list.p, &k, 4, hipMemcpyHostToDevice, st));
					hipLaunchKernelGGL(bam_seg_walk_kernel, dim3(1), dim3(256), 0, st, data, data_len, d->d_list.p, 1u, d->seg_start.p, d->seg_count.p, d->seg_exit.p,
					                   (const uint32_t *)nullptr, (uint64_t *)nullptr, d->d_bad.p + 1);
					HIP_CHECK(hipMemcpyAsync(d->h_seg_exit.p + k, d->seg_exit.p + k, 8, hipMemcpyDeviceToHost, st));
					HIP_CHECK(hipMemcpyAsync(d->h_count.p + k, d->seg_count.p + k, 4, hipMemcpyDeviceToHost, st));
					HIP_CHECK(hipStreamSynchronize(st));
				}
				d->h_base.p[k] = uint32_t(n_rec);
				n_rec += d->h_count.p[k];
				if (want != BAM_NONE) expect = d->h_seg_exit.p[k];
			}
			if (n_rec > 0xFFFFFFF0ull) throw UnsupportedError("more than 2^32 records in one window");
		}
		const uint64_t tail_start = n_segs ? expect : (tail ? 0 : first_skip);
		if (tail_start > data_len) throw InvalidError("Corrupt BAM record");
		if (final && tail_start != data_len) throw InvalidError("Truncated BAM file");
		out->ms_boundaries = ms_since(t0);
		// the cut-off record opens the next window
		const uint64_t new_tail = data_len - tail_start;
		if (new_tail) {
			dec->d_tail.ensure(new_tail + new_tail / 4 + 64);
			HIP_CHECK(hipMemcpyAsync(dec->d_tail.p, data + tail_start, new_tail, hipMemcpyDeviceToDevice, st));
			HIP_CHECK(hipStreamSynchronize(st));
		}
		dec->tail_len = new_tail;
		d->data_len = data_len; d->tail_start = tail_start; d->n_rec = n_rec; d->n_segs = n_segs;
		d->refused = out->refused_blocks; d->repaired = out->guesses_repaired;
		d->ms_boundaries = out->ms_boundaries;
		d->begun = true;
	});
}

// The two parts one after the other (the first half of a window as rounds before 6 had it)
extern "C" int dropest_bam_decoder_window_begin(dropest_bam_decoder *dec, const uint8_t *comp, uint64_t len, uint32_t first_skip, int final,
                                                dropest_bgzf_host_inflate inflate_fallback, void *user, int *slot) {
	if (dropest_bam_decoder_window_inflate(dec, comp, len, slot)) return 1;
	const int rc = dropest_bam_decoder_window_chain(dec, *slot, first_skip, final, inflate_fallback, user);
	if (rc && dec) dec->front[*slot].inflating = false;
	return rc;
}

extern "C" int dropest_bam_decoder_window_finish(dropest_bam_decoder *d, int slot, dropest_bam_window *out) {
	return bgzf_guarded([&] {
		if (!d || !out || slot < 0 || slot > 1) throw InvalidError("bad argument");
		BamFront &F = d->front[slot];
		if (!F.begun) throw InvalidError("this window was not begun (or its first half failed)");
		F.begun = false;
		using clk = std::chrono::steady_clock;
		auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
		HIP_CHECK(hipSetDevice(d->device));
		bam_ready(d);
		hipStream_t st = d->stream;
		*out = dropest_bam_window{};
		d->last_n_rec = 0; d->last_n_ok = 0; d->last_front = slot;
		const uint64_t n_rec = F.n_rec, data_len = F.data_len;
		const uint32_t n_segs = F.n_segs;
		out->n_blocks = F.n_blocks; out->window_bytes = data_len; out->refused_blocks = F.refused; out->guesses_repaired = F.repaired;
		out->ms_copy = F.ms_copy; out->ms_inflate = F.ms_inflate; out->ms_boundaries = F.ms_boundaries;
		auto t0 = clk::now();
		static const int trace_laps = getenv("DROPEST_BAM_TRACE_READER") ? atoi(getenv("DROPEST_BAM_TRACE_READER")) : 0;
		auto lap = [&](const char *what) { if (trace_laps >= 2 || (trace_laps && !d->traced_first)) std::fprintf(stderr, "[bam] a window's fields (%llu records): %s at %.2f ms\n", (unsigned long long)F.n_rec, what, ms_since(t0)); };
		// record offsets, the fields, the accepted records made dense
		BamWindowCounts wc{};
		uint32_t totals[2] = {0, 0}, bad_record = 0;
		static_assert(sizeof(BamWindowCounts) == 40, "ten words");
		d->h_result.ensure(16);
		lap("result words pinned");
		if (n_rec) {
			const size_t rc = size_t(n_rec) + size_t(n_rec) / 4;
			d->rec_off.ensure(rc);
			lap("rec_off ensured");
			hipLaunchKernelGGL(bam_words_from_host_kernel, dim3((n_segs + 255) / 256), dim3(256), 0, st, F.h_base.p, F.seg_base.p, n_segs);
			lap("segment bases copy queued");
			hipLaunchKernelGGL(bam_seg_walk_kernel, dim3((n_segs + 255) / 256), dim3(256), 0, st, F.data(), data_len, (const uint32_t *)nullptr, n_segs, F.seg_start.p,
			                   F.seg_count.p, F.seg_exit.p, F.seg_base.p, d->rec_off.p, F.d_bad.p);
			lap("record offsets walked (queued)");
			d->o_qoff.ensure(rc); d->dn_qoff.ensure(rc);
			d->o_cb.ensure(rc); d->o_umi.ensure(rc); d->o_gene.ensure(rc); d->o_aux.ensure(rc); d->o_uql.ensure(rc); d->o_status.ensure(rc); d->o_need.ensure(rc);
			d->dn_cb.ensure(rc); d->dn_umi.ensure(rc); d->dn_gene.ensure(rc); d->dn_aux.ensure(rc); d->nd_rec.ensure(rc); d->nd_pos.ensure(rc); d->nd_size.ensure(rc);
			const uint32_t tiles = uint32_t((n_rec + BAM_FIN_TILE - 1) / BAM_FIN_TILE);
			d->tile_ok.ensure(tiles + tiles / 4 + 1); d->tile_need.ensure(tiles + tiles / 4 + 1); d->d_totals.ensure(2); d->d_wc.ensure(1);
			HIP_CHECK(hipMemsetAsync(d->d_wc.p, 0, sizeof(BamWindowCounts), st));
			if (d->annotation) { d->a_chr.ensure(rc); d->a_pos.ensure(rc); d->a_end.ensure(rc); d->a_gene.ensure(rc); d->a_mark.ensure(rc); }
			lap("buffers ensured");
			const BamRecordOut ro{d->o_cb.p, d->o_umi.p, d->o_gene.p, d->o_aux.p, d->o_uql.p, d->o_qoff.p, d->o_status.p, d->o_need.p, d->a_chr.p, d->a_pos.p, d->a_end.p};
			const BamDict dict{d->g_keys.p, d->g_vals.p, d->g_mask, d->d_chr.p, d->annotation ? d->d_ann_chr.p : nullptr, d->annotation ? d->d_ann_id.p : nullptr,
			                   d->n_gene_names ? d->g_name_off.p : nullptr, d->g_name_pool.p, d->n_gene_names, d->hash_mask};
			const BamDense dn{d->dn_cb.p, d->dn_umi.p, d->dn_gene.p, d->dn_aux.p, d->nd_rec.p, d->nd_pos.p, d->nd_size.p, d->dn_qoff.p};
			if (d->annotation) HIP_CHECK(hipMemsetAsync(d->a_chr.p, 0xFF, size_t(n_rec) * 4, st));   // (records that are not accepted: "no such chromosome", ignored)
			hipLaunchKernelGGL(bam_parse_kernel, dim3(uint32_t((n_rec + BAM_PARSE_T - 1) / BAM_PARSE_T)), dim3(BAM_PARSE_T), 0, st, F.data(), d->rec_off.p, uint32_t(n_rec), d->cfg, dict, ro, F.d_bad.p);
			if (d->annotation) {
				if (dropest_annotation_query_device(d->annotation, st, n_rec, d->a_chr.p, d->a_pos.p, d->a_end.p, d->a_gene.p, d->a_mark.p)) throw DeviceError(dropest_annotation_last_error());
				hipLaunchKernelGGL(bam_resolve_annotated_kernel, dim3(uint32_t((n_rec + 255) / 256)), dim3(256), 0, st, uint32_t(n_rec), dict, ro, d->a_gene.p, d->a_mark.p);
			}
			hipLaunchKernelGGL(bam_fin_count_kernel, dim3(tiles), dim3(256), 0, st, d->o_status.p, d->o_need.p, d->o_uql.p, uint32_t(n_rec), d->tile_ok.p, d->tile_need.p, d->d_wc.p);
			hipLaunchKernelGGL(bam_fin_scan_kernel, dim3(1), dim3(1024), 0, st, d->tile_ok.p, d->tile_need.p, tiles, d->d_totals.p);
			hipLaunchKernelGGL(bam_fin_scatter_kernel, dim3(tiles), dim3(256), 0, st, F.data(), d->rec_off.p, ro, uint32_t(n_rec), d->tile_ok.p, d->tile_need.p, dn);
			HIP_CHECK(hipGetLastError());
			// (into pinned words: a copy to pageable memory is staged and waits for the stream each time -- three of them per window)
			HIP_CHECK(hipMemcpyAsync(d->h_result.p, d->d_wc.p, sizeof(wc), hipMemcpyDeviceToHost, st));
			HIP_CHECK(hipMemcpyAsync(d->h_result.p + 10, d->d_totals.p, 8, hipMemcpyDeviceToHost, st));
			HIP_CHECK(hipMemcpyAsync(d->h_result.p + 12, F.d_bad.p, 4, hipMemcpyDeviceToHost, st));
		}
		lap("kernels and copies queued");
		HIP_CHECK(hipStreamSynchronize(st));
		lap("kernels done");
		if (n_rec) { std::memcpy(&wc, d->h_result.p, sizeof(wc)); totals[0] = d->h_result.p[10]; totals[1] = d->h_result.p[11]; bad_record = d->h_result.p[12]; }
		if (bad_record) throw InvalidError("Corrupt BAM record");      // (a block_size below the 32 bytes of a record's fixed part or beyond 2^26, met by the walk from the checked starts; fixed part + name + cigar + bases longer than the record, met by the parse)
		const uint32_t n_need = totals[1];
		if (n_need) {
			d->h_need_rec.ensure(n_need); d->h_need_pos.ensure(n_need); d->h_need_size.ensure(n_need);
			lap("need arrays pinned");
			hipLaunchKernelGGL(bam_need_to_host_kernel, dim3((n_need + 255) / 256), dim3(256), 0, st, d->nd_rec.p, d->nd_pos.p, d->nd_size.p, n_need, d->h_need_rec.p, d->h_need_pos.p, d->h_need_size.p);
			HIP_CHECK(hipGetLastError());
			HIP_CHECK(hipStreamSynchronize(st));
		}
		lap("need arrays copied");
		d->traced_first = true;
		out->ms_parse = ms_since(t0);
		d->last_n_rec = n_rec; d->last_n_ok = totals[0];
		out->n_records = n_rec; out->tail_bytes = data_len - F.tail_start;
		for (int k = 0; k < 5; ++k) out->counts[k] = wc.status[k];
		out->n_accepted = totals[0];
		if (out->n_accepted != out->counts[0]) throw DeviceError("internal: the dense columns and the counters disagree");
		out->d_cb = reinterpret_cast<const uint64_t *>(d->dn_cb.p); out->d_umi = reinterpret_cast<const uint64_t *>(d->dn_umi.p); out->d_gene = d->dn_gene.p; out->d_aux = d->dn_aux.p;
		out->n_need = n_need; out->need_rec = d->h_need_rec.p; out->need_pos = d->h_need_pos.p; out->need_size = d->h_need_size.p;
		out->quality_seen = wc.quality; out->any_gene = wc.any_gene;
		out->quality_len_max = wc.any_gene ? wc.ql_max : 0u; out->quality_len_min = wc.any_gene ? 0xFFFFFFFFu - wc.ql_min_inv : 0u;
	});
}

extern "C" int dropest_bam_decoder_window(dropest_bam_decoder *d, const uint8_t *comp, uint64_t len, uint32_t first_skip, int final,
                                          dropest_bgzf_host_inflate inflate_fallback, void *user, dropest_bam_window *out) {
	int slot = 0;
	if (d) d->halves_in_sequence = true;
	const int rc = dropest_bam_decoder_window_begin(d, comp, len, first_skip, final, inflate_fallback, user, &slot);
	if (d) d->halves_in_sequence = false;
	if (rc) return 1;
	return dropest_bam_decoder_window_finish(d, slot, out);
}

__global__ __launch_bounds__(256) void bam_record_sizes_kernel(const uint8_t *__restrict__ data, const uint64_t *__restrict__ rec_off, const uint32_t *__restrict__ idx, uint32_t n,
                                                               uint32_t *__restrict__ size) {
	const uint32_t k = blockIdx.x * 256 + threadIdx.x;
	if (k < n) size[k] = 4u + b_le32(data + rec_off[idx[k]]);
}

// (The indices, sizes, offsets and the gathered bytes go through pinned memory that the kernels read and write themselves: the five copies this call
// used to make were 17 KB to 1 MB each -- sizes at which a hipMemcpyAsync out of or into pageable memory is staged, and took 8 ms whenever the
// staging grew: 3.3 -> 12.4 ms per file once the second window was 8 MB.)
extern "C" int dropest_bam_decoder_fetch_records(dropest_bam_decoder *d, const uint32_t *idx, uint32_t n, uint8_t *dst, uint64_t dst_cap, uint64_t *dst_off) {
	return bgzf_guarded([&] {
		if (!d || (n && (!idx || !dst || !dst_off))) throw InvalidError("null argument");
		if (!n) return;
		HIP_CHECK(hipSetDevice(d->device));
		bam_ready(d);
		for (uint32_t k = 0; k < n; ++k) if (idx[k] >= d->last_n_rec) throw RangeError("record index outside the window");
		d->h_gidx.ensure(n); d->h_goff.ensure(n); d->h_gsize.ensure(n);
		std::memcpy(d->h_gidx.p, idx, size_t(n) * 4);
		hipLaunchKernelGGL(bam_record_sizes_kernel, dim3((n + 255) / 256), dim3(256), 0, d->stream, d->front[d->last_front].data(), d->rec_off.p, d->h_gidx.p, n, d->h_gsize.p);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(hipStreamSynchronize(d->stream));
		uint64_t total = 0;
		for (uint32_t k = 0; k < n; ++k) { dst_off[k] = total; d->h_goff.p[k] = total; total += d->h_gsize.p[k]; }
		if (total > dst_cap) throw InvalidError("destination too small: " + std::to_string(total) + " bytes needed");
		d->h_gather.ensure(total + total / 4 + 64);
		hipLaunchKernelGGL(bam_gather_records_kernel, dim3((n + 3) / 4), dim3(256), 0, d->stream, d->front[d->last_front].data(), d->rec_off.p, d->h_gidx.p, d->h_goff.p, n, d->h_gather.p);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(hipStreamSynchronize(d->stream));
		std::memcpy(dst, d->h_gather.p, total);
	});
}

extern "C" int dropest_bam_decoder_columns_to_host(dropest_bam_decoder *d, uint64_t *cb, uint64_t *umi, uint32_t *gene, uint32_t *aux) {
	return bgzf_guarded([&] {
		if (!d) throw InvalidError("null argument");
		const uint64_t n = d->last_n_ok;
		if (!n) return;
		HIP_CHECK(hipSetDevice(d->device));
		bam_ready(d);
		if (cb) HIP_CHECK(hipMemcpyAsync(cb, d->dn_cb.p, n * 8, hipMemcpyDeviceToHost, d->stream));
		if (umi) HIP_CHECK(hipMemcpyAsync(umi, d->dn_umi.p, n * 8, hipMemcpyDeviceToHost, d->stream));
		if (gene) HIP_CHECK(hipMemcpyAsync(gene, d->dn_gene.p, n * 4, hipMemcpyDeviceToHost, d->stream));
		if (aux) HIP_CHECK(hipMemcpyAsync(aux, d->dn_aux.p, n * 4, hipMemcpyDeviceToHost, d->stream));
		HIP_CHECK(hipStreamSynchronize(d->stream));
	});
}

extern "C" int dropest_bam_decoder_quality_rows(dropest_bam_decoder *d, uint32_t ql, const uint8_t **rows) {
	return bgzf_guarded([&] {
		if (!d || !rows || !ql || ql > 255) throw InvalidError("bad argument");
		*rows = nullptr;
		const uint64_t n = d->last_n_ok;
		if (!n) return;
		HIP_CHECK(hipSetDevice(d->device));
		bam_ready(d);
		d->dn_qual.ensure(n * ql + n * ql / 4); d->h_qual.ensure(n * ql);
		hipLaunchKernelGGL(bam_quality_rows_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, d->stream, d->front[d->last_front].data(), d->dn_qoff.p, uint32_t(n), ql, d->dn_qual.p);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(hipMemcpyAsync(d->h_qual.p, d->dn_qual.p, n * ql, hipMemcpyDeviceToHost, d->stream));
		HIP_CHECK(hipStreamSynchronize(d->stream));
		*rows = d->h_qual.p;
	});
}

extern "C" int dropest_bam_decoder_patch(dropest_bam_decoder *d, const uint32_t *pos, const uint64_t *cb, const uint64_t *umi, const uint32_t *gene,
                                         const uint32_t *aux, uint32_t n) {
	return bgzf_guarded([&] {
		if (!d || (n && (!pos || !cb || !umi || !gene || !aux))) throw InvalidError("null argument");
		if (!n) return;
		HIP_CHECK(hipSetDevice(d->device));
		bam_ready(d);
		for (uint32_t k = 0; k < n; ++k) if (pos[k] >= d->last_n_ok) throw RangeError("row outside the dense columns");
		// (the caller's five arrays into one pinned buffer that the kernel reads itself: no staged copies)
		d->h_patch.ensure(size_t(n) * 4 + 8);      // words of 8 bytes: cb | umi | pos, gene, aux
		unsigned long long *const q_cb = reinterpret_cast<unsigned long long *>(d->h_patch.p), *const q_umi = q_cb + n;
		uint32_t *const q_pos = reinterpret_cast<uint32_t *>(q_umi + n), *const q_gene = q_pos + n, *const q_aux = q_gene + n;
		std::memcpy(q_cb, cb, size_t(n) * 8); std::memcpy(q_umi, umi, size_t(n) * 8);
		std::memcpy(q_pos, pos, size_t(n) * 4); std::memcpy(q_gene, gene, size_t(n) * 4); std::memcpy(q_aux, aux, size_t(n) * 4);
		const BamDense dn{d->dn_cb.p, d->dn_umi.p, d->dn_gene.p, d->dn_aux.p, nullptr, nullptr, nullptr, nullptr};
		hipLaunchKernelGGL(bam_patch_kernel, dim3((n + 255) / 256), dim3(256), 0, d->stream, q_pos, q_cb, q_umi, q_gene, q_aux, n, dn);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(hipStreamSynchronize(d->stream));
	});
}
