// matrix_decode.h -- host side of the count matrices' wire form (include/dropest_amd.h: dropest_matrix_bytes).
//
// ResultsPrinter::create_matrix (ResultsPrinter.cpp:433-442) wants the dgCMatrix slots i / x; over PCIe a matrix travels as one byte
// of row delta and one byte of count per entry (8 -> 2 bytes per entry: the link is what bounds the end of a pass).  This file turns
// the bytes into the 32-bit slots ON THE WAY: the device-to-host copy is cut into chunks of whole columns, every chunk carries an
// event, and a small pool of host threads widens a chunk's columns as soon as its event has fired -- cm_raw's decode runs under
// cm's emit and copy, and only the last chunk's decode is left when the last byte has landed.
//
// Plain host code (no kernels); dropest_matrix_bytes_widen uses the same walk without events.
#pragma once

#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "util.h"

namespace dropest {

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
	__builtin_ia32_pause();
#elif defined(__aarch64__)
	__asm__ __volatile__("yield");
#else
	std::this_thread::yield();
#endif
}

struct ByteMatrixView {
	const uint8_t *rd = nullptr, *vb = nullptr;   // row deltas, values
	const uint32_t *colptr = nullptr;
	uint64_t ncols = 0, nnz = 0;
};

// Columns [c0, c1) of the byte form into (ro, vo).  Listed entries (a 255) were written to their places beforehand: a listed row is
// read back from ro, a listed value is left alone.
inline void widen_columns_scalar(const ByteMatrixView &m, size_t c0, size_t c1, uint32_t *__restrict ro, uint32_t *__restrict vo) {
	const uint8_t *__restrict rd = m.rd, *__restrict vb = m.vb;
	const uint32_t *__restrict cp = m.colptr;
	for (size_t c = c0; c < c1; ++c) {
		uint32_t prev1 = 0;   // previous row + 1
		const uint32_t k1 = cp[c + 1];
		for (uint32_t k = cp[c]; k < k1; ++k) {
			const uint32_t d = rd[k], v = vb[k];
			const uint32_t row = d == 255u ? ro[k] : prev1 + d - 1u;
			prev1 = row + 1u;
			ro[k] = row;
			if (v != 255u) vo[k] = v;
		}
	}
}

// The same walk, sixteen entries at a time (AVX2): the deltas of a group without a 255 are summed as 16-bit lanes (sixteen deltas
// below 255 stay below 4 096), widened and stored; a group with a listed entry takes the scalar steps.  NT: the slots are written with
// non-temporal stores (they are written once and read by somebody else later; without them every line is first read for ownership:
// 16 instead of 8 bytes of memory traffic per entry) -- possible when ro and vo are equally aligned modulo 32 bytes.
__attribute__((target("avx2"))) inline void widen_columns_avx2(const ByteMatrixView &m, size_t c0, size_t c1, uint32_t *__restrict ro,
                                                               uint32_t *__restrict vo, bool nt) {
	const uint8_t *__restrict rd = m.rd, *__restrict vb = m.vb;
	const uint32_t *__restrict cp = m.colptr;
	if (nt && ((reinterpret_cast<uintptr_t>(ro) ^ reinterpret_cast<uintptr_t>(vo)) & 31u)) nt = false;
	const __m128i ff = _mm_set1_epi8(char(0xFF));
	for (size_t c = c0; c < c1; ++c) {
		uint32_t prev1 = 0;
		uint32_t k = cp[c];
		const uint32_t k1 = cp[c + 1];
		auto scalar_to = [&](uint32_t end) {
			for (; k < end; ++k) {
				const uint32_t d = rd[k], v = vb[k];
				const uint32_t row = d == 255u ? ro[k] : prev1 + d - 1u;
				prev1 = row + 1u;
				ro[k] = row;
				if (v != 255u) vo[k] = v;
			}
		};
		if (k1 - k >= 32u) {
			// up to the next 32-byte boundary of the outputs (8 entries)
			const uint32_t mis = uint32_t((reinterpret_cast<uintptr_t>(ro + k) >> 2) & 7u);
			if (mis) scalar_to(std::min(k1, k + (8u - mis)));
			while (k + 16u <= k1) {
				const __m128i d8 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(rd + k));
				const __m128i v8 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(vb + k));
				if (_mm_movemask_epi8(_mm_or_si128(_mm_cmpeq_epi8(d8, ff), _mm_cmpeq_epi8(v8, ff)))) { scalar_to(k + 16u); continue; }
				__m256i s = _mm256_cvtepu8_epi16(d8);
				s = _mm256_add_epi16(s, _mm256_slli_si256(s, 2));
				s = _mm256_add_epi16(s, _mm256_slli_si256(s, 4));
				s = _mm256_add_epi16(s, _mm256_slli_si256(s, 8));      // inclusive sums inside each half of eight
				const __m128i lo = _mm256_castsi256_si128(s);
				__m128i hi = _mm256_extracti128_si256(s, 1);
				hi = _mm_add_epi16(hi, _mm_set1_epi16(short(_mm_extract_epi16(lo, 7))));
				const __m256i base = _mm256_set1_epi32(int(prev1 - 1u));
				const __m256i r0 = _mm256_add_epi32(_mm256_cvtepu16_epi32(lo), base), r1 = _mm256_add_epi32(_mm256_cvtepu16_epi32(hi), base);
				const __m256i x0 = _mm256_cvtepu8_epi32(v8), x1 = _mm256_cvtepu8_epi32(_mm_srli_si128(v8, 8));
				if (nt) {
					_mm256_stream_si256(reinterpret_cast<__m256i *>(ro + k), r0); _mm256_stream_si256(reinterpret_cast<__m256i *>(ro + k + 8), r1);
					_mm256_stream_si256(reinterpret_cast<__m256i *>(vo + k), x0); _mm256_stream_si256(reinterpret_cast<__m256i *>(vo + k + 8), x1);
				} else {
					_mm256_storeu_si256(reinterpret_cast<__m256i *>(ro + k), r0); _mm256_storeu_si256(reinterpret_cast<__m256i *>(ro + k + 8), r1);
					_mm256_storeu_si256(reinterpret_cast<__m256i *>(vo + k), x0); _mm256_storeu_si256(reinterpret_cast<__m256i *>(vo + k + 8), x1);
				}
				prev1 += uint32_t(uint16_t(_mm_extract_epi16(hi, 7)));
				k += 16u;
			}
		}
		scalar_to(k1);
	}
	if (nt) _mm_sfence();
}

// The same with 512-bit registers (the hosts of the GPU boxes are Zen 5): sixteen entries per step as ONE register of sixteen 32-bit
// sums, and -- NT -- one full 64-byte line per store: a non-temporal store that fills a whole line needs no read for ownership and no
// write-combining across instructions (the 32-byte halves of the AVX2 walk reached memory as partial lines whenever a line's halves
// were written by different steps: slower than plain stores on those hosts).
__attribute__((target("avx512f,avx512bw,avx512vl"))) inline void widen_columns_avx512(const ByteMatrixView &m, size_t c0, size_t c1, uint32_t *__restrict ro,
                                                                                      uint32_t *__restrict vo, bool nt) {
	const uint8_t *__restrict rd = m.rd, *__restrict vb = m.vb;
	const uint32_t *__restrict cp = m.colptr;
	if (nt && ((reinterpret_cast<uintptr_t>(ro) ^ reinterpret_cast<uintptr_t>(vo)) & 63u)) nt = false;
	const __m128i ff = _mm_set1_epi8(char(0xFF));
	const __m512i zero = _mm512_setzero_si512();
	for (size_t c = c0; c < c1; ++c) {
		uint32_t prev1 = 0;
		uint32_t k = cp[c];
		const uint32_t k1 = cp[c + 1];
		auto scalar_to = [&](uint32_t end) {
			for (; k < end; ++k) {
				const uint32_t d = rd[k], v = vb[k];
				const uint32_t row = d == 255u ? ro[k] : prev1 + d - 1u;
				prev1 = row + 1u;
				ro[k] = row;
				if (v != 255u) vo[k] = v;
			}
		};
		if (k1 - k >= 48u) {
			const uint32_t mis = uint32_t((reinterpret_cast<uintptr_t>(ro + k) >> 2) & 15u);   // up to the next 64-byte line of the outputs
			if (mis) scalar_to(std::min(k1, k + (16u - mis)));
			while (k + 16u <= k1) {
				const __m128i d8 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(rd + k));
				const __m128i v8 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(vb + k));
				if (_mm_movemask_epi8(_mm_or_si128(_mm_cmpeq_epi8(d8, ff), _mm_cmpeq_epi8(v8, ff)))) { scalar_to(k + 16u); continue; }
				__m512i s = _mm512_cvtepu8_epi32(d8);
				s = _mm512_add_epi32(s, _mm512_alignr_epi32(s, zero, 15));    // lane i += lane i - 1
				s = _mm512_add_epi32(s, _mm512_alignr_epi32(s, zero, 14));
				s = _mm512_add_epi32(s, _mm512_alignr_epi32(s, zero, 12));
				s = _mm512_add_epi32(s, _mm512_alignr_epi32(s, zero, 8));     // inclusive sums of the sixteen deltas
				const __m512i rows = _mm512_add_epi32(s, _mm512_set1_epi32(int(prev1 - 1u)));
				const __m512i x = _mm512_cvtepu8_epi32(v8);
				if (nt) { _mm512_stream_si512(reinterpret_cast<__m512i *>(ro + k), rows); _mm512_stream_si512(reinterpret_cast<__m512i *>(vo + k), x); }
				else { _mm512_storeu_si512(ro + k, rows); _mm512_storeu_si512(vo + k, x); }
				prev1 += uint32_t(_mm_extract_epi32(_mm512_extracti32x4_epi32(s, 3), 3));
				k += 16u;
			}
		}
		scalar_to(k1);
	}
	if (nt) _mm_sfence();
}

// DROPEST_DECODE=scalar | avx2 | avx512 picks the walk (default: the widest the CPU has); DROPEST_DECODE_NT=0 / 1 the stores (default:
// non-temporal with 512-bit registers -- whole lines --, plain otherwise; measured on the GPU boxes, see profiles/NOTES_r04.md)
inline int decode_isa() {
	static const int isa = [] {
		const char *e = getenv("DROPEST_DECODE");
		const bool has512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
		const bool has2 = __builtin_cpu_supports("avx2");
		if (e && !strcmp(e, "scalar")) return 0;
		if (e && !strcmp(e, "avx2")) return has2 ? 1 : 0;
		return has512 ? 2 : (has2 ? 1 : 0);
	}();
	return isa;
}
inline bool decode_use_nt() {
	static const bool on = [] { const char *e = getenv("DROPEST_DECODE_NT"); return e ? atoi(e) != 0 : decode_isa() == 2; }();
	return on;
}
inline void widen_columns(const ByteMatrixView &m, size_t c0, size_t c1, uint32_t *ro, uint32_t *vo) {
	switch (decode_isa()) {
	case 2: widen_columns_avx512(m, c0, c1, ro, vo, decode_use_nt()); break;
	case 1: widen_columns_avx2(m, c0, c1, ro, vo, decode_use_nt()); break;
	default: widen_columns_scalar(m, c0, c1, ro, vo);
	}
}

// Cuts the columns [c0, c1) into runs of about `target` entries (whole columns: a long column is a run by itself).
inline void cut_columns(const uint32_t *colptr, size_t c0, size_t c1, uint64_t target, std::vector<uint32_t> &ends) {
	size_t c = c0;
	while (c < c1) {
		const uint64_t want = uint64_t(colptr[c]) + target;
		size_t lo = c + 1, hi = c1;                     // first boundary e in (c, c1] with colptr[e] >= want
		while (lo < hi) { const size_t mid = (lo + hi) / 2; if (colptr[mid] >= want) hi = mid; else lo = mid + 1; }
		ends.push_back(uint32_t(lo));
		c = lo;
	}
}

// One matrix on its way to the 32-bit slots.  Filled by the owner, then handed to DecodePool::submit; wait() returns when every column is
// widened -- or when the lists turned out longer than their capacity (status OVERFLOW: nothing usable was written, the owner emits a wider form).
struct DecodeJob {
	enum Status { RUNNING = 0, DONE = 1, OVERFLOW = 2, BAD_ROW = 3, BAD_VALUE = 4, FAILED = 5 };
	int device = 0;
	ByteMatrixView m;
	uint32_t *ro = nullptr, *vo = nullptr;
	// the two lists (positions, rows / values); their lengths are read from *r_count / *v_count once ev_lists has fired
	const uint32_t *r_count = nullptr, *r_pos = nullptr, *r_val = nullptr, *v_count = nullptr, *v_pos = nullptr, *v_val = nullptr;
	uint32_t rcap = 0, vcap = 0;
	hipEvent_t ev_lists = nullptr;                  // fires when both lists are on the host (null: they already are)
	std::vector<hipEvent_t> ev_chunk;               // fires when the bytes of chunk j are on the host (empty: everything already is)
	std::vector<uint32_t> chunk_end;                // chunk j = columns [chunk_end[j - 1], chunk_end[j])
	// ---- set by prepare() ----
	std::vector<uint32_t> slice_end, slice_chunk;   // slice s = columns [slice_end[s - 1], slice_end[s]) of chunk slice_chunk[s]
	std::unique_ptr<std::atomic<uint8_t>[]> chunk_ready;
	std::atomic<uint32_t> lists_state{0};           // 0 nobody looks, 1 somebody waits for the event, 2 counts known
	std::atomic<uint32_t> list_next{0}, list_done{0}, slice_next{0}, slice_done{0};
	uint32_t n_r = 0, n_v = 0, n_list_ranges = 0, n_r_ranges = 0;
	std::atomic<int> status{RUNNING};
	std::mutex mu;
	std::condition_variable cv;
	static constexpr uint32_t LIST_RANGE = 16384;

	void prepare(uint64_t slice_entries) {
		if (chunk_end.empty()) chunk_end.push_back(uint32_t(m.ncols));
		size_t c0 = 0;
		for (size_t j = 0; j < chunk_end.size(); ++j) {
			const size_t before = slice_end.size();
			cut_columns(m.colptr, c0, chunk_end[j], slice_entries, slice_end);
			slice_chunk.resize(slice_end.size(), uint32_t(j));
			(void)before;
			c0 = chunk_end[j];
		}
		chunk_ready.reset(new std::atomic<uint8_t>[chunk_end.size()]);
		for (size_t j = 0; j < chunk_end.size(); ++j) chunk_ready[j].store(ev_chunk.empty() ? 1 : 0, std::memory_order_relaxed);
		if (slice_end.empty()) finish(DONE);
	}
	void finish(int st) {
		int expect = RUNNING;
		if (status.compare_exchange_strong(expect, st)) { std::lock_guard<std::mutex> lk(mu); cv.notify_all(); }
	}
	bool running() const { return status.load(std::memory_order_acquire) == RUNNING; }
	static bool poll_event(hipEvent_t ev, const std::atomic<int> &st) {   // false: the job ended meanwhile or the event failed
		for (uint32_t it = 0;; ++it) {
			const hipError_t e = hipEventQuery(ev);
			if (e == hipSuccess) return true;
			if (e != hipErrorNotReady) return false;
			if (st.load(std::memory_order_relaxed) != RUNNING) return false;
			cpu_relax();
			if ((it & 1023u) == 1023u) std::this_thread::yield();
		}
	}
	// A worker's share of the job; returns when there is nothing left to claim.
	void work() {
		if (!running()) return;
		// the lists: one worker waits for them, the others for it
		uint32_t zero = 0;
		if (lists_state.load(std::memory_order_acquire) != 2u) {
			if (lists_state.compare_exchange_strong(zero, 1u)) {
				if (ev_lists && !poll_event(ev_lists, status)) { finish(FAILED); return; }
				n_r = r_count ? *r_count : 0u; n_v = v_count ? *v_count : 0u;
				if (n_r > rcap || n_v > vcap) { finish(OVERFLOW); return; }
				n_r_ranges = (n_r + LIST_RANGE - 1) / LIST_RANGE;
				n_list_ranges = n_r_ranges + (n_v + LIST_RANGE - 1) / LIST_RANGE;
				lists_state.store(2u, std::memory_order_release);
			} else
				while (lists_state.load(std::memory_order_acquire) != 2u) { if (!running()) return; cpu_relax(); }
		}
		for (;;) {
			const uint32_t i = list_next.fetch_add(1, std::memory_order_relaxed);
			if (i >= n_list_ranges) break;
			const bool rows = i < n_r_ranges;
			const uint32_t *lpos = rows ? r_pos : v_pos, *lval = rows ? r_val : v_val;
			const uint8_t *mark = rows ? m.rd : m.vb;
			uint32_t *out = rows ? ro : vo;
			const uint32_t n = rows ? n_r : n_v, b = (rows ? i : i - n_r_ranges) * LIST_RANGE, e = std::min(n, b + LIST_RANGE);
			// (the bytes themselves may still be on their way: whether a listed entry stands on a 255 is checked by the walk's owner only in
			// dropest_matrix_bytes_widen, where everything is there; here a wrong position shows as a position beyond the matrix)
			for (uint32_t k = b; k < e; ++k) {
				const uint32_t pos = lpos[k];
				if (pos >= m.nnz) { finish(rows ? BAD_ROW : BAD_VALUE); return; }
				if (check_marks && mark[pos] != 255u) { finish(rows ? BAD_ROW : BAD_VALUE); return; }
				out[pos] = lval[k];
			}
			list_done.fetch_add(1, std::memory_order_release);
		}
		while (list_done.load(std::memory_order_acquire) < n_list_ranges) { if (!running()) return; cpu_relax(); }
		const uint32_t n_slices = uint32_t(slice_end.size());
		for (;;) {
			const uint32_t s = slice_next.fetch_add(1, std::memory_order_relaxed);
			if (s >= n_slices) break;
			const uint32_t j = slice_chunk[s];
			if (!wait_chunk(j)) { if (running()) finish(FAILED); return; }
			const auto t0 = trace ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
			widen_columns(m, s ? slice_end[s - 1] : 0u, slice_end[s], ro, vo);
			if (trace) {
				const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
				uint64_t cur = slowest_slice_ns.load(std::memory_order_relaxed);
				while (uint64_t(us * 1e3) > cur && !slowest_slice_ns.compare_exchange_weak(cur, uint64_t(us * 1e3))) {}
			}
			if (slice_done.fetch_add(1, std::memory_order_acq_rel) + 1u == n_slices) finish(DONE);
		}
	}
	// Chunk j's bytes are on the host.  ONE thread at a time asks the runtime (hipEventQuery takes the runtime's locks: a dozen threads
	// polling events slowed the owner's own launches and copies down); the others watch the flags it sets.
	std::atomic<uint32_t> poller{0}, next_event{0};
	bool wait_chunk(uint32_t j) {
		for (uint32_t it = 0;; ++it) {
			if (chunk_ready[j].load(std::memory_order_acquire)) return true;
			if (!running()) return false;
			uint32_t free_ = 0;
			if (poller.compare_exchange_strong(free_, 1u, std::memory_order_acquire)) {
				bool ok = true;
				for (uint32_t e = next_event.load(std::memory_order_relaxed); e < ev_chunk.size(); ++e) {   // the events fire in order
					const hipError_t r = hipEventQuery(ev_chunk[e]);
					if (r == hipErrorNotReady) break;
					if (r != hipSuccess) { ok = false; break; }
					chunk_ready[e].store(1, std::memory_order_release);
					next_event.store(e + 1, std::memory_order_relaxed);
				}
				poller.store(0u, std::memory_order_release);
				if (!ok) return false;
			}
			cpu_relax();
			if ((it & 1023u) == 1023u) std::this_thread::yield();
		}
	}
	bool check_marks = false;
	bool trace = false;                              // DROPEST_WIRE_TRACE: the slowest slice of the job (a descheduled worker shows here)
	std::atomic<uint64_t> slowest_slice_ns{0};
	int wait() {
		for (uint32_t it = 0; it < (1u << 16); ++it) { if (!running()) return status.load(); cpu_relax(); }
		std::unique_lock<std::mutex> lk(mu);
		cv.wait(lk, [&] { return !running(); });
		return status.load();
	}
};

// The threads that widen.  Jobs are taken in the order they were submitted; every thread works on the oldest unfinished job until it
// has nothing left to claim there.  DROPEST_DECODE_THREADS sets the number (default: 14 -- the GPU boxes give a process 16 CPUs --, at most the hardware threads - 2, at least 1).
class DecodePool {
	std::vector<std::thread> threads;
	std::mutex mu;
	std::condition_variable cv;
	std::deque<std::pair<uint64_t, std::shared_ptr<DecodeJob>>> jobs;
	uint64_t next_id = 0;
	bool stop = false;
	DecodePool() {
		unsigned n = 14;
		if (const char *e = getenv("DROPEST_DECODE_THREADS")) n = unsigned(std::max(1, atoi(e)));
		const unsigned hw = std::thread::hardware_concurrency();
		if (hw > 2) n = std::min(n, hw - 2); else n = 1;
		for (unsigned t = 0; t < n; ++t)
			threads.emplace_back([this] {
				uint64_t want = 0;
				int device = -1;
				for (;;) {
					std::shared_ptr<DecodeJob> job;
					{
						std::unique_lock<std::mutex> lk(mu);
						cv.wait(lk, [&] { return stop || (!jobs.empty() && jobs.back().first >= want); });
						if (stop) return;
						for (auto &j : jobs) if (j.first >= want) { job = j.second; want = j.first + 1; break; }
					}
					if (!job) continue;
					if (job->device != device) { device = job->device; (void)hipSetDevice(device); }
					try { job->work(); } catch (...) { job->finish(DecodeJob::FAILED); }
				}
			});
	}
public:
	~DecodePool() {
		{ std::lock_guard<std::mutex> lk(mu); stop = true; }
		cv.notify_all();
		for (auto &t : threads) t.join();
	}
	static DecodePool &get() { static DecodePool p; return p; }
	unsigned size() const { return unsigned(threads.size()); }
	void submit(const std::shared_ptr<DecodeJob> &job) {
		{
			std::lock_guard<std::mutex> lk(mu);
			while (!jobs.empty() && !jobs.front().second->running()) jobs.pop_front();
			jobs.emplace_back(next_id++, job);
		}
		cv.notify_all();
	}
};

}  // namespace dropest
