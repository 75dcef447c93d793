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
#include <sys/prctl.h>

#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

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

inline void cpu_relax() { host_cpu_relax(); }

// NUMA node that holds the page of p, or -1.  (The page is read first: a page of a shared mapping this process has not touched yet -- the
// node-shared result buffer of a sharded run -- is not in its page table, and move_pages answers -EFAULT for it.)
inline int numa_node_of(const void *p) {
	if (!p) return -1;
	(void)*static_cast<const volatile unsigned char *>(p);
	void *page = reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(p) & ~uintptr_t(4095));
	int status = -1;
	if (syscall(SYS_move_pages, 0, 1ul, &page, nullptr, &status, 0) != 0 || status < 0) return -1;
	return status;
}

struct ByteMatrixView {
	const uint8_t *rd = nullptr, *vb = nullptr;   // row deltas, values
	const uint32_t *colptr = nullptr;             // column c = entries [colptr[c], colend ? colend[c] : colptr[c + 1])
	const uint32_t *colend = nullptr;             // set: the columns are a SELECTION of a larger matrix's (one shard's columns of the global
	                                              // matrix of a sharded run, csrc/shard_run.h), each with its own begin and end
	const uint32_t *bytebeg = nullptr;            // set (with colend): the bytes of column c stand at [bytebeg[c], ...) of rd / vb -- a shard's LOCAL byte
	                                              // arrays, contiguous in its own column order -- while colptr / colend name the column's place in ro / vo
	uint64_t ncols = 0, nnz = 0;
};

// Columns [c0, c1) of the byte form into (ro, vo).  Listed entries (a 255) were written to their places beforehand: a listed row is
// read back from ro, a listed value is left alone.
inline void widen_columns_scalar(const ByteMatrixView &m, size_t c0, size_t c1, uint32_t *__restrict ro, uint32_t *__restrict vo) {
	const uint32_t *__restrict cp = m.colptr, *__restrict ce = m.colend;
	for (size_t c = c0; c < c1; ++c) {
		const ptrdiff_t sh = m.bytebeg ? ptrdiff_t(m.bytebeg[c]) - ptrdiff_t(cp[c]) : 0;
		const uint8_t *__restrict rd = m.rd + sh, *__restrict vb = m.vb + sh;
		uint32_t prev1 = 0;   // previous row + 1
		const uint32_t k1 = ce ? ce[c] : cp[c + 1];
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
	const uint32_t *__restrict cp = m.colptr;
	if (nt && ((reinterpret_cast<uintptr_t>(ro) ^ reinterpret_cast<uintptr_t>(vo)) & 31u)) nt = false;
	const __m128i ff = _mm_set1_epi8(char(0xFF));
	for (size_t c = c0; c < c1; ++c) {
		const ptrdiff_t sh = m.bytebeg ? ptrdiff_t(m.bytebeg[c]) - ptrdiff_t(cp[c]) : 0;
		const uint8_t *__restrict rd = m.rd + sh, *__restrict vb = m.vb + sh;
		uint32_t prev1 = 0;
		uint32_t k = cp[c];
		const uint32_t k1 = m.colend ? m.colend[c] : cp[c + 1];
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
	const uint32_t *__restrict cp = m.colptr;
	if (nt && ((reinterpret_cast<uintptr_t>(ro) ^ reinterpret_cast<uintptr_t>(vo)) & 63u)) nt = false;
	const __m128i ff = _mm_set1_epi8(char(0xFF));
	const __m512i zero = _mm512_setzero_si512();
	for (size_t c = c0; c < c1; ++c) {
		const ptrdiff_t sh = m.bytebeg ? ptrdiff_t(m.bytebeg[c]) - ptrdiff_t(cp[c]) : 0;
		const uint8_t *__restrict rd = m.rd + sh, *__restrict vb = m.vb + sh;
		uint32_t prev1 = 0;
		uint32_t k = cp[c];
		const uint32_t k1 = m.colend ? m.colend[c] : cp[c + 1];
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

// ---- a matrix that rides on another one's rows (round 6: cm on cm_raw; k_misc.h: emit_values_on_rows_kernel) -------------------------------
// The BASE matrix's byte form gives the rows (m.rd, m.colptr: entries [colptr[c], colptr[c + 1]) of base column c); vb holds, aligned with the
// base's entries, the value each entry has in THIS matrix (0: not an entry here, 255: listed -- already written to its slot).  Column c's
// kept entries go to [out_begin[c], out_begin[c] + out_count[c]) of (ro, vo); out_begin[c] == 0xFFFFFFFF: the column is not in this matrix.
// A listed row of the base (delta 255) is read from the base's slots base_ro (the base's job has put it there: DecodeJob waits for that).
// Listed values (255) are written to their slots AFTER the walk (a rider's job scatters its list last).
// Returns false when a column does not keep the number of entries it was announced with.
struct DerivedView {
	const uint32_t *base_ro = nullptr, *out_begin = nullptr, *out_count = nullptr;
};
inline bool widen_derived_scalar(const ByteMatrixView &m, const DerivedView &dv, size_t c0, size_t c1, uint32_t *__restrict ro, uint32_t *__restrict vo) {
	const uint32_t *__restrict cp = m.colptr;
	for (size_t c = c0; c < c1; ++c) {
		uint32_t out = dv.out_begin[c];
		if (out == 0xFFFFFFFFu) continue;
		const uint32_t end = out + dv.out_count[c];
		uint32_t prev1 = 0;
		for (uint32_t k = cp[c]; k < cp[c + 1]; ++k) {
			const uint32_t d = m.rd[k], v = m.vb[k];
			const uint32_t row = d == 255u ? dv.base_ro[k] : prev1 + d - 1u;
			prev1 = row + 1u;
			if (!v) continue;
			if (out >= end) return false;
			ro[out] = row;
			vo[out] = v;      // (255: the listed value comes after the walk)
			++out;
		}
		if (out != end) return false;
	}
	return true;
}
// sixteen base entries per step: rows by the prefix sums of the plain walk, the entries with a value compressed to the front of a small
// pending buffer, and whole 64-byte lines of it streamed to the slots (non-temporal: a line that is written whole needs no read for ownership --
// with plain stores the rider's walk moved twice the bytes of the plain one and was the tail of the step).  Listed values are written to their
// slots AFTER the walk (DecodeJob: a rider scatters its list last), so a line may be stored over a slot that waits for one.
__attribute__((target("avx512f,avx512bw,avx512vl"))) inline bool widen_derived_avx512(const ByteMatrixView &m, const DerivedView &dv, size_t c0, size_t c1,
                                                                                      uint32_t *__restrict ro, uint32_t *__restrict vo) {
	const uint32_t *__restrict cp = m.colptr;
	const __m128i ff = _mm_set1_epi8(char(0xFF));
	const __m512i zero = _mm512_setzero_si512();
	const bool nt = decode_use_nt() && !((reinterpret_cast<uintptr_t>(ro) ^ reinterpret_cast<uintptr_t>(vo)) & 63u);
	alignas(64) uint32_t pr[48], pv[48];      // pending kept entries of the column at hand (fewer than 16 between steps)
	for (size_t c = c0; c < c1; ++c) {
		uint32_t out = dv.out_begin[c];
		if (out == 0xFFFFFFFFu) continue;
		const uint32_t end = out + dv.out_count[c];
		uint32_t prev1 = 0, k = cp[c], pend = 0;      // `out` counts the entries stored; out + pend the entries kept
		const uint32_t k1 = cp[c + 1];
		bool ok = true;
		auto flush_all = [&]() { for (uint32_t i = 0; i < pend; ++i) { ro[out + i] = pr[i]; vo[out + i] = pv[i]; } out += pend; pend = 0; };
		auto scalar_to = [&](uint32_t stop) {
			flush_all();
			for (; k < stop; ++k) {
				const uint32_t d = m.rd[k], v = m.vb[k];
				const uint32_t row = d == 255u ? dv.base_ro[k] : prev1 + d - 1u;
				prev1 = row + 1u;
				if (!v) continue;
				if (out >= end) { ok = false; k = stop; return; }
				ro[out] = row;
				vo[out] = v;      // (255: the listed value comes after the walk)
				++out;
			}
		};
		if (k1 - k >= 64u) {
			while (ok && k < k1 && ((reinterpret_cast<uintptr_t>(ro + out) & 63u) != 0)) scalar_to(k + 1u);      // to a line boundary of the slots
			while (ok && k + 16u <= k1) {
				const __m128i d8 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(m.rd + k));
				const __m128i v8 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(m.vb + k));
				if (_mm_movemask_epi8(_mm_cmpeq_epi8(d8, ff))) {      // a listed row of the base: these sixteen one by one (then to a line boundary again)
					scalar_to(k + 16u);
					while (ok && k < k1 && ((reinterpret_cast<uintptr_t>(ro + out) & 63u) != 0)) scalar_to(k + 1u);
					continue;
				}
				__m512i sum = _mm512_cvtepu8_epi32(d8);
				sum = _mm512_add_epi32(sum, _mm512_alignr_epi32(sum, zero, 15));
				sum = _mm512_add_epi32(sum, _mm512_alignr_epi32(sum, zero, 14));
				sum = _mm512_add_epi32(sum, _mm512_alignr_epi32(sum, zero, 12));
				sum = _mm512_add_epi32(sum, _mm512_alignr_epi32(sum, zero, 8));
				const __m512i rows = _mm512_add_epi32(sum, _mm512_set1_epi32(int(prev1 - 1u)));
				const __m512i x = _mm512_cvtepu8_epi32(v8);
				const __mmask16 keep = _mm512_test_epi32_mask(x, x);
				const uint32_t cnt = uint32_t(__builtin_popcount(unsigned(keep)));
				if (out + pend + cnt > end) { ok = false; break; }
				_mm512_storeu_si512(pr + pend, _mm512_maskz_compress_epi32(keep, rows));
				_mm512_storeu_si512(pv + pend, _mm512_maskz_compress_epi32(keep, x));
				pend += cnt;
				if (pend >= 16u) {
					const __m512i lr = _mm512_load_si512(pr), lv = _mm512_load_si512(pv);
					if (nt) { _mm512_stream_si512(reinterpret_cast<__m512i *>(ro + out), lr); _mm512_stream_si512(reinterpret_cast<__m512i *>(vo + out), lv); }
					else { _mm512_storeu_si512(ro + out, lr); _mm512_storeu_si512(vo + out, lv); }
					out += 16u; pend -= 16u;
					_mm512_store_si512(pr, _mm512_loadu_si512(pr + 16)); _mm512_store_si512(pv, _mm512_loadu_si512(pv + 16));
				}
				prev1 += uint32_t(_mm_extract_epi32(_mm512_extracti32x4_epi32(sum, 3), 3));
				k += 16u;
			}
		}
		if (ok) scalar_to(k1);
		if (!ok || out != end) return false;
	}
	if (nt) _mm_sfence();
	return true;
}
inline bool widen_derived(const ByteMatrixView &m, const DerivedView &dv, size_t c0, size_t c1, uint32_t *ro, uint32_t *vo) {
	return decode_isa() == 2 ? widen_derived_avx512(m, dv, c0, c1, ro, vo) : widen_derived_scalar(m, dv, c0, c1, ro, vo);
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
	// Arrival flags in pinned host memory, written by the device between the kernels that move the data (k_misc.h: the kernel of chunk
	// j + 1 starts by writing chunk j's flag -- a kernel starts when its predecessor on the stream is complete and visible): flags[0] ==
	// epoch: both lists are on the host; flags[1 + j] == epoch: the bytes of chunk j are.  The workers read memory, never the runtime
	// (a dozen threads asking hipEventQuery held the runtime's locks under the owner's own launches: passes of 16 ms among passes of 9).
	// flags == nullptr: everything is on the host already.
	const volatile uint32_t *flags = nullptr;
	uint32_t epoch = 0;
	const uint32_t *cut = nullptr;                  // [ncols + 1] running entry counts the slices are cut by (default: m.colptr; a selection of columns brings its own)
	// A matrix that rides on another one's rows (DerivedView above): m describes the BASE's byte form with this matrix's value bytes as vb, the
	// chunks are the base's, a chunk is ready when BOTH matrices' bytes of it have landed (flags2 / epoch2: the base's flags; null: all there),
	// and the columns wait until the base's job has put its listed rows into its slots (base_job; null or finished: they are there).
	bool derived = false;
	DerivedView dv;
	std::shared_ptr<DecodeJob> base_job;
	const volatile uint32_t *flags2 = nullptr;
	uint32_t epoch2 = 0;
	std::vector<uint32_t> chunk_end;                // chunk j = columns [chunk_end[j - 1], chunk_end[j])
	// ---- set by prepare() ----
	std::vector<uint32_t> slice_end, slice_chunk;   // slice s = columns [slice_end[s - 1], slice_end[s]) of chunk slice_chunk[s]
	std::unique_ptr<std::atomic<uint8_t>[]> chunk_ready;
	std::unique_ptr<std::atomic<uint8_t>[]> slice_state;   // 0 unclaimed, 1 somebody walks it, 2 done
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
			cut_columns(cut ? cut : m.colptr, c0, chunk_end[j], slice_entries, slice_end);
			slice_chunk.resize(slice_end.size(), uint32_t(j));
			(void)before;
			c0 = chunk_end[j];
		}
		slice_state.reset(new std::atomic<uint8_t>[slice_end.size() + 1]);
		for (size_t k = 0; k < slice_end.size(); ++k) slice_state[k].store(0, std::memory_order_relaxed);
		chunk_ready.reset(new std::atomic<uint8_t>[chunk_end.size()]);
		for (size_t j = 0; j < chunk_end.size(); ++j) chunk_ready[j].store(flags ? 0 : 1, std::memory_order_relaxed);
		if (slice_end.empty()) finish(DONE);
	}
	void finish(int st) {
		int expect = RUNNING;
		if (status.compare_exchange_strong(expect, st)) { std::lock_guard<std::mutex> lk(mu); cv.notify_all(); }
	}
	bool running() const { return status.load(std::memory_order_acquire) == RUNNING; }
	// No stage of the job may hang on ONE thread making progress: the hosts are shared, a worker (or the one thread that happens to poll
	// an event) loses its CPU for milliseconds now and then -- seen as passes of 16 ms among passes of 9.  So every wait below has a
	// patience: whoever has waited PATIENCE spins does the thing itself (everything here can be done twice with the same result).
	static constexpr uint32_t PATIENCE = 4096;   // ~ 100 us of pause instructions
	void scatter_list_range(uint32_t i, bool &bad) {
		const bool rows = i < n_r_ranges;
		const uint32_t *lpos = rows ? r_pos : v_pos, *lval = rows ? r_val : v_val;
		const uint8_t *mark = rows ? m.rd : m.vb;
		uint32_t *out = rows ? ro : vo;
		const uint32_t n = rows ? n_r : n_v, b = (rows ? i : i - n_r_ranges) * LIST_RANGE, e = std::min(n, b + LIST_RANGE);
		// (the bytes themselves may still be on their way: whether a listed entry stands on a 255 is checked only by
		// dropest_matrix_bytes_widen, where everything is there; here a wrong position shows as a position beyond the matrix)
		// (a shard's selection of columns: the lists arrive with GLOBAL positions -- translated on the device, k_misc.h: matrix_lists_out_global)
		for (uint32_t k = b; k < e; ++k) {
			const uint32_t pos = lpos[k];
			if (pos >= m.nnz || (check_marks && mark[pos] != 255u)) { finish(rows ? BAD_ROW : BAD_VALUE); bad = true; return; }
			out[pos] = lval[k];
		}
	}
	// A thread's share of the job; returns when there is nothing left to do for it.  rescue: the caller owns the job (it has nothing else
	// to do until the matrix is complete) and walks what others hold at once instead of after a patience.
	// Threads inside work(): a slice somebody else has finished meanwhile (rescue) may still be under a straggler's pen when the job is
	// DONE -- the results are complete, but the buffers must not be freed or rewritten until quiesce() has seen everybody leave.
	std::atomic<int> inside{0};
	struct Inside { std::atomic<int> &n; explicit Inside(std::atomic<int> &x) : n(x) { n.fetch_add(1, std::memory_order_acq_rel); } ~Inside() { n.fetch_sub(1, std::memory_order_acq_rel); } };
	void quiesce() const {
		for (uint32_t it = 0; inside.load(std::memory_order_acquire) != 0; ++it) { if (it < 4096u) cpu_relax(); else std::this_thread::yield(); }
	}
	void work(bool rescue = false) {
		Inside guard(inside);
		if (!running()) return;
		// 1. the lists have landed: their lengths
		for (uint32_t it = 0; lists_state.load(std::memory_order_acquire) != 2u; ++it) {
			if (!running()) return;
			if (!flags || flags[0] == epoch) {
				std::atomic_thread_fence(std::memory_order_acquire);
				const uint32_t nr = r_count ? *r_count : 0u, nv = v_count ? *v_count : 0u;
				if (nr > rcap || nv > vcap) { finish(OVERFLOW); return; }
				n_r = nr; n_v = nv;   // (several threads may write the same numbers)
				n_r_ranges = (nr + LIST_RANGE - 1) / LIST_RANGE;
				n_list_ranges = n_r_ranges + (nv + LIST_RANGE - 1) / LIST_RANGE;
				if (derived) n_r_ranges = n_list_ranges = 0;   // (a rider's listed values go to their slots after the walk, by the owner: finish_rider)
				lists_state.store(2u, std::memory_order_release);
				break;
			}
			idle(it);
		}
		// 2. the listed entries to their places
		bool bad = false;
		for (;;) {
			const uint32_t i = list_next.fetch_add(1, std::memory_order_relaxed);
			if (i >= n_list_ranges) break;
			scatter_list_range(i, bad);
			if (bad) return;
			list_done.fetch_add(1, std::memory_order_release);
		}
		for (uint32_t it = 0; list_done.load(std::memory_order_acquire) < n_list_ranges; ++it) {
			if (!running()) return;
			if (it > PATIENCE) {   // somebody sits on a range: all of them again, here
				for (uint32_t i = 0; i < n_list_ranges && !bad; ++i) scatter_list_range(i, bad);
				if (bad) return;
				break;
			}
			cpu_relax();
		}
		// (a rider: the base's listed rows must stand in the base's slots before a column reads them)
		if (derived && base_job) {
			for (uint32_t it = 0;; ++it) {
				if (!running()) return;
				const int bst = base_job->status.load(std::memory_order_acquire);
				if (bst == DONE) break;
				if (bst != RUNNING) { finish(bst == OVERFLOW ? OVERFLOW : FAILED); return; }      // the base takes a wider form: so does the rider
				if (base_job->lists_state.load(std::memory_order_acquire) == 2u && base_job->list_done.load(std::memory_order_acquire) >= base_job->n_list_ranges) break;
				if (it > PATIENCE) { base_job->work(false); it = 0; continue; }   // nobody is on the base's lists: this thread is
				idle(it);
			}
		}
		// 3. the columns, slice by slice as their chunks land
		const uint32_t n_slices = uint32_t(slice_end.size());
		for (;;) {
			const uint32_t s = slice_next.fetch_add(1, std::memory_order_relaxed);
			if (s >= n_slices) break;
			if (!wait_chunk(slice_chunk[s])) { if (running()) finish(FAILED); return; }
			// never downgrade the state: a rescuer (or a worker past its patience) may have walked this slice while we waited for the chunk --
			// a plain store of 1 over its 2 would let walk_slice count the slice a second time and finish the job one slice early
			{ uint8_t expect = 0; (void)slice_state[s].compare_exchange_strong(expect, uint8_t(1), std::memory_order_acq_rel); }
			if (test_delay_us) std::this_thread::sleep_for(std::chrono::microseconds(test_delay_us));   // (tests: a worker that loses its CPU right here)
			walk_slice(s, n_slices);
		}
		// Nothing left to claim: the slices others have claimed and not finished, oldest first (a slice walked twice gets the same values twice)
		for (uint32_t it = 0; running(); ++it) {
			if (rescue || it > PATIENCE) {
				for (uint32_t s = 0; s < n_slices && running(); ++s) {
					uint8_t st = slice_state[s].load(std::memory_order_acquire);
					if (st == 0 && s < slice_next.load(std::memory_order_relaxed)) st = 1;   // claimed, its owner still waits for the chunk (or lost its CPU there)
					if (st == 1 && wait_chunk(slice_chunk[s])) walk_slice(s, n_slices);
				}
				if (!rescue) return;
				it = 0;
			}
			if (!rescue && it == 0) { /* a worker gives the others PATIENCE spins before it doubles their work */ }
			cpu_relax();
		}
	}
	void walk_slice(uint32_t s, uint32_t n_slices) {
		const auto t0 = trace ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
		if (!derived) widen_columns(m, s ? slice_end[s - 1] : 0u, slice_end[s], ro, vo);
		else if (!widen_derived(m, dv, s ? slice_end[s - 1] : 0u, slice_end[s], ro, vo)) { finish(BAD_ROW); return; }
		if (trace) {
			const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
			uint64_t cur = slowest_slice_ns.load(std::memory_order_relaxed);
			while (uint64_t(us * 1e3) > cur && !slowest_slice_ns.compare_exchange_weak(cur, uint64_t(us * 1e3))) {}
		}
		if (slice_state[s].exchange(2, std::memory_order_acq_rel) != 2 && slice_done.fetch_add(1, std::memory_order_acq_rel) + 1u == n_slices) finish(DONE);
	}
	// Waiting for data that is still on the link: a short spin, then naps -- fourteen spinning threads take the cores (and their SMT
	// siblings) from the owner's thread, which is preparing the next matrix at that very time.
	static void idle(uint32_t it) {
		if (it < 2048u) { cpu_relax(); return; }
		std::this_thread::sleep_for(std::chrono::microseconds(20));
	}
	// Chunk j's bytes are on the host.  A flag that never comes (a device fault, an aborted stream: the kernel that raises it never ran)
	// fails the job after FLAG_TIMEOUT_S instead of parking the pool's workers -- and every later settle() -- for ever.
	static constexpr double FLAG_TIMEOUT_S = 120.0;
	bool wait_chunk(uint32_t j) {
		std::chrono::steady_clock::time_point t0;
		for (uint32_t it = 0;; ++it) {
			if (chunk_ready[j].load(std::memory_order_acquire)) return true;
			if (!running()) return false;
			if (flags[1 + j] == epoch && (!flags2 || flags2[1 + j] == epoch2)) { std::atomic_thread_fence(std::memory_order_acquire); chunk_ready[j].store(1, std::memory_order_release); return true; }
			if (it == 4096u) t0 = std::chrono::steady_clock::now();
			else if (it > 4096u && (it & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > FLAG_TIMEOUT_S) { finish(FAILED); return false; }
			idle(it);
		}
	}
	// A rider's walk stores whole lines over slots that wait for a listed value, so the list is applied LAST -- by the job's owner, once the job
	// is DONE and every thread has left it (a straggler that walks a slice a second time would write its 255s over the values again).
	// Returns the job's status (BAD_VALUE: a listed entry outside the matrix).
	int finish_rider() {
		const int st = wait();
		if (!derived || st != DONE) return st;
		quiesce();
		for (uint32_t k = 0; k < n_v; ++k) {
			if (v_pos[k] >= m.nnz) return BAD_VALUE;
			vo[v_pos[k]] = v_val[k];
		}
		return DONE;
	}
	bool check_marks = false;
	uint32_t test_delay_us = 0;                      // DROPEST_DECODE_TEST_DELAY_US (tests only): every worker naps between claiming a slice and walking it
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
	// The workers run on the NUMA node that holds the buffers they write (ROCm puts pinned host memory on the GPU's node; a worker on
	// the other socket writes it at about half the rate -- measured: the same pass 9.4 ms or 10.7 ms depending on where the scheduler had
	// put the threads).  prefer_node_of(p) is called by the owner of a job with its output buffer; the workers move when the node changes.
	// DROPEST_DECODE_NUMA=0 leaves them where the scheduler puts them.
	static std::vector<int> cpus_of_node(int node) {
		std::vector<int> list;
		char path[96];
		snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
		FILE *f = fopen(path, "r");
		if (!f) return list;
		char buf[4096] = {0};
		const bool got = fgets(buf, sizeof(buf), f) != nullptr;
		fclose(f);
		if (!got) return list;
		for (char *p = buf; *p;) {
			char *end = nullptr;
			const long a = strtol(p, &end, 10);
			if (end == p) break;
			long b = a;
			if (*end == '-') { p = end + 1; b = strtol(p, &end, 10); }
			for (long c = a; c <= b; ++c) list.push_back(int(c));
			if (*end != ',') break;
			p = end + 1;
		}
		return list;
	}
	std::atomic<int> wanted_node{-1};
public:
	void prefer_node_of(const void *p) {
		static const bool off = [] { const char *e = getenv("DROPEST_DECODE_NUMA"); return e && atoi(e) == 0; }();
		if (off || !p) return;
		const int node = numa_node_of(p);
		if (node >= 0) wanted_node.store(node, std::memory_order_relaxed);
	}
private:
	DecodePool() {
		unsigned n = 14;
		if (const char *e = getenv("DROPEST_DECODE_THREADS")) n = unsigned(std::max(1, atoi(e)));
		const unsigned hw = std::thread::hardware_concurrency();
		if (hw > 2) n = std::min(n, hw - 2); else n = 1;
		for (unsigned t = 0; t < n; ++t)
			threads.emplace_back([this] {
				(void)prctl(PR_SET_TIMERSLACK, 1000ul, 0ul, 0ul, 0ul);   // the 20 us naps of idle() mean 20 us (the default slack adds 50)
				uint64_t want = 0;
				int device = -1, node = -1;
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
					const int wn = wanted_node.load(std::memory_order_relaxed);
					if (wn != node && wn >= 0) {
						node = wn;
						const std::vector<int> cpus = cpus_of_node(node);
						cpu_set_t set;
						CPU_ZERO(&set);
						for (int c : cpus) if (c < CPU_SETSIZE) CPU_SET(c, &set);
						if (cpus.size() >= 4) (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
					}
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
