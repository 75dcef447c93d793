// k_inflate_par.h -- raw DEFLATE of BGZF blocks with the 64 lanes of a wave decoding DIFFERENT parts of one block's symbol stream.
//
// k_inflate.h gives a block a wave and runs ONE chain of symbols through it: every lane computes the same value, ~40 instructions per
// symbol, and a block of a file that deflates 3 x (a real 10x BAM: ~35 000 symbols, nine of ten literals) lasts 7-11 ms -- 50 GB/s with the
// device full, 0.6 % of the HBM peak, and the window granularity of the BAM path (NOTES_r05 section 13, NOTES_r06 section 3).  Here the body
// of a Huffman-coded DEFLATE block is cut into spans of 64 chunks of INFP_CHUNK_BITS input bits, one chunk per lane:
//   (A) every lane scans its chunk from a guessed start (the chunk's first bit; lane 0 from the true position): symbols counted, output bytes
//       and matches counted, the bit position where it leaves the chunk noted.  A prefix code re-synchronises after a few symbols, so
//       most lanes leave their chunk at a TRUE symbol boundary whatever their start was.  Lane i + 1 then takes lane i's exit as its start and
//       scans again if that changed; repeated until no start changes (lane k is certain after k + 1 rounds; in practice two or three rounds).
//   (B) prefix sums of the lanes' output bytes and matches place every lane in the output and in the span's match list;
//   (C) every lane decodes its chunk once more: literals go to their bytes of the output, matches (destination, length, distance) to the list;
//   (D) the matches are copied in order, 64 bytes per step by all lanes (a match may read what the match before it wrote).
// The end-of-block symbol ends a span early (the lanes behind it are dropped); block headers, the Huffman tables (ballots), stored blocks,
// ISIZE and the CRC-32 check are k_inflate.h's.  Written from RFC 1951; replaces the same BamTools code (BgzfStream::InflateBlock under
// BamReader::GetNextAlignment, Estimation/BamProcessing/BamController.cpp:85).  Integer work, no MFMA.
#pragma once

#include "k_inflate.h"

namespace dropest {

constexpr int INFP_WAVES = 4;                       // waves per workgroup
#ifndef INFP_CHUNK
#define INFP_CHUNK 512
#endif
constexpr uint32_t INFP_CHUNK_BITS = INFP_CHUNK;     // input bits per lane and span
constexpr uint32_t INFP_MATCH_CAP = INFP_CHUNK_BITS * 64 / 2;   // a match is at least two bits: what one span can hold
#ifndef INFP_OVERLAP_BITS
#define INFP_OVERLAP_BITS 0
#endif
// A lane's first, guessed walk may start this many bits BEFORE its chunk (by the chunk's first bit it has usually fallen in step with the true
// symbols, and then needs no second walk).  Measured with 96: 93.8 -> 89.5 GB/s on the 3.2 x file -- the longer first walks cost more than the
// second walks they save.  0: off.
constexpr uint32_t INFP_OVERLAP = INFP_OVERLAP_BITS;
constexpr uint32_t INFP_SPAN_WORDS = INFP_CHUNK_BITS + 8;       // 64-bit words of input a span may look at: its 64 chunks, the word its first bit stands in, and a symbol's reach behind its last bit
enum : uint32_t { INFP_NONE = 0, INFP_EOB = 1, INFP_BAD = 2 };
// -DINFP_PROFILE (scripts/experiments/inflate_variants): where a block's time goes, summed over the launch in 10 ns ticks (wall_clock64) by lane 0 of
// every wave -- 0 header + tables, 1 span input to LDS, 2 (A), 3 (B), 4 (C), 5 (D), 6 CRC-32; counts: 7 rounds of (A), 8 spans, 11 BGZF blocks, 12 matches; inside (D):
// 9 window moved and filled, 10 the rounds, 13 the batch to memory, 14 rounds, 15 batches; 17 rounds of (A) in which some lane walked.  (-DINFP_PROFILE_LOOP instead of the (D) slots: 13 iterations of the symbol
// loop per wave, 14 / 15 of them with a literal-length / distance code beyond the root table in some lane -- an atomic per iteration, the times mean nothing then.)  dropest_bgzf_inflate_profile() reads and clears them.
#ifdef INFP_PROFILE
__device__ unsigned long long infp_prof[24];
#define INFP_TICK(t) const uint64_t t = wall_clock64()
#define INFP_SUM(acc, t) acc += wall_clock64() - (t)
#define INFP_DECL(acc) uint64_t acc = 0
#define INFP_ACC(slot, t) do { if (lane == 0) atomicAdd(&infp_prof[slot], (unsigned long long)(wall_clock64() - (t))); } while (0)
#define INFP_CNT(slot, v) do { if (lane == 0) atomicAdd(&infp_prof[slot], (unsigned long long)(v)); } while (0)
#else
#define INFP_TICK(t) do { } while (0)
#define INFP_SUM(acc, t) do { } while (0)
#define INFP_DECL(acc) do { } while (0)
#define INFP_ACC(slot, t) do { } while (0)
#define INFP_CNT(slot, v) do { } while (0)
#endif

struct InfpMatch { uint32_t dst; uint16_t len, dist; };   // dst: offset in the block's output
static_assert(sizeof(InfpMatch) == 8, "match record");

// Codes longer than the root table's index.  A canonical code read most-significant-bit first and padded to 15 bits lies below
// limit[l] = (first code of length l + number of codes of length l) << (15 - l) exactly when it is at most l bits long, and limit[] does not
// fall as l grows: the length is the smallest l with r < limit[l], the symbol sym[base[l] + (r >> (15 - l))] with base[l] = (codes shorter
// than l) - (first code of length l).  One entry per length, limit in the low half and base in the high half; made from count[] (inf_build) by
// the first 16 lanes.  (k_inflate.h walks the lengths bit by bit from 1, a load per step: with 64 lanes on different symbols some lane was on
// that path in a third of the iterations -- measured -- and every lane waited for its 11 to 15 steps.)
struct InfpLong { uint32_t lit[16], dst[16]; };

__device__ inline void infp_long_table(const uint16_t *count, uint32_t *lc, uint32_t lane) {
	if (lane < 16u) {
		uint32_t first = 0, shorter = 0;
		for (uint32_t l = 1; l <= lane; ++l) { const uint32_t below = l > 1u ? uint32_t(count[l - 1u]) : 0u; first = (first + below) << 1; shorter += below; }
		const uint32_t limit = lane ? (first + uint32_t(count[lane])) << (15u - lane) : 0u;      // (<= 0x8000: the code is not over-subscribed)
		lc[lane] = limit | ((shorter - first) & 0xFFFFu) << 16;
	}
}

// one symbol of a canonical code out of the low bits of buf: the root table, else the lengths beyond it (above)
__device__ inline uint32_t infp_sym(uint64_t buf, const uint16_t *root, uint32_t root_bits, const uint32_t *lc, const uint16_t *sym, uint32_t &used) {
	const uint32_t e = root[uint32_t(buf) & ((1u << root_bits) - 1u)];
	if (e) { used = e & 15u; return e >> 4; }
#ifdef INFP_PROFILE_LOOP
	if (__lane_id() == uint32_t(__builtin_ctzll(__ballot(1)))) atomicAdd(&infp_prof[root_bits > 8u ? 14 : 15], 1ull);
#endif
	const uint32_t r = __brev(uint32_t(buf)) >> 17;
	uint32_t len = 16u, ent = 0;
	for (uint32_t l = 15u; l > root_bits; --l) { const uint32_t c = lc[l]; if (r < (c & 0xFFFFu)) { len = l; ent = c; } }      // (independent loads; the smallest such l stays)
	if (len > 15u) { used = 1; return 0xFFFFu; }
	used = len;
	return sym[((ent >> 16) + (r >> (15u - len))) & 0xFFFFu];
}

// DEFLATE's length and distance codes without their tables: base and extra bits by arithmetic (a table lookup is an LDS round trip on the path
// every lane of the wave waits for)
__device__ inline uint32_t infp_len_base(uint32_t k, uint32_t &extra) {      // k = symbol - 257, 0 .. 28
	if (k < 8u) { extra = 0; return 3u + k; }
	if (k == 28u) { extra = 0; return 258u; }
	extra = (k >> 2) - 1u;
	return 3u + ((4u + (k & 3u)) << extra);
}
__device__ inline uint32_t infp_dist_base(uint32_t d, uint32_t &extra) {     // d = 0 .. 29
	if (d < 4u) { extra = 0; return 1u + d; }
	extra = (d >> 1) - 1u;
	return 1u + ((2u + (d & 1u)) << extra);
}

// the low n (< 8) bytes of v to global memory at p, any alignment
__device__ inline void infp_put_tail_global(uint8_t *p, uint64_t v, uint32_t n) {
	if (n & 4u) { const uint32_t x = uint32_t(v); __builtin_memcpy(p, &x, 4); p += 4; v >>= 32; }
	if (n & 2u) { const uint16_t x = uint16_t(v); __builtin_memcpy(p, &x, 2); p += 2; v >>= 16; }
	if (n & 1u) *p = uint8_t(v);
}

// Between two rounds of (D): this wave's stores before this wave's loads.  The vector memory operations of ONE wave are performed in order (a wavefront-scope
// fence is a matter for the compiler only); -DINFP_ROUND_FENCE_WG: the workgroup-scope fence of the first version, which also waits for the stores' acknowledgements.
#ifdef INFP_ROUND_FENCE_WG
#define INFP_ROUND_FENCE() __threadfence_block()
#else
#define INFP_ROUND_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#endif

// A lane's walk over [start, stop) of the block body (bit offsets relative to base_bits).  EMIT = false: counts only.  EMIT = true: literals
// to out[obyte ...] (gathered eight to a store), matches to list[mslot ...] (their destinations are offsets in the block's output).  Returns the
// flag; rel = where it stands.  The input: three words of the span in registers, the next one loaded when the position crosses a word -- long
// before it is looked at -- so that a symbol costs one LDS round trip (its root entry), a match two.
template <bool EMIT>
__device__ inline uint32_t infp_walk(const uint64_t *span, uint32_t span_bit0, uint32_t start, uint32_t count_from, uint32_t &entry, uint32_t stop, uint32_t limit,
                                     const InfWaveLds &L, const InfpLong &X, uint32_t &rel_out, uint32_t &n_bytes, uint32_t &n_match,
                                     uint8_t *__restrict__ out, uint32_t obyte, InfpMatch *__restrict__ list, uint32_t mslot, uint32_t &bad_dist, uint32_t own_bytes = 0) {
	uint32_t rel = start, nb = 0, nm = 0, flag = INFP_NONE;
	uint32_t W = (span_bit0 + rel) >> 6;
	uint64_t w0 = span[W], w1 = span[W + 1u], w2 = span[W + 2u];
	uint64_t acc = 0;      // EMIT: the literals not yet stored, the oldest in the low byte
	uint32_t na = 0;
	entry = 0xFFFFFFFFu;      // the first symbol boundary at or behind count_from: what is counted starts there
	for (uint32_t steps = 0; rel < stop; ++steps) {
		if (steps > INFP_CHUNK_BITS + INFP_OVERLAP + 64u) { flag = INFP_BAD; break; }      // (every symbol takes a bit: never reached)
		if (entry == 0xFFFFFFFFu && rel >= count_from) { entry = rel; nb = 0; nm = 0; }
#ifdef INFP_PROFILE_LOOP
		if (__lane_id() == uint32_t(__builtin_ctzll(__ballot(1)))) atomicAdd(&infp_prof[13], 1ull);
#endif
		const uint32_t p = span_bit0 + rel;
		if ((p >> 6) != W) { w0 = w1; w1 = w2; ++W; w2 = span[W + 2u]; }      // (a symbol is at most 48 bits: one word at a time)
		const uint32_t sh = p & 63u;
		uint64_t buf = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;
		uint32_t used;
		const uint32_t sy = infp_sym(buf, L.lroot, INF_LROOT, X.lit, L.lsym, used);
		buf >>= used; rel += used;
		if (sy < 256u) {
			if (EMIT) {
				acc |= uint64_t(sy) << (8u * na);
				if (++na == 8u) { __builtin_memcpy(out + obyte + nb - 7u, &acc, 8); acc = 0; na = 0; }
			}
			++nb;
		} else if (sy == 256u) { flag = INFP_EOB; break; }
		else if (sy > 285u) { flag = INFP_BAD; break; }
		else {
			uint32_t le, de;
			uint32_t len = infp_len_base(sy - 257u, le);
			len += uint32_t(buf) & ((1u << le) - 1u); buf >>= le; rel += le;
			const uint32_t ds = infp_sym(buf, L.droot, INF_DROOT, X.dst, L.dsym, used);
			buf >>= used; rel += used;
			if (ds > 29u) { flag = INFP_BAD; break; }
			uint32_t dist = infp_dist_base(ds, de);
			dist += uint32_t(buf) & ((1u << de) - 1u); rel += de;
			if (EMIT) {
				// the literals in front of the match: a whole word when the eight bytes all lie in this lane's own part of the output (what it writes
				// beyond the literals is the match's place, filled in (D), and bytes this lane writes itself later), else exactly
				if (na) {
					if (nb - na + 8u <= own_bytes) __builtin_memcpy(out + obyte + nb - na, &acc, 8);
					else infp_put_tail_global(out + obyte + nb - na, acc, na);
					acc = 0; na = 0;
				}
				if (dist > obyte + nb) bad_dist = 1u;
				list[mslot + nm] = InfpMatch{obyte + nb, uint16_t(len), uint16_t(dist)};
			}
			nb += len; ++nm;
		}
		if (rel > limit) { flag = INFP_BAD; break; }
	}
	if (EMIT && na) infp_put_tail_global(out + obyte + nb - na, acc, na);
	rel_out = rel; n_bytes = nb; n_match = nm;
	return flag;
}

// the low n (< 8) bytes of v to p (LDS, any alignment): four, two, one -- never a byte beyond, the next match's place belongs to another lane
__device__ inline void infp_put_tail(uint8_t *p, uint64_t v, uint32_t n) {
	if (n & 4u) { const uint32_t x = uint32_t(v); __builtin_memcpy(p, &x, 4); p += 4; v >>= 32; }
	if (n & 2u) { const uint16_t x = uint16_t(v); __builtin_memcpy(p, &x, 2); p += 2; v >>= 16; }
	if (n & 1u) *p = uint8_t(v);
}

__device__ inline uint32_t infp_excl_scan(uint32_t v, uint32_t lane, uint32_t &total) {
	uint32_t x = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const uint32_t o = uint32_t(__shfl_up(int(x), d, 64)); if (lane >= uint32_t(d)) x += o; }
	total = uint32_t(__shfl(int(x), 63, 64));
	return x - v;
}

// The body of one Huffman-coded DEFLATE block from absolute bit position `body` on: output appended at out + pos.  Returns the error (INF_OK ...),
// body = the bit behind the end-of-block symbol, pos advanced.  All 64 lanes; the tables of the block are in L.
__device__ inline uint32_t infp_block_body(const uint64_t *__restrict__ gin, uint64_t in_words, uint64_t &body, uint64_t in_end_bits, const InfWaveLds &L,
                                           const InfpLong &X, uint8_t *__restrict__ out, uint32_t out_cap, uint32_t &pos,
                                           InfpMatch *__restrict__ list, uint64_t *span, uint32_t lane, uint32_t dbg = 0) {
	for (uint32_t spans = 0;; ++spans) {
		if (spans > 4096u) return INF_BAD_CODE;                           // (64 KB of output are at most a few hundred spans)
		const uint64_t base = body;
		// (bits the block may still read: to the end of the BGZF block's payload and a word beyond -- a damaged stream must not walk into its neighbours)
		if (base > in_end_bits + 64u) return INF_INPUT_OVERRUN;
		const uint64_t left = in_end_bits + 64u - base;
		const uint32_t limit = left > 0xFFFFFFF0ull ? 0xFFFFFFF0u : uint32_t(left);
		// the span's input into LDS: the words from the one `base` stands in on (zeros beyond the buffer)
		const uint64_t word0 = base >> 6;
		const uint32_t bit0 = uint32_t(base & 63u);
		INFP_TICK(t_in);
		for (uint32_t i = lane; i < INFP_SPAN_WORDS; i += 64u) span[i] = word0 + i < in_words ? gin[word0 + i] : 0ull;
		__threadfence_block();
		INFP_ACC(1, t_in); INFP_CNT(8, 1);
		INFP_TICK(t_a);
		// (A) starts: the chunk's first bit (lane 0: the true position), then every lane takes its predecessor's exit until nothing moves
		// (round 0: from INFP_OVERLAP bits before the chunk, counting from the first boundary inside it -- when that boundary is where the lane
		// before leaves ITS chunk, which it mostly is, the lane needs no second walk)
		uint32_t start = lane * INFP_CHUNK_BITS, end = 0, nb = 0, nm = 0, flag = INFP_NONE, dummy = 0, entry = 0;
		bool alive = true, changed = true;
		uint32_t walk_rounds = 0;
		for (uint32_t round = 0; round < 66u; ++round) {
			INFP_CNT(7, 1);
			if (__ballot(changed && alive)) ++walk_rounds;      // (rounds in which the lanes behind an end-of-block symbol drop out one after the other cost nothing)
			if (changed && alive) {
				const uint32_t from = round == 0u && lane ? start - INFP_OVERLAP : start;
				flag = infp_walk<false>(span, bit0, from, start, entry, (lane + 1u) * INFP_CHUNK_BITS, limit, L, X, end, nb, nm, nullptr, 0u, nullptr, 0u, dummy);
				if (round == 0u && lane) start = entry;      // (0xFFFFFFFF: the guess ran out before the chunk began -- the lane before will say where to start)
			}
			const uint32_t p_end = uint32_t(__shfl_up(int(end), 1, 64)), p_flag = uint32_t(__shfl_up(int(flag), 1, 64));
			const bool p_alive = __shfl_up(int(alive), 1, 64) != 0;
			changed = false;
			if (lane) {
				const bool now_alive = p_alive && p_flag == INFP_NONE;
				changed = now_alive != alive || (now_alive && p_end != start);
				alive = now_alive; start = p_end;
			}
			if (!alive) { nb = 0; nm = 0; flag = INFP_NONE; }
			const unsigned long long moved = __ballot(changed);
			if (!moved) break;
			// The lanes below the first one that moved are final (a lane's start hangs on the lanes before it only).  An end-of-block symbol in one of
			// them ends the span THERE: the lanes behind it walk garbage, and kept the wave walking -- one more of them dropped out per round -- for up to
			// 64 rounds in the last span of every block.
			{
				const unsigned long long final_eob = __ballot(alive && flag == INFP_EOB) & ((1ull << __builtin_ctzll(moved)) - 1ull);
				if (final_eob) {
					if (int(lane) > __builtin_ctzll(final_eob)) { alive = false; nb = 0; nm = 0; flag = INFP_NONE; }
					changed = false;
					break;
				}
			}
			// (Codes of nearly one length never fall in step -- six symbols of 2-3 bits, Huffman-only streams: lane k is then certain only after k + 1
			// rounds, 64 full walks per span, 17-28 ms a block against the 7-11 of one chain of symbols.  Handing such a block's rest to k_inflate.h's
			// serial loop after 12 / 24 / 40 walking rounds was built and measured: 9 % of the blocks of a BAM with random bases have a span that takes
			// 40 rounds and more -- and finish it sooner than the serial loop finishes the rest of the block: 115 -> 70 GB/s.  Dropped; DROPEST_INFLATE_PAR=0
			// is the switch for files of that kind.)
		}
		INFP_ACC(2, t_a); INFP_CNT(17, walk_rounds); (void)walk_rounds;
		if (dbg == 3) return 203u;
		if (__ballot(changed)) return INF_BAD_CODE;                       // (cannot happen: lane k is settled after k + 1 rounds)
		if (__ballot(alive && flag == INFP_BAD)) return INF_BAD_CODE;
		// (B) places
		uint32_t tot_b, tot_m;
		INFP_TICK(t_b);
		const uint32_t ob = infp_excl_scan(alive ? nb : 0u, lane, tot_b), om = infp_excl_scan(alive ? nm : 0u, lane, tot_m);
		if (pos + tot_b > out_cap || pos + tot_b < pos) return INF_OUTPUT_OVERRUN;
		if (tot_m > INFP_MATCH_CAP) return INF_BAD_CODE;
		INFP_ACC(3, t_b); INFP_CNT(12, tot_m);
		if (dbg == 4) return 204u;
		// (C) literals and the match list
		uint32_t bad_dist = 0;
		INFP_TICK(t_c);
		if (alive && dbg != 12) {      // (dbg 11 / 12: timing probes -- no match copies / no second walk either; the output is wrong then)
			uint32_t e2, b2, m2;
			(void)infp_walk<true>(span, bit0, start, start, dummy, (lane + 1u) * INFP_CHUNK_BITS, limit, L, X, e2, b2, m2, out, pos + ob, list, om, bad_dist, nb);
		}
		if (__ballot(bad_dist != 0u)) return INF_BAD_DISTANCE;
		INFP_ACC(4, t_c);
		if (dbg == 5) return 205u;
		INFP_TICK(t_d);
		// (D) the matches: 64 at a time, a lane per match.  Everything before the destination of the first match that is still waiting is final
		// (literals are written, earlier matches are copied), so every waiting match whose source ends there or earlier can be copied NOW, side by
		// side -- a BAM's matches reach a record back (~270 bytes, eight matches or so), which is eight matches per round instead of one.
		// The rounds of a block are as many as its chains of matches are deep (a record copies from the record before: ~240 rounds per block), and
		// a round through memory lasts as long as a load that follows a store (0.7-2 us: 600 us of a block's 2 ms with the device empty, 1 450 of
		// 3 100 with it full).  So the rounds run in LDS: the span's input is done with after (C), and its 4 KB hold a WINDOW of the output --
		// bytes [wb, wb + WIN) of the block, the batch's destinations and what lies before them.  A batch loads the part of its range the window
		// does not hold yet (literals of (C); the matches' own places hold anything), the rounds read and write the window (sources in front of it,
		// final long ago, come from memory), and the batch's range goes to memory in whole words at the end.  Long matches: fewer than 64 to a batch.
		__threadfence_block();
		uint8_t *const ring = reinterpret_cast<uint8_t *>(span);
		constexpr uint32_t WIN = INFP_CHUNK_BITS * 8u, KEEP = WIN / 4u;      // bytes of window; history a slide keeps at least
		static_assert(WIN >= 1024u && WIN <= INFP_SPAN_WORDS * 8u, "a match of 258 bytes and its word edges fit the window many times");
		uint32_t wb = 0, ring_hi = 0;
		bool ring_on = false;

		InfpMatch cur{0u, 0, 0};
		if (lane < tot_m) cur = list[lane];
		uint32_t n_rounds = 0, n_batches = 0;
		INFP_DECL(ticks_window); INFP_DECL(ticks_rounds); INFP_DECL(ticks_flush);
		for (uint32_t b0 = 0, n_take = 0; b0 < (dbg == 11 || dbg == 12 ? 0u : tot_m); b0 += n_take) {
			bool have = b0 + lane < tot_m;
			const InfpMatch mine = cur;
			InfpMatch nxt{0u, 0, 0};                  // the 64 matches after these: on their way while the rounds run
			if (b0 + 64u + lane < tot_m) nxt = list[b0 + 64u + lane];
			const uint32_t len = mine.len, dist = mine.dist;
			const uint32_t src_end = mine.dst - dist + (len < dist ? len : dist);      // the bytes of earlier output the match reads end here
			// the batch: the matches of these 64 whose destinations end inside one window's length from the first one's (long matches: fewer than 64)
			const uint32_t lo = uint32_t(__shfl(int(mine.dst), 0, 64)), lo8 = lo & ~7u;
			have = have && mine.dst + len - lo8 <= WIN - 8u;
			n_take = uint32_t(__builtin_popcountll(__ballot(have)));       // (destinations ascend: a prefix; at least the first)
			if (!n_take) return INF_BAD_CODE;
			const uint32_t hi = uint32_t(__shfl(int(mine.dst + len), int(n_take - 1u), 64)), hi8 = (hi + 7u) & ~7u;
			++n_batches;
			INFP_TICK(t_d1);
			if (!ring_on) { wb = lo8; ring_hi = lo8; ring_on = true; }
			if (hi8 - wb > WIN) {      // the window moves up: its last bytes (KEEP of them at least, more if there is room) slide to its start
				uint32_t nwb = lo8 > KEEP ? lo8 - KEEP : 0u;
				if (nwb < hi8 - WIN) nwb = hi8 - WIN;
				if (nwb < wb) nwb = wb;
				if (ring_hi > nwb) {
					const uint32_t shift = (nwb - wb) >> 3, n_words = (ring_hi - nwb) >> 3;
					for (uint32_t i = lane; i < ((n_words + 63u) & ~63u); i += 64u) {      // (a pass reads ahead of what it writes, and all of it before it writes)
						uint64_t v = 0;
						if (i < n_words) v = span[i + shift];
						INFP_ROUND_FENCE();
						if (i < n_words) span[i] = v;
					}
				} else ring_hi = nwb;
				wb = nwb;
			}
			// (asking for the next kilobyte of these words a batch ahead, to land when the next batch begins, changed nothing: 111.9 -> 111.5 GB/s --
			// the kernel waits for its vector ALUs, 74 % busy, not for these loads)
			for (uint32_t p = ring_hi + lane * 8u; p < hi8; p += 512u) {      // (the block's own bytes only: the last word byte by byte)
				uint64_t v = 0;
				if (p + 8u <= out_cap) __builtin_memcpy(&v, out + p, 8);
				else for (uint32_t k = 0; p + k < out_cap; ++k) v |= uint64_t(out[p + k]) << (8u * k);
				span[(p - wb) >> 3] = v;
			}
			ring_hi = hi8;
			INFP_ROUND_FENCE();
			INFP_SUM(ticks_window, t_d1);
			INFP_TICK(t_d2);
			// which matches of the batch must be copied before this one: those whose destination holds a byte of this one's source [q0, src_end).
			// Destinations ascend and do not overlap, so they are the lanes a .. b - 1 with a = destinations that end at or before q0, b = destinations
			// that begin before src_end (all of them lanes below this one): two binary searches over the lanes.  (Until round 6e a match waited for
			// EVERY match before it that was still waiting -- the first one's destination was the frontier -- and a run behind a literal, whose source
			// is that literal, took a round of its own: 5.8 rounds per batch instead of the 2-3 that the records' chains are deep.)
			const uint32_t q0 = mine.dst - dist;
			unsigned long long before = 0;
			{
				const uint32_t my_dst = have ? mine.dst : 0xFFFFFFFFu, my_end = have ? mine.dst + len : 0xFFFFFFFFu;
				uint32_t a = 0, b = 0;
#pragma unroll
				for (uint32_t step = 32u; step; step >>= 1) {
					const uint32_t ea = uint32_t(__shfl(int(my_end), int(a + step - 1u), 64)), db = uint32_t(__shfl(int(my_dst), int(b + step - 1u), 64));
					if (ea <= q0) a += step;
					if (db < src_end) b += step;
				}
				if (b > a) before = ((1ull << (b - a)) - 1ull) << a;
			}
			bool waiting = have;
			for (unsigned long long w = __ballot(waiting); w; w = __ballot(waiting)) {
				++n_rounds;
				if (waiting && !(w & before)) {
					uint8_t *const d = ring + (mine.dst - wb);
					if (q0 >= wb) {                           // in the window: words (a read of eight bytes may reach beyond the source, never beyond the window)
						const uint8_t *const s = ring + (q0 - wb);
						uint64_t a; __builtin_memcpy(&a, s, 8);
						uint32_t i = 0;
						if (dist >= 8u) {
							for (; i + 8u <= len; i += 8u) { __builtin_memcpy(d + i, &a, 8); __builtin_memcpy(&a, s + i + 8u, 8); }      // (never reaches what this match has not written yet)
						} else {                                  // a run: its period filled up to a word, laid down at every multiple of the period that a word holds
							a &= (1ull << (8u * dist)) - 1ull;
							a |= a << (8u * dist);
							if (dist < 4u) a |= a << (16u * dist);
							if (dist < 2u) a |= a << 32;
							const uint32_t stride = (0x7658688u >> (4u * (dist - 1u))) & 15u;      // (8 / dist) * dist for dist = 1 .. 7
							for (; i + 8u <= len; i += stride) __builtin_memcpy(d + i, &a, 8);
						}
						if (i < len) infp_put_tail(d + i, a, len - i);
					} else if (src_end <= wb && dist >= len) {   // in front of the window: from memory
						const uint8_t *const s = out + q0;
						uint32_t i = 0;
						for (; i + 8u <= len; i += 8u) { uint64_t a; __builtin_memcpy(&a, s + i, 8); __builtin_memcpy(d + i, &a, 8); }
						for (; i < len; ++i) d[i] = s[i];
					} else {                                  // across the window's start
						for (uint32_t i = 0, j = 0; i < len; ++i) { const uint32_t q = q0 + j; d[i] = q >= wb ? ring[q - wb] : out[q]; if (++j == dist) j = 0; }
					}
					waiting = false;
				}
				INFP_ROUND_FENCE();
			}
			INFP_SUM(ticks_rounds, t_d2);
			INFP_TICK(t_d3);
			// the batch's range to memory, whole words (what lies behind `hi` in the last word is what memory held: literals, or the place of a
			// match of the next batch, which writes it again); never beyond the block's own bytes
			for (uint32_t p = lo8 + lane * 8u; p < hi8; p += 512u) {
				const uint64_t v = span[(p - wb) >> 3];
				if (p + 8u <= out_cap) __builtin_memcpy(out + p, &v, 8);
				else for (uint32_t k = 0; p + k < out_cap; ++k) out[p + k] = uint8_t(v >> (8u * k));
			}
			INFP_ROUND_FENCE();
			INFP_SUM(ticks_flush, t_d3);
			if (n_take == 64u) cur = nxt;
			else { cur = InfpMatch{0u, 0, 0}; if (b0 + n_take + lane < tot_m) cur = list[b0 + n_take + lane]; }
		}
		INFP_ACC(5, t_d); INFP_CNT(14, n_rounds); INFP_CNT(15, n_batches); (void)n_rounds; (void)n_batches;
		INFP_CNT(9, ticks_window); INFP_CNT(10, ticks_rounds); INFP_CNT(13, ticks_flush);
		if (dbg == 6) return 206u;
		pos += tot_b;
		// where the span ends: behind the end-of-block symbol, or at the last lane's exit
		const unsigned long long eob = __ballot(alive && flag == INFP_EOB);
		const int last = eob ? __builtin_ctzll(eob) : 63;
		const uint32_t span_end = uint32_t(__shfl(int(end), last, 64));
		body = base + span_end;
		if (eob) return INF_OK;
		if (!span_end) return INF_BAD_CODE;                               // (no progress: cannot happen, every symbol takes a bit)
	}
}

// One BGZF block per wave at a time; the waves of the launch take blocks from a counter (*next_block, 0 at launch) until none is left, so that
// a wave's match list can live in a slot of `scratch` that belongs to it (INFP_MATCH_CAP records per wave of the grid).
#ifndef INFP_WAVES_PER_EU
#define INFP_WAVES_PER_EU 4      // 128 vector registers (18 spilled) instead of 153: four waves per SIMD instead of three
#endif
__global__ __launch_bounds__(INFP_WAVES * 64) __attribute__((amdgpu_waves_per_eu(INFP_WAVES_PER_EU, INFP_WAVES_PER_EU))) void bgzf_inflate_par_kernel(const uint8_t *__restrict__ d_in, uint64_t in_total_len, const uint64_t *__restrict__ in_off,
                                                                           const uint32_t *__restrict__ in_len, const uint64_t *__restrict__ out_off,
                                                                           const uint32_t *__restrict__ out_len, uint32_t n_blocks, uint8_t *d_out,
                                                                           uint32_t *__restrict__ status, const uint32_t *__restrict__ crc32, InfpMatch *__restrict__ scratch,
                                                                           uint32_t *next_block, uint32_t dbg) {
	__shared__ InfWaveLds lds[INFP_WAVES];
	__shared__ uint32_t crc_tab[4 * 256];
	__shared__ uint32_t crc_x2n[INFP_WAVES][32];
	__shared__ InfpLong longs[INFP_WAVES];
	__shared__ uint64_t span_in[INFP_WAVES][INFP_SPAN_WORDS];
	if (crc32) {
		uint32_t c = threadIdx.x & 255u;
		for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ INF_CRC_POLY : c >> 1;
		crc_tab[threadIdx.x & 255u] = c;                  // (INFP_WAVES * 64 = 256 threads: one entry of every table each)
		__syncthreads();
		for (int t = 1; t < 4; ++t) { c = (c >> 8) ^ crc_tab[c & 0xFFu]; crc_tab[t * 256 + (threadIdx.x & 255u)] = c; __syncthreads(); }
	}
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	InfWaveLds &L = lds[wave];
	InfpMatch *const list = scratch + size_t(blockIdx.x * INFP_WAVES + wave) * INFP_MATCH_CAP;
	const uint64_t *const gin = reinterpret_cast<const uint64_t *>(d_in);
	const uint64_t in_words = (in_total_len + 7) >> 3;
	for (;;) {
		uint32_t blk = 0;
		if (lane == 0) blk = atomicAdd(next_block, 1u);
		blk = uint32_t(__shfl(int(blk), 0, 64));
		if (blk >= n_blocks) return;
		if (dbg == 1) { if (lane == 0) status[blk] = 201u; continue; }
		InfState s;
		s.gin = gin; s.in_words = in_words;
		const uint64_t in_begin = in_off[blk], in_end = in_begin + in_len[blk];
		s.ipos = in_begin; s.loaded_hi = in_begin & ~uint64_t(255); s.bits = 0; s.cnt = 0;
		uint8_t *const out = d_out + out_off[blk];
		const uint32_t out_cap = out_len[blk];
		uint32_t pos = 0, err = INF_OK;
		uint32_t n_deflate_blocks = 0;
		INFP_CNT(11, 1);
		for (bool last = false; !last && !err;) {
			INFP_TICK(t_h);
			if (s.ipos - uint64_t(s.cnt >> 3) > in_end + 8) { err = INF_INPUT_OVERRUN; break; }
			if (++n_deflate_blocks > 70000u) { err = INF_BAD_BLOCK_TYPE; break; }
			last = inf_take(s, L, lane, 1) != 0;
			const uint32_t type = inf_take(s, L, lane, 2);
			if (type == 0) {   // stored: to the byte boundary, LEN, NLEN, bytes
				const int drop = s.cnt & 7;
				s.bits >>= drop; s.cnt -= drop;
				const uint32_t len = inf_take(s, L, lane, 16), nlen = inf_take(s, L, lane, 16);
				if ((len ^ nlen) != 0xFFFFu) { err = INF_BAD_STORED; break; }
				const uint64_t from = s.ipos - uint64_t(s.cnt >> 3);
				if (from + len > in_end) { err = INF_INPUT_OVERRUN; break; }
				if (pos + len > out_cap) { err = INF_OUTPUT_OVERRUN; break; }
				for (uint32_t i = lane; i < len; i += 64) out[pos + i] = d_in[from + i];
				pos += len;
				s.ipos = from + len; s.bits = 0; s.cnt = 0; s.loaded_hi = s.ipos & ~uint64_t(255);
				continue;
			}
			if (type == 3) { err = INF_BAD_BLOCK_TYPE; break; }
			int hlit, hdist;
			if (type == 1) {
				hlit = 288; hdist = 30;
				for (uint32_t i = lane; i < 288u; i += 64) L.lens[i] = i < 144u ? 8 : (i < 256u ? 9 : (i < 280u ? 7 : 8));
				if (lane < 30u) L.lens[288u + lane] = 5;
			} else {
				hlit = int(inf_take(s, L, lane, 5)) + 257; hdist = int(inf_take(s, L, lane, 5)) + 1;
				const int hclen = int(inf_take(s, L, lane, 4)) + 4;
				if (hlit > 286 || hdist > 30) { err = INF_BAD_LENGTHS; break; }
				if (lane < 19u) L.lens[lane] = 0;
				for (int i = 0; i < hclen; ++i) {
					const uint32_t v = inf_take(s, L, lane, 3);
					if (lane == 0) L.lens[INF_CL_ORDER[i]] = uint8_t(v);
				}
				if (!inf_build(L.lens, 19, 7, L.droot, L.dsym, L.dcount, lane)) { err = INF_OVERSUBSCRIBED; break; }
				int have = 0;
				uint32_t prev = 0;
				const int total = hlit + hdist;
				while (have < total && !err) {
					const uint32_t sy = inf_decode(s, L, lane, L.droot, 7, L.dsym, L.dcount);
					uint32_t rep = 1, val = sy;
					if (sy < 16u) { prev = sy; }
					else if (sy == 16u) { if (!have) { err = INF_BAD_LENGTHS; break; } rep = 3 + inf_take(s, L, lane, 2); val = prev; }
					else if (sy == 17u) { rep = 3 + inf_take(s, L, lane, 3); val = 0; prev = 0; }
					else if (sy == 18u) { rep = 11 + inf_take(s, L, lane, 7); val = 0; prev = 0; }
					else { err = INF_BAD_CODE; break; }
					if (have + int(rep) > total) { err = INF_BAD_LENGTHS; break; }
					for (uint32_t i = lane; i < rep; i += 64) L.lens[uint32_t(have) + i] = uint8_t(val);
					have += int(rep);
				}
				if (err) break;
				if (inf_uni(L.lens[256]) == 0) { err = INF_BAD_LENGTHS; break; }
			}
			if (!inf_build(L.lens, hlit, INF_LROOT, L.lroot, L.lsym, L.lcount, lane)) { err = INF_OVERSUBSCRIBED; break; }
			if (!inf_build(L.lens + hlit, hdist, INF_DROOT, L.droot, L.dsym, L.dcount, lane)) { err = INF_OVERSUBSCRIBED; break; }
			__threadfence_block();
			infp_long_table(L.lcount, longs[wave].lit, lane);
			infp_long_table(L.dcount, longs[wave].dst, lane);
			__threadfence_block();                        // the tables, before the lanes read them at their own places
			INFP_ACC(0, t_h);
			if (dbg == 2) { err = 202u; break; }
			// the body of the block, by all lanes; then the header reader takes up again behind the end-of-block symbol
			uint64_t body = (s.ipos << 3) - uint64_t(s.cnt);
			err = infp_block_body(gin, in_words, body, in_end << 3, L, longs[wave], out, out_cap, pos, list, span_in[wave], lane, dbg);
			if (err) break;
			s.ipos = body >> 3; s.bits = 0; s.cnt = 0; s.loaded_hi = s.ipos & ~uint64_t(255);
			if (body & 7u) (void)inf_take(s, L, lane, int(body & 7u));
		}
		if (!err) {
			if (pos != out_cap) err = INF_SIZE_MISMATCH;
			else if (s.ipos - uint64_t(s.cnt >> 3) > in_end) err = INF_INPUT_OVERRUN;
		}
		if (dbg == 7 && !err) err = 207u;
		if (!err && crc32) {
			__threadfence_block();
			INFP_TICK(t_crc);
			if (inf_crc32_block(out, out_cap, crc_tab, crc_x2n[wave], lane) != crc32[blk]) err = INF_CRC_MISMATCH;
			INFP_ACC(6, t_crc);
		}
		if (lane == 0) status[blk] = err;
	}
}

}  // namespace dropest
