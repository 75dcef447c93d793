// k_inflate.h -- raw DEFLATE (RFC 1951) of BGZF blocks on the device: one 64-lane wave per block.
//
// What it replaces: BamTools' BgzfStream::InflateBlock (zlib inflate per block) under BamReader::GetNextAlignment, the loop
// Estimation/BamProcessing/BamController.cpp:85 spends its time in.  A BGZF block (SAMv1 §4.1) is an independent
// DEFLATE stream of <= 64 KB, so blocks are the parallel axis: thousands of them are in flight, one per wave.  Inside a block the
// symbol stream is serial -- every code's position depends on the one before -- so the wave decodes with wave-uniform values
// (all lanes compute the same thing, LDS look-ups are broadcasts) and uses its 64 lanes where the format allows:
//   * the input is staged through a 512-byte ring in LDS, 256 bytes per coalesced load;
//   * Huffman tables are built with ballots (rank of a symbol among the symbols of its code length), 64 symbols per step;
//   * literals wait in a register of the lane they will be stored by and leave 64 at a time; a match is copied by all lanes, 64 bytes per step.
// Tables per wave: a 10-bit root table for literal / length codes and an 8-bit one for distances (u16 entries: symbol << 4 | length);
// longer codes (rare symbols) are decoded canonically from the per-length counts, bit by bit (the method of zlib's puff.c).
// Written from RFC 1951.  ISIZE and, when the caller hands over the stored values, the CRC-32 of every block are checked (inf_crc32_block).
// Integer work; no MFMA.  4.0 KB of LDS per wave + 5.4 KB per workgroup (CRC-32 tables, base tables): 4 workgroups of 8 waves per CU = 8 waves
// per SIMD, which the symbol loop (a function of its own: 32 vector registers) can use; the rest of the kernel spills at that budget, off the hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dropest {

constexpr int INF_WAVES = 8;
constexpr int INF_LROOT = 10, INF_DROOT = 8;
// status of a block: 0 = ok; anything else: the block was not (completely) written and the caller inflates it elsewhere
enum : uint32_t { INF_OK = 0, INF_BAD_BLOCK_TYPE = 1, INF_BAD_STORED = 2, INF_BAD_LENGTHS = 3, INF_OVERSUBSCRIBED = 4, INF_BAD_CODE = 5,
                  INF_BAD_DISTANCE = 6, INF_OUTPUT_OVERRUN = 7, INF_INPUT_OVERRUN = 8, INF_SIZE_MISMATCH = 9, INF_CRC_MISMATCH = 10 };

struct InfWaveLds {
	uint16_t lroot[1 << INF_LROOT];
	uint16_t droot[1 << INF_DROOT];
	uint16_t lsym[288];      // symbols in canonical order (by code length, then by symbol)
	uint16_t dsym[32];
	uint16_t lcount[16], dcount[16];
	uint8_t lens[320];       // HLIT + HDIST code lengths
	uint64_t in[64];         // input ring: the 512 bytes around the read position, indexed by (absolute offset / 8) mod 64
};

__constant__ const uint16_t INF_LEN_BASE[32] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 0, 0, 0};
__constant__ const uint8_t INF_LEN_EXTRA[32] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0, 0};
__constant__ const uint16_t INF_DIST_BASE[32] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 0, 0};
__constant__ const uint8_t INF_DIST_EXTRA[32] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 0, 0};
__constant__ const uint8_t INF_CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct InfState {
	const uint64_t *gin;     // the compressed bytes as aligned words
	uint64_t in_words;       // words that may be read (the whole buffer)
	uint64_t ipos;           // absolute byte offset of the next byte to enter `bits`
	uint64_t loaded_hi;      // bytes below this absolute offset (a multiple of 512) are in the ring (the last 1 024 of them)
	uint64_t bits;
	int cnt;
};

// Everything the decoder keeps is the same in all 64 lanes; a value read from LDS is a vector register until it is named uniform:
// then the bit buffer, the positions and the table entries live in scalar registers and the scalar unit does their arithmetic.
__device__ inline uint32_t inf_uni(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }
__device__ inline uint64_t inf_uni64(uint64_t v) { return uint64_t(inf_uni(uint32_t(v))) | (uint64_t(inf_uni(uint32_t(v >> 32))) << 32); }

__device__ inline void inf_ring_fill(InfState &s, InfWaveLds &L, uint32_t lane) {
	while (s.ipos + 16 > s.loaded_hi) {
		const uint64_t w = (s.loaded_hi >> 3) + lane;
		if (lane < 32u) L.in[w & 63u] = w < s.in_words ? s.gin[w] : 0ull;
		s.loaded_hi += 256;
	}
}
__device__ inline void inf_refill(InfState &s, InfWaveLds &L, uint32_t lane) {   // at least 56 valid bits afterwards
	inf_ring_fill(s, L, lane);
	const uint32_t idx = uint32_t(s.ipos >> 3) & 63u, sh = uint32_t(s.ipos & 7u) * 8u;
	const uint64_t lo = inf_uni64(L.in[idx]), hi = inf_uni64(L.in[(idx + 1u) & 63u]);
	const uint64_t w = sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
	s.bits |= w << s.cnt;
	const int adv = (63 - s.cnt) >> 3;
	s.ipos += uint64_t(adv);
	s.cnt += adv * 8;
}
__device__ inline uint32_t inf_take(InfState &s, InfWaveLds &L, uint32_t lane, int n) {   // n <= 16
	if (s.cnt < n) inf_refill(s, L, lane);
	const uint32_t v = uint32_t(s.bits) & ((1u << n) - 1u);
	s.bits >>= n; s.cnt -= n;
	return v;
}

// Canonical Huffman code of lens[0 .. n): root table (codes up to root_bits, indexed by the next input bits, LSB first), the symbols in
// canonical order and the number of codes of every length.  All 64 lanes; n <= 320.  false: over-subscribed code.
__device__ inline bool inf_build(const uint8_t *lens, int n, int root_bits, uint16_t *root, uint16_t *sym, uint16_t *count, uint32_t lane) {
	uint32_t mylen[5];
#pragma unroll
	for (int c = 0; c < 5; ++c) { const int s = c * 64 + int(lane); mylen[c] = s < n ? lens[s] : 0u; }
	for (uint32_t i = lane; i < (1u << root_bits); i += 64) root[i] = 0;
	uint32_t cnt[16];
	cnt[0] = 0;
#pragma unroll
	for (int l = 1; l <= 15; ++l) {
		uint32_t c = 0;
#pragma unroll
		for (int k = 0; k < 5; ++k) c += uint32_t(__popcll(__ballot(mylen[k] == uint32_t(l))));
		cnt[l] = c;
	}
	int left = 1;
#pragma unroll
	for (int l = 1; l <= 15; ++l) { left = (left << 1) - int(cnt[l]); if (left < 0) return false; }
	if (lane < 16) count[lane] = 0;
	uint32_t code = 0, off = 0;
	const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
	for (int l = 1; l <= 15; ++l) {
		code = (code + (l > 1 ? cnt[l - 1] : 0u)) << 1;   // first code of length l
		if (lane == 0) count[l] = uint16_t(cnt[l]);
		uint32_t seen = 0;
#pragma unroll
		for (int k = 0; k < 5; ++k) {
			const bool mine = mylen[k] == uint32_t(l);
			const uint64_t m = __ballot(mine);
			if (mine) {
				const uint32_t rank = seen + uint32_t(__popcll(m & lt));
				const uint32_t s = uint32_t(k) * 64u + lane;
				sym[off + rank] = uint16_t(s);
				if (l <= root_bits) {
					const uint32_t c = code + rank;                       // l bits, most significant first
					const uint32_t r = __brev(c) >> (32 - l);             // as it arrives in the stream
					const uint16_t e = uint16_t((s << 4) | uint32_t(l));
					for (uint32_t j = r; j < (1u << root_bits); j += 1u << l) root[j] = e;
				}
			}
			seen += uint32_t(__popcll(m));
		}
		off += cnt[l];
	}
	return true;
}

// One symbol: root table, or bit by bit for the codes the table does not hold.  Returns the symbol, 0xFFFF = no such code.
__device__ inline uint32_t inf_decode(InfState &s, InfWaveLds &L, uint32_t lane, const uint16_t *root, int root_bits, const uint16_t *sym, const uint16_t *count) {
	if (s.cnt < 15) inf_refill(s, L, lane);
	const uint32_t e = inf_uni(root[uint32_t(s.bits) & ((1u << root_bits) - 1u)]);
	if (e) { s.bits >>= (e & 15u); s.cnt -= int(e & 15u); return e >> 4; }
	uint32_t code = 0, first = 0, index = 0;
	uint64_t b = s.bits;
	for (int len = 1; len <= 15; ++len) {
		code |= uint32_t(b & 1u); b >>= 1;
		const uint32_t c = inf_uni(count[len]);
		if (code < first + c) { s.bits >>= len; s.cnt -= len; return inf_uni(sym[index + (code - first)]); }
		index += c; first += c; first <<= 1; code <<= 1;
	}
	return 0xFFFFu;
}

// ---- CRC-32 of the inflated block (the gzip trailer's, RFC 1952 8.) ----------------------------------------------------------------
// Every lane takes 1 KB of the block, four bytes per step through four 256-entry tables (LDS, one set per workgroup); the 64 partial values are joined
// in GF(2): crc(A || B) = crc(A) * x^(8 |B|) mod P  xor  crc(B), with x^(2^k) mod P squared up in LDS once per wave (the scheme of zlib's
// crc32_combine).
constexpr uint32_t INF_CRC_POLY = 0xEDB88320u;   // reflected
// x^(2^k) mod P, k = 0 .. 31 (x^1 squared up; checked against zlib.crc32 of concatenations when the constants were made)
__constant__ const uint32_t INF_X2N[32] = {0x40000000u, 0x20000000u, 0x08000000u, 0x00800000u, 0x00008000u, 0xedb88320u, 0xb1e6b092u, 0xa06a2517u,
                                           0xed627daeu, 0x88d14467u, 0xd7bbfe6au, 0xec447f11u, 0x8e7ea170u, 0x6427800eu, 0x4d47bae0u, 0x09fe548fu,
                                           0x83852d0fu, 0x30362f1au, 0x7b5a9cc3u, 0x31fec169u, 0x9fec022au, 0x6c8dedc4u, 0x15d6874du, 0x5fde7a4eu,
                                           0xbad90e37u, 0x2e4e5eefu, 0x4eaba214u, 0xa8a472c0u, 0x429a969eu, 0x148d302au, 0xc40ba6d0u, 0xc4e22c3cu};
__device__ inline uint32_t inf_multmodp(uint32_t a, uint32_t b) {   // a(x) * b(x) mod P, reflected representation (bit 31 = x^0)
	uint32_t m = 1u << 31, p = 0;
	for (;;) {
		if (a & m) { p ^= b; if ((a & (m - 1u)) == 0u) break; }
		m >>= 1;
		b = (b & 1u) ? (b >> 1) ^ INF_CRC_POLY : b >> 1;
	}
	return p;
}
__device__ inline uint32_t inf_x8n(const uint32_t *x2n, uint32_t n_bytes) {   // x^(8 n) mod P
	uint32_t p = 1u << 31, k = 3;
	for (uint32_t n = n_bytes; n; n >>= 1, ++k) if (n & 1u) p = inf_multmodp(x2n[k & 31u], p);
	return p;
}
// crc_tab: [4][256] of the workgroup; x2n: [32] of the wave.  All 64 lanes; returns the CRC-32 of out[0 .. len) in every lane.
// A lane takes ONE 128-byte line per round (a kilobyte per lane re-fetched every line 32 times: + 19 % on the kernel), the 64 values of a
// round of 8 KB are joined pairwise over six levels of shuffles -- the operator of a piece of 128 * 2^d bytes IS x2n[10 + d] -- and the rounds
// one after the other.
__device__ inline uint32_t inf_crc32_block(const uint8_t *out, uint32_t len, const uint32_t *crc_tab, uint32_t *x2n, uint32_t lane) {
	if (lane < 32u) x2n[lane] = INF_X2N[lane];
	uint32_t total = 0;
	for (uint32_t base = 0; base < len; base += 8192u) {
		const uint32_t begin = base + lane * 128u;
		const uint32_t mine = begin < len ? (len - begin < 128u ? len - begin : 128u) : 0u;
		uint32_t c = 0xFFFFFFFFu;
		const uint8_t *p = out + begin;
		uint32_t i = 0;
		for (; i < mine && (uintptr_t(p + i) & 3u); ++i) c = crc_tab[(c ^ p[i]) & 0xFFu] ^ (c >> 8);      // to a word boundary
		auto step4 = [&](uint32_t w) {                                                                   // four bytes per step (slicing by 4)
			c ^= w;
			c = crc_tab[768u + (c & 0xFFu)] ^ crc_tab[512u + ((c >> 8) & 0xFFu)] ^ crc_tab[256u + ((c >> 16) & 0xFFu)] ^ crc_tab[c >> 24];
		};
		for (; i + 32u <= mine; i += 32u) {              // eight words in flight before the chain through the tables waits for any of them
			const uint32_t *q = reinterpret_cast<const uint32_t *>(p + i);
			const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4], w5 = q[5], w6 = q[6], w7 = q[7];
			step4(w0); step4(w1); step4(w2); step4(w3); step4(w4); step4(w5); step4(w6); step4(w7);
		}
		for (; i + 4u <= mine; i += 4u) step4(*reinterpret_cast<const uint32_t *>(p + i));
		for (; i < mine; ++i) c = crc_tab[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
		c = ~c;                                          // (an empty piece: 0)
		uint32_t clen = mine;
#pragma unroll
		for (int d = 0; d < 6; ++d) {
			const uint32_t oc = uint32_t(__shfl_xor(int(c), 1 << d)), ol = uint32_t(__shfl_xor(int(clen), 1 << d));
			if (!(lane & ((2u << d) - 1u))) {            // the left piece (lanes 0, 2^(d+1), ...) takes the right one in
				const uint32_t op = ol == (128u << d) ? x2n[10 + d] : inf_x8n(x2n, ol);
				c = inf_multmodp(op, c) ^ oc;
				clen += ol;
			}
		}
		const uint32_t rc = uint32_t(__shfl(int(c), 0)), rl = uint32_t(__shfl(int(clen), 0));
		total = base ? (inf_multmodp(rl == 8192u ? x2n[16] : inf_x8n(x2n, rl), total) ^ rc) : rc;
	}
	return total;
}

// The symbols of one DEFLATE block, as a function of its own (a real call): what it keeps -- bit buffer, counts, positions -- is named uniform on
// entry and nothing lane-dependent decides a branch inside, so the state lives in scalar registers and the scalar unit does the bit-buffer
// arithmetic (the look-up's address and the literal's select are what is left for the vector unit).  Inlined into the kernel the same code keeps
// this state in vector registers (the compiler calls the table build's result divergent, and the block loop that hangs on it with it), and named
// uniform there it spills: the kernel as a whole wants more than a wave's 100 scalar registers (NOTES_r05 §13).
struct InfLoopIO {
	uint64_t ipos, loaded_hi, bits;
	int cnt;
	uint32_t pos, nlit, mylit, err;
};
using InfLdsPtr = __attribute__((address_space(3))) InfWaveLds *;
using InfLdsU32 = __attribute__((address_space(3))) uint32_t *;
__device__ __attribute__((noinline)) InfLoopIO inf_symbol_loop(InfLoopIO io, const uint64_t *gin, uint64_t in_words, uint32_t lds_wave, uint32_t lds_len_tab, uint32_t lds_dist_tab,
                                                               uint8_t *out_, uint32_t out_cap_) {
	const uint32_t lane = threadIdx.x & 63u;
	InfWaveLds &L = *(InfWaveLds *)(InfLdsPtr)(uintptr_t)inf_uni(lds_wave);
	const uint32_t *len_tab = (const uint32_t *)(InfLdsU32)(uintptr_t)inf_uni(lds_len_tab), *dist_tab = (const uint32_t *)(InfLdsU32)(uintptr_t)inf_uni(lds_dist_tab);
	using GlobalU8 = __attribute__((address_space(1))) uint8_t *;      // (device memory said aloud: global_, not flat_ loads and stores)
	const GlobalU8 out = (GlobalU8)(uintptr_t)inf_uni64(reinterpret_cast<uint64_t>(out_));
	const uint32_t out_cap = inf_uni(out_cap_);
	InfState s;
	s.gin = reinterpret_cast<const uint64_t *>(inf_uni64(reinterpret_cast<uint64_t>(gin))); s.in_words = inf_uni64(in_words);
	s.ipos = inf_uni64(io.ipos); s.loaded_hi = inf_uni64(io.loaded_hi); s.bits = inf_uni64(io.bits); s.cnt = int(inf_uni(uint32_t(io.cnt)));
	uint32_t pos = inf_uni(io.pos), nlit = inf_uni(io.nlit), err = INF_OK;
	uint32_t mylit = io.mylit;      // literal number `lane` of those waiting for their store (a register per lane: no LDS round for a literal)
	auto flush = [&]() {
		if (nlit) { if (lane < nlit) out[pos + lane] = uint8_t(mylit); pos += nlit; nlit = 0; }
	};
	for (;;) {
		const uint32_t sy = inf_uni(inf_decode(s, L, lane, L.lroot, INF_LROOT, L.lsym, L.lcount));
		if (sy < 256u) {
			mylit = lane == nlit ? sy : mylit;
			if (++nlit == 64u) { if (pos + 64u > out_cap) { err = INF_OUTPUT_OVERRUN; break; } flush(); }
			continue;
		}
		if (sy == 256u) break;
		if (sy > 285u) { err = INF_BAD_CODE; break; }
		if (pos + nlit > out_cap) { err = INF_OUTPUT_OVERRUN; break; }
		flush();
		const uint32_t li = sy - 257u;
		const uint32_t lt = inf_uni(len_tab[li]);
		uint32_t len = lt & 0xFFFFu;
		const int le = int(lt >> 16);
		if (le) len += inf_take(s, L, lane, le);
		const uint32_t ds = inf_uni(inf_decode(s, L, lane, L.droot, INF_DROOT, L.dsym, L.dcount));
		if (ds > 29u) { err = INF_BAD_CODE; break; }
		const uint32_t dt = inf_uni(dist_tab[ds]);
		uint32_t dist = dt & 0xFFFFu;
		const int de = int(dt >> 16);
		if (de) dist += inf_take(s, L, lane, de);
		if (dist > pos) { err = INF_BAD_DISTANCE; break; }
		if (pos + len > out_cap) { err = INF_OUTPUT_OVERRUN; break; }
		// the source window [pos - dist, pos) is complete: a copy longer than dist repeats it
#ifndef INF_NO_FENCE
		__threadfence_block();   // this wave's earlier stores, before other lanes read them
#endif
		const GlobalU8 src = out + (pos - dist);
		if (dist >= len) { for (uint32_t i = lane; i < len; i += 64) out[pos + i] = src[i]; }
		else { for (uint32_t i = lane; i < len; i += 64) out[pos + i] = src[i % dist]; }
		pos += len;
	}
	InfLoopIO r;
	r.ipos = s.ipos; r.loaded_hi = s.loaded_hi; r.bits = s.bits; r.cnt = s.cnt; r.pos = pos; r.nlit = nlit; r.mylit = mylit; r.err = err;
	return r;
}

// One BGZF block per wave.  in_off / in_len: the DEFLATE payload inside d_in (behind the 18-byte header); out_off / out_len: where its
// ISIZE bytes go in d_out.  d_in must be 8-byte aligned and in_total_len is the number of bytes that may be read.
// Waves per SIMD the register allocation aims at (scripts/experiments/inflate_variants/run.sh; 3.3 GB synthetic 10x BAM / the same with random
// bases and binned qualities): with the symbol loop inlined 4 / 5 / 6 waves gave 84.5 / 99.9 / 112.2 and 31.6 / 37.6 / 40.0 GB/s, and 8 real waves
// 85.3 / 30.0 (the loop's state spilled); with the loop as a function of its own 6 waves 103.2 / 37.8 and 8 waves 131.3 / 44.7 (kept).  The fence
// before a match copy costs nothing at any of them.
#ifndef INF_WAVES_PER_EU
#define INF_WAVES_PER_EU 8
#endif
__global__ __launch_bounds__(INF_WAVES * 64) __attribute__((amdgpu_waves_per_eu(INF_WAVES_PER_EU, INF_WAVES_PER_EU))) void bgzf_inflate_kernel(const uint8_t *__restrict__ d_in, uint64_t in_total_len,
                                                                       const uint64_t *__restrict__ in_off, const uint32_t *__restrict__ in_len,
                                                                       const uint64_t *__restrict__ out_off, const uint32_t *__restrict__ out_len,
                                                                       uint32_t n_blocks, uint8_t *d_out, uint32_t *__restrict__ status,
                                                                       const uint32_t *__restrict__ crc32 /* the blocks' stored CRC-32, or null: not checked */) {
	__shared__ InfWaveLds lds[INF_WAVES];
	__shared__ uint32_t crc_tab[4 * 256];               // slicing by 4: table k advances a byte that has k more bytes behind it in the word
	__shared__ uint32_t crc_x2n[INF_WAVES][32];
	if (crc32) {
		uint32_t c = threadIdx.x & 255u;                // the first 256 threads: one entry of every table each
		for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ INF_CRC_POLY : c >> 1;
		if (threadIdx.x < 256) crc_tab[threadIdx.x] = c;
		__syncthreads();
		for (int t = 1; t < 4; ++t) { c = (c >> 8) ^ crc_tab[c & 0xFFu]; if (threadIdx.x < 256) crc_tab[t * 256 + threadIdx.x] = c; __syncthreads(); }
	}
	// base | extra bits << 16 of the length / distance symbols, in LDS: as __constant__ arrays indexed by a decoded symbol they were two
	// dependent loads through the vector memory path per match (94.1 -> 110.6 GB/s on the 10x BAM with them here)
	__shared__ uint32_t len_tab[32], dist_tab[32];
	if (threadIdx.x < 32) len_tab[threadIdx.x] = uint32_t(INF_LEN_BASE[threadIdx.x]) | uint32_t(INF_LEN_EXTRA[threadIdx.x]) << 16;
	else if (threadIdx.x < 64) dist_tab[threadIdx.x - 32] = uint32_t(INF_DIST_BASE[threadIdx.x - 32]) | uint32_t(INF_DIST_EXTRA[threadIdx.x - 32]) << 16;
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t blk = blockIdx.x * INF_WAVES + wave;
	if (blk >= n_blocks) return;
	InfWaveLds &L = lds[wave];
	InfState s;
	s.gin = reinterpret_cast<const uint64_t *>(d_in);
	s.in_words = (in_total_len + 7) >> 3;
	const uint64_t in_begin = in_off[blk], in_end = in_begin + in_len[blk];
	s.ipos = in_begin; s.loaded_hi = in_begin & ~uint64_t(255); s.bits = 0; s.cnt = 0;
	uint8_t *const out = d_out + out_off[blk];
	const uint32_t out_cap = out_len[blk];
	uint32_t pos = 0, nlit = 0, err = INF_OK;
	uint32_t mylit = 0;      // literal number `lane` of those waiting for their store (a register per lane: no LDS round for a literal)

	auto flush = [&]() {
		if (nlit) { if (lane < nlit) out[pos + lane] = uint8_t(mylit); pos += nlit; nlit = 0; }
	};

	for (bool last = false; !last && !err;) {
		if (s.ipos - uint64_t(s.cnt >> 3) > in_end + 8) { err = INF_INPUT_OVERRUN; break; }   // (a damaged stream of empty blocks must not read on through its neighbours)
		last = inf_take(s, L, lane, 1) != 0;
		const uint32_t type = inf_take(s, L, lane, 2);
		if (type == 0) {   // stored: to the byte boundary, LEN, NLEN, bytes
			// (the literals an earlier DEFLATE block left waiting go out first -- inside this block's own output range only: a stream
			// longer than its ISIZE must not write into its neighbour's bytes)
			if (pos + nlit > out_cap) { err = INF_OUTPUT_OVERRUN; break; }
			flush();
			const int drop = s.cnt & 7;
			s.bits >>= drop; s.cnt -= drop;
			const uint32_t len = inf_take(s, L, lane, 16), nlen = inf_take(s, L, lane, 16);
			if ((len ^ nlen) != 0xFFFFu) { err = INF_BAD_STORED; break; }
			const uint64_t from = s.ipos - uint64_t(s.cnt >> 3);          // first byte not consumed yet
			if (from + len > in_end) { err = INF_INPUT_OVERRUN; break; }
			if (pos + len > out_cap) { err = INF_OUTPUT_OVERRUN; break; }
			for (uint32_t i = lane; i < len; i += 64) out[pos + i] = d_in[from + i];
			pos += len;
			s.ipos = from + len; s.bits = 0; s.cnt = 0; s.loaded_hi = s.ipos & ~uint64_t(255);
			continue;
		}
		if (type == 3) { err = INF_BAD_BLOCK_TYPE; break; }
		int hlit, hdist;
		if (type == 1) {   // fixed code (RFC 1951 3.2.6)
			hlit = 288; hdist = 30;
			for (uint32_t i = lane; i < 288u; i += 64) L.lens[i] = i < 144u ? 8 : (i < 256u ? 9 : (i < 280u ? 7 : 8));
			if (lane < 30u) L.lens[288u + lane] = 5;
		} else {           // dynamic code (3.2.7)
			hlit = int(inf_take(s, L, lane, 5)) + 257; hdist = int(inf_take(s, L, lane, 5)) + 1;
			const int hclen = int(inf_take(s, L, lane, 4)) + 4;
			if (hlit > 286 || hdist > 30) { err = INF_BAD_LENGTHS; break; }
			if (lane < 19u) L.lens[lane] = 0;
			for (int i = 0; i < hclen; ++i) {
				const uint32_t v = inf_take(s, L, lane, 3);
				if (lane == 0) L.lens[INF_CL_ORDER[i]] = uint8_t(v);
			}
			// the code-length code borrows the distance tables (7-bit root)
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
				// (the lengths land behind the 19 code-length lengths: lens[] is read again by inf_build below, from offset 0 -- so they are
				// written to their final places only after the code-length table is no longer needed: it lives in droot / dsym, not in lens)
				for (uint32_t i = lane; i < rep; i += 64) L.lens[uint32_t(have) + i] = uint8_t(val);
				have += int(rep);
			}
			if (err) break;
			if (inf_uni(L.lens[256]) == 0) { err = INF_BAD_LENGTHS; break; }   // no end-of-block code
		}
		if (!inf_build(L.lens, hlit, INF_LROOT, L.lroot, L.lsym, L.lcount, lane)) { err = INF_OVERSUBSCRIBED; break; }
		if (!inf_build(L.lens + hlit, hdist, INF_DROOT, L.droot, L.dsym, L.dcount, lane)) { err = INF_OVERSUBSCRIBED; break; }

		{
			InfLoopIO io;
			io.ipos = s.ipos; io.loaded_hi = s.loaded_hi; io.bits = s.bits; io.cnt = s.cnt; io.pos = pos; io.nlit = nlit; io.mylit = mylit; io.err = 0;
			io = inf_symbol_loop(io, s.gin, s.in_words, uint32_t(uintptr_t((InfLdsPtr)&L)), uint32_t(uintptr_t((InfLdsU32)len_tab)), uint32_t(uintptr_t((InfLdsU32)dist_tab)), out, out_cap);
			s.ipos = io.ipos; s.loaded_hi = io.loaded_hi; s.bits = io.bits; s.cnt = io.cnt; pos = io.pos; nlit = io.nlit; mylit = io.mylit; err = io.err;
		}
	}
	if (!err) {
		if (pos + nlit > out_cap) err = INF_OUTPUT_OVERRUN;
		else {
			flush();
			if (pos != out_cap) err = INF_SIZE_MISMATCH;
			else if (s.ipos - uint64_t(s.cnt >> 3) > in_end) err = INF_INPUT_OVERRUN;
		}
	}
	if (!err && crc32) {
		__threadfence_block();   // the block's last stores, before every lane reads its kilobyte back
		if (inf_crc32_block(out, out_cap, crc_tab, crc_x2n[wave], lane) != crc32[blk]) err = INF_CRC_MISMATCH;
	}
	if (lane == 0) status[blk] = err;
}

}  // namespace dropest
