// fast_inflate.h -- raw DEFLATE (RFC 1951) decoder and CRC-32 for BGZF blocks (one whole block in, one whole block out: no streaming
// state, no window).  zlib 1.2.11's inflate + crc32 were what the BAM reader waited for (profiles/NOTES_r03.md): this decoder keeps 64 bits
// of input in a register, looks literal / length codes up in an 11-bit table (longer codes through sub-tables), copies matches in 8-byte
// words, and the CRC walks 8 bytes per step (slicing-by-8).  Written from the RFC; the reader (bam_ingest.cpp) still checks every block's
// CRC-32 and ISIZE and hands a block this decoder refuses to zlib.  tests/test_fast_inflate.py compares it with zlib over streams of
// every block type and level.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace fastinflate {

constexpr int LIT_BITS = 11, OFF_BITS = 8;
constexpr uint32_t F_LITERAL = 0x8000u, F_EOB = 0x4000u, F_SUB = 0x2000u, F_BAD = 0x1000u;
// table entry: bits 0..5 code length to consume (sub-table pointer: LIT_BITS / OFF_BITS), bits 6..10 number of extra bits (pointer: index
// bits of the sub-table), flags, bits 16..31 literal / base value / first index of the sub-table

struct Tables {
	uint32_t lit[(1 << LIT_BITS) + 288 * 16];   // every sub-table has 2^(longest code - root bits) entries; at most one per symbol
	uint32_t off[(1 << OFF_BITS) + 30 * 128];
};

// Canonical Huffman code (lens[i] = code length of symbol i, 0 = unused) -> look-up table indexed by the next `tbits` input bits
// (LSB first: codes are stored bit-reversed).  payload(sym) gives the entry without its length field.  Returns false on an
// over-subscribed code or a table overflow; an incomplete code leaves F_BAD entries (a stream that reaches one is refused).
template <class Payload>
inline bool build_table(const uint8_t *lens, int n, int tbits, uint32_t *table, size_t cap, Payload payload) {
	int count[16] = {0};
	for (int i = 0; i < n; ++i) ++count[lens[i]];
	count[0] = 0;
	int left = 1;
	for (int l = 1; l <= 15; ++l) { left = (left << 1) - count[l]; if (left < 0) return false; }
	uint32_t next[16];
	uint32_t code = 0;
	for (int l = 1; l <= 15; ++l) { code = (code + uint32_t(count[l - 1])) << 1; next[l] = code; }
	const uint32_t tsize = 1u << tbits;
	for (uint32_t i = 0; i < tsize; ++i) table[i] = F_BAD | 1u;
	int maxlen = 0;
	for (int l = 15; l >= 1; --l) if (count[l]) { maxlen = l; break; }
	auto reverse = [](uint32_t c, int l) { uint32_t r = 0; for (int i = 0; i < l; ++i) { r = (r << 1) | (c & 1u); c >>= 1; } return r; };
	// short codes: replicated over the table
	size_t used = tsize;
	const int sub_bits = maxlen > tbits ? maxlen - tbits : 0;
	for (int i = 0; i < n; ++i) {
		const int l = lens[i];
		if (!l) continue;
		const uint32_t c = reverse(next[l]++, l);
		if (l <= tbits) {
			const uint32_t e = payload(i) | uint32_t(l);
			for (uint32_t j = c; j < tsize; j += 1u << l) table[j] = e;
		} else {
			const uint32_t low = c & (tsize - 1u);
			if (!(table[low] & F_SUB)) {   // open the sub-table of this prefix: 2^sub_bits entries
				if (used + (size_t(1) << sub_bits) > cap) return false;
				table[low] = (uint32_t(used) << 16) | F_SUB | (uint32_t(sub_bits) << 6) | uint32_t(tbits);
				for (uint32_t j = 0; j < (1u << sub_bits); ++j) table[used + j] = F_BAD | 1u;
				used += size_t(1) << sub_bits;
			}
			const uint32_t start = table[low] >> 16, e = payload(i) | uint32_t(l - tbits);
			for (uint32_t j = c >> tbits; j < (1u << sub_bits); j += 1u << (l - tbits)) table[start + j] = e;
		}
	}
	return true;
}

inline uint32_t litlen_payload(int sym) {
	static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
	static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
	if (sym < 256) return (uint32_t(sym) << 16) | F_LITERAL;
	if (sym == 256) return F_EOB;
	if (sym > 285) return F_BAD;
	return (uint32_t(base[sym - 257]) << 16) | (uint32_t(extra[sym - 257]) << 6);
}
inline uint32_t offset_payload(int sym) {
	static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
	static const uint8_t extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
	if (sym > 29) return F_BAD;
	return (uint32_t(base[sym]) << 16) | (uint32_t(extra[sym]) << 6);
}

struct BitReader {
	const uint8_t *p, *end;
	uint64_t bits = 0;
	int cnt = 0;          // valid bits in `bits` (bits above them may hold a preview of the next byte: harmless, it is OR-ed in again)
	int virt = 0;         // zero bytes invented past the end of the input (a stream that CONSUMES any of them is truncated)
	inline void refill() {   // at least 56 bits afterwards
		if (end - p >= 8) {
			uint64_t w;
			std::memcpy(&w, p, 8);
			bits |= w << cnt;
			p += (63 - cnt) >> 3;
			cnt |= 56;
		} else {
			while (cnt <= 56) { if (p < end) bits |= uint64_t(*p++) << cnt; else ++virt; cnt += 8; }
		}
	}
	inline uint32_t peek(int n) const { return uint32_t(bits & ((1ull << n) - 1ull)); }
	inline void drop(int n) { bits >>= n; cnt -= n; }
	inline uint32_t take(int n) { if (cnt < n) refill(); const uint32_t v = peek(n); drop(n); return v; }
	inline bool overrun() const { return cnt < virt * 8; }   // more bits were consumed than the input holds
};

// Decodes in[0 .. in_len) into out[0 .. out_len) exactly.  false: malformed stream, or the output does not come out at out_len bytes.
inline bool inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
	static thread_local Tables T;
	static thread_local bool fixed_ready = false;
	static thread_local Tables F;
	BitReader br{in, in + in_len};
	uint8_t *o = out, *const oend = out + out_len;
	for (;;) {
		const uint32_t last = br.take(1), type = br.take(2);
		if (br.overrun()) return false;
		if (type == 0) {   // stored: skip to the byte boundary, LEN, NLEN, bytes
			br.drop(br.cnt & 7);
			const uint32_t len = br.take(16), nlen = br.take(16);
			if (br.overrun() || (len ^ nlen) != 0xFFFFu) return false;
			const uint8_t *src = br.p - ((br.cnt - br.virt * 8) >> 3);   // the real bytes still in the register were read ahead (whole bytes here)
			if (size_t(br.end - src) < len || size_t(oend - o) < len) return false;
			std::memcpy(o, src, len);
			o += len;
			br = BitReader{src + len, in + in_len};
		} else if (type == 1 || type == 2) {
			const Tables *tab;
			if (type == 1) {
				if (!fixed_ready) {
					uint8_t l[288 + 32];
					for (int i = 0; i < 144; ++i) l[i] = 8;
					for (int i = 144; i < 256; ++i) l[i] = 9;
					for (int i = 256; i < 280; ++i) l[i] = 7;
					for (int i = 280; i < 288; ++i) l[i] = 8;
					for (int i = 0; i < 32; ++i) l[288 + i] = 5;
					if (!build_table(l, 288, LIT_BITS, F.lit, sizeof(F.lit) / 4, litlen_payload)) return false;
					if (!build_table(l + 288, 32, OFF_BITS, F.off, sizeof(F.off) / 4, offset_payload)) return false;
					fixed_ready = true;
				}
				tab = &F;
			} else {
				const uint32_t hlit = br.take(5) + 257, hdist = br.take(5) + 1, hclen = br.take(4) + 4;
				if (hlit > 286 || hdist > 30) return false;
				static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
				uint8_t cl[19] = {0};
				for (uint32_t i = 0; i < hclen; ++i) cl[order[i]] = uint8_t(br.take(3));
				uint32_t ct[128 + 8];
				if (!build_table(cl, 19, 7, ct, sizeof(ct) / 4, [](int s) { return uint32_t(s) << 16; })) return false;
				uint8_t lens[286 + 30 + 138];
				uint32_t n = 0;
				while (n < hlit + hdist) {
					if (br.cnt < 7 + 7) br.refill();
					const uint32_t e = ct[br.peek(7)];
					if (e & F_BAD) return false;
					br.drop(int(e & 63u));
					const uint32_t s = e >> 16;
					if (s < 16) lens[n++] = uint8_t(s);
					else {
						uint32_t rep, v = 0;
						if (s == 16) { if (!n) return false; v = lens[n - 1]; rep = 3 + br.take(2); }
						else if (s == 17) rep = 3 + br.take(3);
						else rep = 11 + br.take(7);
						if (n + rep > hlit + hdist) return false;
						while (rep--) lens[n++] = uint8_t(v);
					}
				}
				if (br.overrun() || lens[256] == 0) return false;
				if (!build_table(lens, int(hlit), LIT_BITS, T.lit, sizeof(T.lit) / 4, litlen_payload)) return false;
				if (!build_table(lens + hlit, int(hdist), OFF_BITS, T.off, sizeof(T.off) / 4, offset_payload)) return false;
				tab = &T;
			}
			const uint32_t *lt = tab->lit, *ot = tab->off;
			constexpr uint32_t LMASK = (1u << LIT_BITS) - 1u, OMASK = (1u << OFF_BITS) - 1u;
			// One symbol: the table entry for the bits at hand, sub-table followed, its code bits dropped.
#define FI_LITLEN(e) do { e = lt[br.bits & LMASK]; if (e & F_SUB) { br.drop(LIT_BITS); e = lt[(e >> 16) + br.peek(int((e >> 6) & 31u))]; } br.drop(int(e & 63u)); } while (0)
			bool eob = false;
			// fast loop: room for the longest match plus the slack of the word copies, and whole words of input left
			while (oend - o >= 258 + 16 && br.end - br.p >= 8) {
				br.refill();                                            // >= 56 bits
				uint32_t e;
				FI_LITLEN(e);                                           // <= 15 bits
				if (e & F_LITERAL) {
					*o++ = uint8_t(e >> 16);
					FI_LITLEN(e);                                       // <= 30
					if (e & F_LITERAL) {
						*o++ = uint8_t(e >> 16);
						FI_LITLEN(e);                                   // <= 45
						if (e & F_LITERAL) { *o++ = uint8_t(e >> 16); continue; }
					}
					br.refill();                                        // (a length or the end of the block follows: it wants up to 48 bits)
				}
				if (e & (F_EOB | F_BAD)) { if (e & F_BAD) return false; eob = true; break; }
				const int leb = int((e >> 6) & 31u);
				const uint32_t len = (e >> 16) + br.peek(leb);
				br.drop(leb);                                           // <= 20 since the last refill
				uint32_t d = ot[br.bits & OMASK];
				if (d & F_SUB) { br.drop(OFF_BITS); d = ot[(d >> 16) + br.peek(int((d >> 6) & 31u))]; }
				if (d & F_BAD) return false;
				br.drop(int(d & 63u));                                  // <= 35
				const int deb = int((d >> 6) & 31u);
				const uint32_t dist = (d >> 16) + br.peek(deb);
				br.drop(deb);                                           // <= 48
				if (dist > size_t(o - out)) return false;
				const uint8_t *s = o - dist;
				uint8_t *t = o;
				o += len;
				if (dist >= 16) {   // sixteen bytes a step (BAM records repeat each other at distances of a record or more: matches are long)
					struct W16 { uint64_t a, b; } w;
					do { std::memcpy(&w, s, 16); std::memcpy(t, &w, 16); s += 16; t += 16; } while (t < o);
				} else if (dist >= 8) {
					uint64_t w;
					std::memcpy(&w, s, 8); std::memcpy(t, &w, 8);
					std::memcpy(&w, s + 8, 8); std::memcpy(t + 8, &w, 8);
					if (len > 16) { s += 16; t += 16; do { std::memcpy(&w, s, 8); std::memcpy(t, &w, 8); s += 8; t += 8; } while (t < o); }
				} else if (dist == 1) std::memset(t, *s, len);
				else { do { *t++ = *s++; } while (t < o); }
			}
			// careful loop: the tail of the output and of the input
			while (!eob) {
				if (br.cnt < 48) br.refill();
				uint32_t e;
				FI_LITLEN(e);
				if (e & F_BAD) return false;
				if (e & F_LITERAL) {
					if (o >= oend) return false;
					*o++ = uint8_t(e >> 16);
					continue;
				}
				if (e & F_EOB) break;
				const int leb = int((e >> 6) & 31u);
				const uint32_t len = (e >> 16) + br.peek(leb);
				br.drop(leb);
				uint32_t d = ot[br.bits & OMASK];
				if (d & F_SUB) { br.drop(OFF_BITS); d = ot[(d >> 16) + br.peek(int((d >> 6) & 31u))]; }
				if (d & F_BAD) return false;
				br.drop(int(d & 63u));
				if (br.cnt < 13) br.refill();
				const int deb = int((d >> 6) & 31u);
				const uint32_t dist = (d >> 16) + br.peek(deb);
				br.drop(deb);
				if (dist > size_t(o - out) || len > size_t(oend - o)) return false;
				const uint8_t *s = o - dist;
				for (uint32_t i = 0; i < len; ++i) o[i] = s[i];
				o += len;
				if (br.overrun()) return false;
			}
#undef FI_LITLEN
			if (br.overrun()) return false;
		} else return false;
		if (last) break;
	}
	return o == oend && !br.overrun();
}

// CRC-32 (IEEE 802.3, the one of gzip), slicing-by-8
struct CrcTables {
	uint32_t t[8][256];
	CrcTables() {
		for (uint32_t i = 0; i < 256; ++i) {
			uint32_t c = i;
			for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
			t[0][i] = c;
		}
		for (uint32_t i = 0; i < 256; ++i)
			for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xFFu];
	}
};
#if defined(__x86_64__)
// Carry-less-multiply folding (the algorithm of Intel's "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ", constants of the
// gzip polynomial): 64 bytes per step in four 128-bit lanes, folded to 128, 64 and 32 bits.  n >= 64, a multiple of 16; c is the running
// (pre-inversion) state.  About 10x the table walk below; the caller checks the CPU for PCLMULQDQ once.
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_clmul(const uint8_t *buf, size_t len, uint32_t c) {
	alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
	alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
	alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
	alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
	auto load = [](const void *p) { return _mm_loadu_si128(static_cast<const __m128i *>(p)); };
	__m128i x0, x1, x2, x3, x4, x5, x6, x7, x8;
	x1 = load(buf); x2 = load(buf + 16); x3 = load(buf + 32); x4 = load(buf + 48);
	x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128(int(c)));
	x0 = _mm_load_si128(reinterpret_cast<const __m128i *>(k1k2));
	buf += 64; len -= 64;
	while (len >= 64) {
		x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
		x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
		x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
		x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
		x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), load(buf)); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), load(buf + 16));
		x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), load(buf + 32)); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), load(buf + 48));
		buf += 64; len -= 64;
	}
	x0 = _mm_load_si128(reinterpret_cast<const __m128i *>(k3k4));
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
	while (len >= 16) {
		x2 = load(buf);
		x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
		buf += 16; len -= 16;
	}
	x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
	x3 = _mm_setr_epi32(~0, 0, ~0, 0);
	x1 = _mm_srli_si128(x1, 8);
	x1 = _mm_xor_si128(x1, x2);
	x0 = _mm_loadl_epi64(reinterpret_cast<const __m128i *>(k5k0));
	x2 = _mm_srli_si128(x1, 4);
	x1 = _mm_and_si128(x1, x3);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	x0 = _mm_load_si128(reinterpret_cast<const __m128i *>(poly));
	x2 = _mm_and_si128(x1, x3);
	x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
	x2 = _mm_and_si128(x2, x3);
	x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	return uint32_t(_mm_extract_epi32(x1, 1));
}
#endif

inline uint32_t crc32(const uint8_t *p, size_t n) {
	static const CrcTables C;
	uint32_t c = 0xFFFFFFFFu;
#if defined(__x86_64__)
	static const bool clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
	if (clmul && n >= 64) { const size_t chunk = n & ~size_t(15); c = crc32_clmul(p, chunk, c); p += chunk; n -= chunk; }
#endif
	while (n >= 8) {
		uint64_t w;
		std::memcpy(&w, p, 8);
		w ^= c;
		c = C.t[7][w & 0xFF] ^ C.t[6][(w >> 8) & 0xFF] ^ C.t[5][(w >> 16) & 0xFF] ^ C.t[4][(w >> 24) & 0xFF] ^
		    C.t[3][(w >> 32) & 0xFF] ^ C.t[2][(w >> 40) & 0xFF] ^ C.t[1][(w >> 48) & 0xFF] ^ C.t[0][w >> 56];
		p += 8; n -= 8;
	}
	while (n--) c = (c >> 8) ^ C.t[0][(c ^ *p++) & 0xFFu];
	return ~c;
}

}  // namespace fastinflate
