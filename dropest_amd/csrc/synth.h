// synth.h -- the per-read function of the synthetic stream (include/dropest_synth.h), host + device.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dropest_synth.h"
#include "util.h"

namespace dropest {

struct Philox4 { uint32_t v[4]; };

__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) { return uint32_t((uint64_t(a) * b) >> 32); }

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
	const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
	for (int r = 0; r < 10; ++r) {
		const uint32_t hi0 = mulhi32(M0, c0), lo0 = M0 * c0;
		const uint32_t hi1 = mulhi32(M1, c2), lo1 = M1 * c2;
		const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
		c0 = n0; c1 = n1; c2 = n2; c3 = n3;
		k0 += W0; k1 += W1;
	}
	return Philox4{{c0, c1, c2, c3}};
}

__host__ __device__ inline uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __umul64hi(a, b);
#else
	return uint64_t((static_cast<unsigned __int128>(a) * b) >> 64);
#endif
}

// first index i with cdf[i] >= x (cdf ascending, last == 0xFFFFFFFF)
__host__ __device__ inline uint32_t cdf_pick(const uint32_t *cdf, uint32_t n, uint32_t x) {
	uint32_t lo = 0, hi = n - 1;
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (cdf[mid] >= x) hi = mid; else lo = mid + 1;
	}
	return lo;
}
__host__ __device__ inline uint32_t cdf_width(const uint32_t *cdf, uint32_t i) {   // probability * 2^32 (approx.)
	return i ? cdf[i] - cdf[i - 1] : cdf[0] + 1u;
}

struct SynthRead { uint64_t cb, umi; uint32_t gene, aux; };

__host__ __device__ inline SynthRead synth_read(const dropest_synth_params &p, const uint64_t *cell_cb,
                                                const uint32_t *cell_cdf, const uint32_t *gene_cdf, uint64_t ordinal) {
	const uint32_t k0 = uint32_t(p.seed) ^ (p.stream_id * 0x9E3779B1u), k1 = uint32_t(p.seed >> 32) + p.stream_id;
	const Philox4 a = philox4x32_10(uint32_t(ordinal), uint32_t(ordinal >> 32), 0u, 0u, k0, k1);
	const Philox4 b = philox4x32_10(uint32_t(ordinal), uint32_t(ordinal >> 32), 1u, 0u, k0, k1);
	SynthRead r;
	const uint32_t cat = mulhi32(a.v[0], 1000u);                      // 0..999
	const uint32_t cell = cdf_pick(cell_cdf, p.n_cells, a.v[1]);
	const uint32_t g = cdf_pick(gene_cdf, p.n_genes, a.v[2]);
	const uint64_t cb_mask = (1ull << (2 * p.cb_len)) - 1ull, cb_sent = 1ull << (2 * p.cb_len);
	const uint64_t umi_mask = (1ull << (2 * p.umi_len)) - 1ull, umi_sent = 1ull << (2 * p.umi_len);

	uint64_t cb = cell_cb[cell];
	bool ambient = false;
	if (cat < p.permille_ambient) {
		cb = cb_sent | ((uint64_t(b.v[0]) << 32 | b.v[1]) & cb_mask);
		ambient = true;
	} else if (cat < p.permille_ambient + p.permille_neighbour) {
		const uint32_t pos = mulhi32(b.v[0], p.cb_len);               // substitution position
		const uint32_t sub = 1u + mulhi32(b.v[1], 3u);                // 1..3: always a different base
		const uint64_t base = (cb >> (2 * pos)) & 3ull;
		cb = (cb & ~(3ull << (2 * pos))) | (((base + sub) & 3ull) << (2 * pos));
	}
	r.cb = cb;

	// molecule pool of (cell, gene): expected reads / reads_per_molecule, at least 1
	const uint64_t t = uint64_t(cdf_width(cell_cdf, cell)) * uint64_t(cdf_width(gene_cdf, g));   // prob * 2^64
	const uint64_t pool = mulhi64(t, p.n_effective) / p.reads_per_molecule + 1ull;
	const uint64_t mol = (uint64_t(a.v[3]) << 32 | b.v[2]) % pool;
	uint64_t h = mix64(mix64(uint64_t(cell) * 0x9E3779B97F4A7C15ull + g) ^ (mol * 0xC2B2AE3D27D4EB4Full) ^ p.seed);
	if (ambient) h = mix64(h ^ (uint64_t(b.v[0]) << 32 | b.v[1]));     // ambient barcodes do not share pools
	r.umi = umi_sent | (h & umi_mask);

	const uint32_t ig = mulhi32(b.v[3], 1000u);
	const uint32_t mk = mulhi32(a.v[3] ^ 0xA5A5A5A5u, 1000u);
	uint32_t mark = 2u;                                               // exon
	if (mk < p.permille_intron) mark = 4u;
	else if (mk < p.permille_intron + p.permille_exon_na) mark = 3u;  // exon + not annotated
	uint32_t chr;
	if (ig < p.permille_intergenic) { r.gene = 0xFFFFFFFFu; chr = (b.v[2] >> 7) % p.n_chr; }
	else { r.gene = g; chr = g % p.n_chr; }
	r.aux = chr | (mark << 16);
	return r;
}

__global__ __launch_bounds__(256) void synth_kernel(dropest_synth_params p, const uint64_t *cell_cb, const uint32_t *cell_cdf,
                                                    const uint32_t *gene_cdf, uint64_t first, uint64_t n,
                                                    uint64_t *cb, uint64_t *umi, uint32_t *gene, uint32_t *aux) {
	const uint64_t stride = uint64_t(gridDim.x) * 256;
	for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
		const SynthRead r = synth_read(p, cell_cb, cell_cdf, gene_cdf, first + i);
		cb[i] = r.cb; umi[i] = r.umi; gene[i] = r.gene; aux[i] = r.aux;
	}
}

}  // namespace dropest
