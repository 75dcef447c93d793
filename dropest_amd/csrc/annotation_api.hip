// annotation_api.hip -- include/dropest_annotation.h: the flat gene annotation on the device and the per-read decision
// of ReadParamsParser::get_gene_from_reference (ReadParamsParser.cpp:92-176).  Integer work, one thread per read.
#include "../../include/dropest_annotation.h"
#include "util.h"

#include <string>

using namespace dropest;

struct dropest_annotation {
	int device = 0;
	dropest_flat_annotation f{};          // device pointers
	DevBuf<uint32_t> buf[13];
};

namespace {

thread_local std::string g_ann_error;

constexpr int ANN_CAP = 16;
constexpr uint32_t T_INTRON = 1, T_EXON = 2;   // GtfRecord::RecordType (GtfRecord.h:20-26)

struct ResultSet { uint32_t v[ANN_CAP]; uint32_t n; bool overflow; };   // entries gene << 2 | type, unique

__device__ inline void rs_add(ResultSet &s, uint32_t gene, uint32_t type) {
	const uint32_t e = (gene << 2) | type;
	for (uint32_t i = 0; i < s.n; ++i) if (s.v[i] == e) return;
	if (s.n == ANN_CAP) { s.overflow = true; return; }
	s.v[s.n++] = e;
}

// some span with end > s and start < e (IntervalsContainer::get_intervals, IntervalsContainer.h:216-238)
__device__ inline bool spans_intersect(const uint32_t *start, const uint32_t *end, uint32_t b, uint32_t e2, uint32_t s, uint32_t e) {
	uint32_t lo = b, hi = e2;
	while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (end[mid] <= s) lo = mid + 1; else hi = mid; }
	return lo < e2 && start[lo] < e;
}

// RefGenesContainer::get_gene_info for [s, e) on chromosome c (RefGenesContainer.cpp:179-211)
__device__ inline void collect(const dropest_flat_annotation &f, uint32_t c, uint32_t s, uint32_t e, ResultSet &out) {
	const uint32_t b = f.chr_seg_begin[c], hi0 = f.chr_seg_begin[c + 1];
	uint32_t lo = b, hi = hi0;
	while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (f.seg_end[mid] <= s) lo = mid + 1; else hi = mid; }
	uint32_t seen[ANN_CAP], n_seen = 0;
	for (uint32_t seg = lo; seg < hi0 && f.seg_start[seg] < e; ++seg) {
		for (uint32_t k = f.seg_tr_begin[seg]; k < f.seg_tr_begin[seg + 1]; ++k) {
			const uint32_t t = f.seg_tr[k];
			bool dup = false;
			for (uint32_t j = 0; j < n_seen; ++j) dup |= seen[j] == t;
			if (dup) continue;
			if (n_seen == ANN_CAP) { out.overflow = true; return; }
			seen[n_seen++] = t;
			const bool ex = spans_intersect(f.exon_start, f.exon_end, f.tr_exon_begin[t], f.tr_exon_begin[t + 1], s, e);
			const bool in = spans_intersect(f.intron_start, f.intron_end, f.tr_intron_begin[t], f.tr_intron_begin[t + 1], s, e);
			const uint32_t g = f.tr_gene[t];
			if (!ex && !in) { if (!f.use_introns_from_gtf) rs_add(out, g, T_INTRON); continue; }   // :198-202
			if (ex) rs_add(out, g, T_EXON);
			if (in) rs_add(out, g, T_INTRON);
		}
	}
}

__device__ inline int type_bit(uint32_t e) { return (e & 3u) == T_EXON ? 2 : 4; }   // Mark::add(RecordType), UMI.cpp:87-100

// ReadParamsParser::find_exon (:153-172): the exon of the set if all its exons name one gene
__device__ inline bool find_exon(const ResultSet &s, uint32_t &exon) {
	exon = 0xFFFFFFFFu;
	for (uint32_t i = 0; i < s.n; ++i) {
		if ((s.v[i] & 3u) != T_EXON) continue;
		if (exon == 0xFFFFFFFFu) { exon = s.v[i]; continue; }
		if ((exon >> 2) != (s.v[i] >> 2)) return false;
	}
	return true;
}

__global__ __launch_bounds__(256) void annotate_reads_kernel(dropest_flat_annotation f, uint64_t n, const int32_t *__restrict__ chr,
                                                             const uint32_t *__restrict__ position, const uint32_t *__restrict__ end_position,
                                                             uint32_t *__restrict__ gene, int32_t *__restrict__ mark) {
	const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
	if (i >= n) return;
	gene[i] = 0xFFFFFFFFu;
	const int32_t c = chr[i];
	if (c < 0 || uint32_t(c) >= f.n_chr) { mark[i] = -1; return; }
	ResultSet s1, s2;
	s1.n = s2.n = 0; s1.overflow = s2.overflow = false;
	collect(f, uint32_t(c), position[i], position[i] + 1, s1);                                    // the two end points (:96-101)
	if (end_position[i] >= 1) collect(f, uint32_t(c), end_position[i] - 1, end_position[i], s2);
	if (s1.overflow || s2.overflow) { mark[i] = -2; return; }
	int m = 0;
	if (s1.n == 0 && s2.n == 0) { mark[i] = 0; return; }
	if (s1.n == 1 && s2.n == 1) {
		if ((s1.v[0] >> 2) == (s2.v[0] >> 2)) { m = type_bit(s1.v[0]) | type_bit(s2.v[0]); gene[i] = s1.v[0] >> 2; }
		mark[i] = m; return;
	}
	if (s1.n <= 1 && s2.n <= 1) {
		const uint32_t e = s1.n ? s1.v[0] : s2.v[0];
		gene[i] = e >> 2;
		mark[i] = type_bit(e) | 1; return;                                                        // + HAS_NOT_ANNOTATED
	}
	if (s1.n == 0 || s2.n == 0) { mark[i] = 0; return; }
	uint32_t e1, e2;
	if (!find_exon(s1, e1) || !find_exon(s2, e2)) { mark[i] = 0; return; }
	if (e1 != 0xFFFFFFFFu && e2 != 0xFFFFFFFFu) {
		if ((e1 >> 2) != (e2 >> 2)) { mark[i] = 0; return; }
		gene[i] = e1 >> 2;
		mark[i] = type_bit(e1) | type_bit(e2); return;
	}
	mark[i] = 0;
}

}  // namespace

extern "C" {

const char *dropest_annotation_last_error(void) { return g_ann_error.c_str(); }

int dropest_annotation_create(int device, const dropest_flat_annotation *flat, dropest_annotation **out) {
	try {
		if (!flat || !out) throw InvalidError("null argument");
		HIP_CHECK(hipSetDevice(device));
		auto *a = new dropest_annotation();
		a->device = device;
		a->f = *flat;
		const uint32_t n_cover = flat->n_seg ? flat->seg_tr_begin[flat->n_seg] : 0u;
		const uint32_t n_exon = flat->n_tr ? flat->tr_exon_begin[flat->n_tr] : 0u, n_intron = flat->n_tr ? flat->tr_intron_begin[flat->n_tr] : 0u;
		struct Item { const uint32_t *src; size_t n; const uint32_t **dst; };
		const Item items[13] = {
			{flat->chr_seg_begin, size_t(flat->n_chr) + 1, &a->f.chr_seg_begin}, {flat->seg_start, flat->n_seg, &a->f.seg_start},
			{flat->seg_end, flat->n_seg, &a->f.seg_end}, {flat->seg_tr_begin, size_t(flat->n_seg) + 1, &a->f.seg_tr_begin},
			{flat->seg_tr, n_cover, &a->f.seg_tr}, {flat->tr_gene, flat->n_tr, &a->f.tr_gene},
			{flat->tr_exon_begin, size_t(flat->n_tr) + 1, &a->f.tr_exon_begin}, {flat->tr_intron_begin, size_t(flat->n_tr) + 1, &a->f.tr_intron_begin},
			{flat->exon_start, n_exon, &a->f.exon_start}, {flat->exon_end, n_exon, &a->f.exon_end},
			{flat->intron_start, n_intron, &a->f.intron_start}, {flat->intron_end, n_intron, &a->f.intron_end}, {nullptr, 0, nullptr}};
		for (int i = 0; i < 12; ++i) {
			a->buf[i].alloc(items[i].n ? items[i].n : 1);
			if (items[i].n) HIP_CHECK(hipMemcpy(a->buf[i].p, items[i].src, items[i].n * 4, hipMemcpyHostToDevice));
			*items[i].dst = a->buf[i].p;
		}
		*out = a;
		return 0;
	} catch (const std::exception &e) { g_ann_error = e.what(); return 1; }
}

void dropest_annotation_destroy(dropest_annotation *a) { delete a; }

int dropest_annotation_query(dropest_annotation *a, uint64_t n, const int32_t *chr, const uint32_t *position, const uint32_t *end_position,
                             uint32_t *gene, int32_t *mark) {
	try {
		if (!a || (n && (!chr || !position || !end_position || !gene || !mark))) throw InvalidError("null argument");
		if (!n) return 0;
		HIP_CHECK(hipSetDevice(a->device));
		DevBuf<int32_t> d_chr, d_mark; DevBuf<uint32_t> d_pos, d_end, d_gene;
		d_chr.alloc(n); d_mark.alloc(n); d_pos.alloc(n); d_end.alloc(n); d_gene.alloc(n);
		HIP_CHECK(hipMemcpy(d_chr.p, chr, n * 4, hipMemcpyHostToDevice));
		HIP_CHECK(hipMemcpy(d_pos.p, position, n * 4, hipMemcpyHostToDevice));
		HIP_CHECK(hipMemcpy(d_end.p, end_position, n * 4, hipMemcpyHostToDevice));
		hipLaunchKernelGGL(annotate_reads_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, nullptr, a->f, n, d_chr.p, d_pos.p, d_end.p,
		                   d_gene.p, d_mark.p);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(hipMemcpy(gene, d_gene.p, n * 4, hipMemcpyDeviceToHost));
		HIP_CHECK(hipMemcpy(mark, d_mark.p, n * 4, hipMemcpyDeviceToHost));
		return 0;
	} catch (const std::exception &e) { g_ann_error = e.what(); return 1; }
}

// the same for arrays that are in this GPU's memory already (the device BAM path, bgzf_api.hip): asynchronous on `stream`
int dropest_annotation_query_device(dropest_annotation *a, void *stream, uint64_t n, const int32_t *d_chr, const uint32_t *d_position,
                                    const uint32_t *d_end_position, uint32_t *d_gene, int32_t *d_mark) {
	try {
		if (!a || (n && (!d_chr || !d_position || !d_end_position || !d_gene || !d_mark))) throw InvalidError("null argument");
		if (!n) return 0;
		HIP_CHECK(hipSetDevice(a->device));
		hipLaunchKernelGGL(annotate_reads_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, hipStream_t(stream), a->f, n, d_chr, d_position, d_end_position, d_gene, d_mark);
		HIP_CHECK(hipGetLastError());
		return 0;
	} catch (const std::exception &e) { g_ann_error = e.what(); return 1; }
}

uint32_t dropest_annotation_genes(const dropest_annotation *a) { return a ? a->f.n_genes : 0u; }
int dropest_annotation_device(const dropest_annotation *a) { return a ? a->device : -1; }

}  // extern "C"
