// synth_api.hip -- C entry points of the synthetic stream generator and the small device-memory helpers
// (include/dropest_synth.h).  Bench / test input plumbing; not part of the reference's interface.

#include "synth.h"

#include <cstring>

using namespace dropest;

extern "C" {

int dropest_synth_generate_host(const dropest_synth_params *p, uint64_t first, uint64_t n, uint64_t *cb, uint64_t *umi,
                                uint32_t *gene, uint32_t *aux) {
	if (!p || !p->cell_cb || !p->cell_cdf || !p->gene_cdf || p->n_cells == 0 || p->n_genes == 0 || p->n_chr == 0 ||
	    p->cb_len == 0 || p->cb_len > 31 || p->umi_len == 0 || p->umi_len > 31 || p->reads_per_molecule == 0)
		return 1;
	for (uint64_t i = 0; i < n; ++i) {
		const SynthRead r = synth_read(*p, p->cell_cb, p->cell_cdf, p->gene_cdf, first + i);
		cb[i] = r.cb; umi[i] = r.umi; gene[i] = r.gene; aux[i] = r.aux;
	}
	return 0;
}

int dropest_synth_generate_device(const dropest_synth_params *p, int device, uint64_t first, uint64_t n, uint64_t *d_cb,
                                  uint64_t *d_umi, uint32_t *d_gene, uint32_t *d_aux) {
	if (!p || !p->cell_cb || !p->cell_cdf || !p->gene_cdf || p->n_cells == 0 || p->n_genes == 0 || p->n_chr == 0 ||
	    p->cb_len == 0 || p->cb_len > 31 || p->umi_len == 0 || p->umi_len > 31 || p->reads_per_molecule == 0)
		return 1;
	try {
		HIP_CHECK(hipSetDevice(device));
		DevBuf<uint64_t> ccb; DevBuf<uint32_t> ccdf, gcdf;
		ccb.alloc(p->n_cells); ccdf.alloc(p->n_cells); gcdf.alloc(p->n_genes);
		HIP_CHECK(hipMemcpy(ccb.p, p->cell_cb, size_t(p->n_cells) * 8, hipMemcpyHostToDevice));
		HIP_CHECK(hipMemcpy(ccdf.p, p->cell_cdf, size_t(p->n_cells) * 4, hipMemcpyHostToDevice));
		HIP_CHECK(hipMemcpy(gcdf.p, p->gene_cdf, size_t(p->n_genes) * 4, hipMemcpyHostToDevice));
		if (n) {
			const uint64_t want = (n + 255) / 256;
			const unsigned blocks = unsigned(want < 8192 ? want : 8192);
			hipLaunchKernelGGL(synth_kernel, dim3(blocks), dim3(256), 0, nullptr, *p, ccb.p, ccdf.p, gcdf.p, first, n, d_cb,
			                   d_umi, d_gene, d_aux);
			HIP_CHECK(hipGetLastError());
		}
		HIP_CHECK(hipDeviceSynchronize());
		return 0;
	} catch (const std::exception &e) {
		std::fprintf(stderr, "dropest_synth_generate_device: %s\n", e.what());
		return 2;
	}
}

int dropest_dev_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}
int dropest_dev_alloc(int device, uint64_t bytes, void **out) {
	if (hipSetDevice(device) != hipSuccess) return 1;
	return hipMalloc(out, bytes ? bytes : 1) == hipSuccess ? 0 : 2;
}
int dropest_dev_free(int device, void *p) {
	if (hipSetDevice(device) != hipSuccess) return 1;
	return hipFree(p) == hipSuccess ? 0 : 2;
}
int dropest_dev_copy_to_host(int device, void *dst, const void *d_src, uint64_t bytes) {
	if (hipSetDevice(device) != hipSuccess) return 1;
	return hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
int dropest_dev_copy_from_host(int device, void *d_dst, const void *src, uint64_t bytes) {
	if (hipSetDevice(device) != hipSuccess) return 1;
	return hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 2;
}
int dropest_dev_sync(int device) {
	if (hipSetDevice(device) != hipSuccess) return 1;
	return hipDeviceSynchronize() == hipSuccess ? 0 : 2;
}

}  // extern "C"
