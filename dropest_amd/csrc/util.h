// util.h -- small host/device helpers shared by every kernel file of the dropEst hot path (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace dropest {

struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };
struct InvalidError : std::runtime_error { using std::runtime_error::runtime_error; };
struct RangeError : std::runtime_error { using std::runtime_error::runtime_error; };
struct UnsupportedError : std::runtime_error { using std::runtime_error::runtime_error; };
struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };

#define HIP_CHECK(expr)                                                                                  \
	do {                                                                                                 \
		hipError_t e_ = (expr);                                                                          \
		if (e_ != hipSuccess)                                                                            \
			throw ::dropest::DeviceError(std::string(#expr) + ": " + hipGetErrorString(e_) + " (" +       \
			                             __FILE__ + ":" + std::to_string(__LINE__) + ")");              \
	} while (0)

// splitmix64 finaliser: the hash used for the barcode table and for owner(cb) in the multi-GPU shard map
__host__ __device__ inline uint64_t mix64(uint64_t x) {
	x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
	x ^= x >> 27; x *= 0x94d049bb133111ebull;
	x ^= x >> 31;
	return x;
}

__host__ __device__ inline int bit_length(uint64_t x) { return x ? 64 - __builtin_clzll(x) : 0; }

constexpr int WAVE = 64;   // gfx950 wavefront

// RAII device buffer
template <typename T>
struct DevBuf {
	T *p = nullptr;
	size_t n = 0;
	DevBuf() = default;
	DevBuf(const DevBuf &) = delete;
	DevBuf &operator=(const DevBuf &) = delete;
	DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
	DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
	~DevBuf() { release(); }
	void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
	void alloc(size_t count) {
		release();
		if (count == 0) count = 1;
		HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
		n = count;
		// DROPEST_POISON_ALLOC=<byte>: fresh device memory reads as zero pages in practice; tests run with a poison byte to find
		// code that leans on that
		static const int poison = [] { const char *e = getenv("DROPEST_POISON_ALLOC"); return e ? atoi(e) : -1; }();
		if (poison >= 0) { (void)hipDeviceSynchronize(); (void)hipMemset(p, poison, count * sizeof(T)); (void)hipDeviceSynchronize(); }
	}
	void ensure(size_t count) { if (count > n) alloc(count); }
	size_t bytes() const { return n * sizeof(T); }
};

// RAII pinned host buffer (results land here so that D2H runs at PCIe rate and callers get zero-copy views)
template <typename T>
struct PinnedBuf {
	T *p = nullptr;
	size_t n = 0;
	PinnedBuf() = default;
	PinnedBuf(const PinnedBuf &) = delete;
	PinnedBuf &operator=(const PinnedBuf &) = delete;
	~PinnedBuf() { release(); }
	void release() { if (p) { (void)hipHostFree(p); p = nullptr; n = 0; } }
	void ensure(size_t count) {
		if (count <= n) return;
		release();
		size_t cap = count + count / 8 + 16;
		HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p), cap * sizeof(T), hipHostMallocDefault));
		n = cap;
	}
};

// ---- wave / block primitives (256- or 512-thread blocks, 64-lane waves) ----

__device__ inline uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ inline uint32_t wave_id() { return threadIdx.x >> 6; }

__device__ inline uint32_t wave_incl_scan_u32(uint32_t v) {
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t o = __shfl_up(v, d, 64);
		if (lane_id() >= uint32_t(d)) v += o;
	}
	return v;
}

// Workgroup barrier that orders LDS traffic only: waits for this wave's LDS / scalar-memory operations (lgkmcnt) but
// not for its outstanding global loads and stores (vmcnt), then s_barrier.  For kernels whose threads communicate
// through LDS alone, global stores of one tile then stay in flight under the work on the next one.
// s_waitcnt immediate (gfx9 encoding): vmcnt = 63 (bits 3:0 and 15:14), expcnt = 7 (6:4), lgkmcnt = 0 (11:8).
__device__ inline void lds_barrier() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
	__builtin_amdgcn_s_waitcnt(0xC07F);
	__builtin_amdgcn_s_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// exclusive scan over the block; `scratch` needs blockDim.x/64 + 1 entries; returns prefix, sets total
template <int THREADS, bool LDS_ONLY = false>
__device__ inline uint32_t block_excl_scan_u32(uint32_t v, uint32_t *scratch, uint32_t &total) {
	constexpr int NW = THREADS / 64;
	uint32_t incl = wave_incl_scan_u32(v);
	if (lane_id() == 63) scratch[wave_id()] = incl;
	if (LDS_ONLY) lds_barrier(); else __syncthreads();
	if (threadIdx.x == 0) {
		uint32_t run = 0;
#pragma unroll
		for (int w = 0; w < NW; ++w) { uint32_t t = scratch[w]; scratch[w] = run; run += t; }
		scratch[NW] = run;
	}
	if (LDS_ONLY) lds_barrier(); else __syncthreads();
	uint32_t res = scratch[wave_id()] + incl - v;
	total = scratch[NW];
	if (LDS_ONLY) lds_barrier(); else __syncthreads();   // scratch may be reused right after
	return res;
}

__device__ inline uint64_t wave_reduce_min_u64(uint64_t v) {
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) { uint64_t o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
	return v;
}
__device__ inline uint64_t wave_reduce_max_u64(uint64_t v) {
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) { uint64_t o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
	return v;
}
__device__ inline uint64_t wave_reduce_or_u64(uint64_t v) {
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) v |= __shfl_xor(v, d, 64);
	return v;
}
__device__ inline uint64_t wave_reduce_and_u64(uint64_t v) {
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) v &= __shfl_xor(v, d, 64);
	return v;
}
__device__ inline uint64_t wave_reduce_add_u64(uint64_t v) {
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
	return v;
}

}  // namespace dropest
