// util.h -- small host/device helpers shared by every kernel file of the dropEst hot path (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <atomic>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <algorithm>
#include <chrono>
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

// one spin of a polling loop on the host
inline void host_cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
	__builtin_ia32_pause();
#elif defined(__aarch64__)
	__asm__ __volatile__("yield");
#endif
}

// Waiting for a stream / an event WITHOUT going to sleep on an interrupt.  A pass has about a dozen points where the host needs
// a count from the device before it can size the next launch; with the ROCm default (HSA_ENABLE_INTERRUPT=1) a blocked thread
// takes 50-100 us -- on a loaded host milliseconds -- to run again after the signal, and the GPU idles meanwhile.  Polling the
// completion signal (hipStreamQuery / hipEventQuery read it without blocking) is local to the calling thread and needs no
// process-wide setting; after 50 ms of polling the wait falls back to the blocking call.  DROPEST_NO_SPIN_WAIT=1: always block.
inline bool spin_wait_enabled() { static const bool on = getenv("DROPEST_NO_SPIN_WAIT") == nullptr; return on; }
inline hipError_t stream_wait(hipStream_t st) {
	if (spin_wait_enabled()) {
		const auto t0 = std::chrono::steady_clock::now();
		for (uint32_t it = 0;; ++it) {
			const hipError_t e = hipStreamQuery(st);
			if (e != hipErrorNotReady) return e;
			if ((it & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
			host_cpu_relax();
		}
	}
	return hipStreamSynchronize(st);
}
inline hipError_t event_wait(hipEvent_t ev) {
	if (spin_wait_enabled()) {
		const auto t0 = std::chrono::steady_clock::now();
		for (uint32_t it = 0;; ++it) {
			const hipError_t e = hipEventQuery(ev);
			if (e != hipErrorNotReady) return e;
			if ((it & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
			host_cpu_relax();
		}
	}
	return hipEventSynchronize(ev);
}

// ---- debug allocator (tests and soak scripts; every switch is an environment variable read at allocation time) ----
// Every DevBuf allocation is numbered and registered (pointer, bytes, ordinal, call site).  On top of that:
//   DROPEST_POISON_ALLOC=<byte>   fill every fresh allocation with one byte (round 2)
//   DROPEST_POISON_SEED=<s>       fill every allocation -- fresh or recycled -- with a pseudo-random pattern of (s, ordinal)
//   DROPEST_DEBUG_POOL=1          released blocks are recycled WITHOUT being cleared (what a buffer pool does: a block then
//                                 holds real stale data of an earlier stage, the adversary that found the round-2 bug)
//   DROPEST_POISON_ZERO=a:b       allocations with ordinal in [a, b) are zero-filled instead (bisecting a dependence)
//   DROPEST_ALLOC_TRACE=1         one stderr line per allocation: ordinal, bytes, file:line of the caller
// dev_debug_poison_all(seed) overwrites every live, non-persistent registered block (the buffers ensure() keeps across
// passes of a context): dropest_debug_poison_scratch in the C-ABI.  None of this is on the product path: with no variable set
// the cost is one map insert per hipMalloc.
static __global__ void poison_fill_kernel(uint32_t *p, size_t words, uint64_t seed) {
	for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < words; i += size_t(gridDim.x) * blockDim.x)
		p[i] = uint32_t(mix64(seed + i * 0x9E3779B97F4A7C15ull) >> 16);
}
// The debug switches, read once (and again on dropest_debug_refresh): they were five getenv() calls per allocation, one under the registry's lock.
// With none of them set -- the product path -- allocations do not touch the registry at all: no lock shared by the threads of all
// shards, no map.  dropest_debug_poison_scratch needs the registry: it asks for DROPEST_DEBUG_REGISTRY=1 (or any other switch).
struct DevDebug {
	bool pool, trace, trace_verbose, poison_seed, poison_alloc, poison_zero, any;
	long long seed; int alloc_byte; unsigned long long zero_a, zero_b;
	static DevDebug read_env() {
		{
			DevDebug x{};
			x.pool = getenv("DROPEST_DEBUG_POOL") != nullptr;
			if (const char *t = getenv("DROPEST_ALLOC_TRACE")) { x.trace = true; x.trace_verbose = atoi(t) > 1; }
			if (const char *s = getenv("DROPEST_POISON_SEED")) { x.poison_seed = true; x.seed = atoll(s); }
			if (const char *b = getenv("DROPEST_POISON_ALLOC")) { x.poison_alloc = true; x.alloc_byte = atoi(b); }
			if (const char *z = getenv("DROPEST_POISON_ZERO")) x.poison_zero = sscanf(z, "%llu:%llu", &x.zero_a, &x.zero_b) == 2;
			x.any = x.pool || x.trace || x.poison_seed || x.poison_alloc || x.poison_zero || getenv("DROPEST_DEBUG_REGISTRY") != nullptr;
			return x;
		}
	}
	// An immutable snapshot behind an atomic pointer: refresh() publishes a NEW one (the old ones stay allocated -- a few dozen bytes per
	// refresh, tests only), so a shard thread that reads the switches while another thread refreshes them sees one consistent set.
	static std::atomic<const DevDebug *> &slot() { static std::atomic<const DevDebug *> d{new DevDebug(read_env())}; return d; }
	static const DevDebug &get() { return *slot().load(std::memory_order_acquire); }
	// has the registry ever been on in this process?  Blocks registered then must leave `live` when they are freed, whatever the switches say now
	static std::atomic<bool> &ever_on() { static std::atomic<bool> b{false}; return b; }
	static void refresh() { slot().store(new DevDebug(read_env()), std::memory_order_release); }   // dropest_debug_refresh: tests switch the variables inside one process
};
struct DevRegistry {
	struct Entry { size_t bytes; uint64_t ordinal; bool persistent; int device; };
	struct Block { void *p; size_t bytes; int device; };
	std::mutex mu;
	std::map<void *, Entry> live;
	std::vector<Block> pool;
	struct Site { uint64_t ordinal; size_t bytes; const char *file; int line; bool recycled; };
	std::vector<Site> sites;   // DROPEST_ALLOC_TRACE: call site of every allocation, kept for dropest_debug_alloc_site
	uint64_t next = 0;
	static DevRegistry &get() { static DevRegistry r; return r; }
	static void fill_random(void *p, size_t bytes, uint64_t seed) {
		(void)hipDeviceSynchronize();
		const size_t words = bytes / 4;
		if (words) hipLaunchKernelGGL(poison_fill_kernel, dim3(unsigned(std::min<size_t>((words + 255) / 256, 4096))), dim3(256), 0, nullptr, static_cast<uint32_t *>(p), words, seed);
		if (bytes & 3) (void)hipMemset(static_cast<char *>(p) + words * 4, int(seed & 0xFF), bytes & 3);
		(void)hipDeviceSynchronize();
	}
	void *allocate(size_t bytes, const char *file, int line) {
		const DevDebug &dbg = DevDebug::get();
		if (!dbg.any) {   // the product path: hipMalloc and nothing else
			void *p = nullptr;
			const hipError_t e = hipMalloc(&p, bytes);
			if (e != hipSuccess) throw DeviceError(std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e) + " (" + file + ":" + std::to_string(line) + ")");
			return p;
		}
		DevDebug::ever_on().store(true, std::memory_order_relaxed);
		void *p = nullptr;
		bool recycled = false;
		uint64_t ordinal;
		int device = 0;
		(void)hipGetDevice(&device);
		{
			std::lock_guard<std::mutex> lk(mu);
			ordinal = next++;
			if (dbg.pool) {   // smallest block that fits and is not more than twice as large
				size_t best = pool.size();
				for (size_t i = 0; i < pool.size(); ++i)
					if (pool[i].device == device && pool[i].bytes >= bytes && pool[i].bytes <= bytes * 2 + (size_t(1) << 16) && (best == pool.size() || pool[i].bytes < pool[best].bytes)) best = i;
				if (best != pool.size()) { p = pool[best].p; pool[best] = pool.back(); pool.pop_back(); recycled = true; }
			}
		}
		if (!p) {
			hipError_t e = hipMalloc(&p, bytes);
			if (e != hipSuccess) {   // give the recycled blocks back before giving up
				trim_pool();
				e = hipMalloc(&p, bytes);
			}
			if (e != hipSuccess) throw DeviceError(std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e) + " (" + file + ":" + std::to_string(line) + ")");
		}
		{
			std::lock_guard<std::mutex> lk(mu);
			live[p] = Entry{bytes, ordinal, false, device};   // (the device it was allocated on: what a pooled block is recycled for)
			if (dbg.trace && sites.size() < (size_t(1) << 20)) sites.push_back(Site{ordinal, bytes, file, line, recycled});
		}
		if (dbg.trace_verbose) fprintf(stderr, "[alloc] #%llu %zu B %s %s:%d\n", (unsigned long long)ordinal, bytes, recycled ? "recycled" : "fresh", file, line);
		if (dbg.poison_zero && ordinal >= dbg.zero_a && ordinal < dbg.zero_b) { (void)hipDeviceSynchronize(); (void)hipMemset(p, 0, bytes); (void)hipDeviceSynchronize(); }
		else if (dbg.poison_seed) fill_random(p, bytes, mix64(uint64_t(dbg.seed) * 0x100000001B3ull + ordinal));
		else if (dbg.poison_alloc) { if (!recycled) { (void)hipDeviceSynchronize(); (void)hipMemset(p, dbg.alloc_byte, bytes); (void)hipDeviceSynchronize(); } }
		return p;
	}
	void release(void *p) {
		const DevDebug &dbg = DevDebug::get();
		// (a block registered while the switches were on and freed with them off must still leave `live`: poison_all would otherwise
		// write into freed -- or reused -- memory once the registry is switched on again)
		if (!dbg.any && !DevDebug::ever_on().load(std::memory_order_relaxed)) { (void)hipFree(p); return; }
		size_t bytes = 0;
		int device = 0;
		{
			std::lock_guard<std::mutex> lk(mu);
			auto it = live.find(p);
			if (it != live.end()) { bytes = it->second.bytes; device = it->second.device; live.erase(it); }
			if (bytes && dbg.pool && pool.size() < 512) { pool.push_back(Block{p, bytes, device}); return; }
		}
		(void)hipFree(p);
	}
	void trim_pool() {
		std::vector<Block> drop;
		{ std::lock_guard<std::mutex> lk(mu); drop.swap(pool); }
		for (auto &b : drop) (void)hipFree(b.p);
	}
	void set_persistent(void *p) { if (!DevDebug::get().any) return; std::lock_guard<std::mutex> lk(mu); auto it = live.find(p); if (it != live.end()) it->second.persistent = true; }
	size_t poison_all(uint64_t seed) {   // every live block that is not marked persistent
		std::vector<std::pair<void *, Entry>> todo;
		{ std::lock_guard<std::mutex> lk(mu); for (auto &kv : live) if (!kv.second.persistent) todo.push_back(kv); }
		for (auto &kv : todo) fill_random(kv.first, kv.second.bytes, mix64(seed * 0x100000001B3ull + kv.second.ordinal));
		return todo.size();
	}
};

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
	void release() { if (p) { DevRegistry::get().release(p); p = nullptr; n = 0; } }
	void alloc(size_t count, const char *file = __builtin_FILE(), int line = __builtin_LINE()) {
		release();
		if (count == 0) count = 1;
		p = static_cast<T *>(DevRegistry::get().allocate(count * sizeof(T), file, line));
		n = count;
	}
	void ensure(size_t count, const char *file = __builtin_FILE(), int line = __builtin_LINE()) { if (count > n) alloc(count, file, line); }
	// inputs that outlive a pass (reads, whitelists, qualities): dev_debug_poison_all leaves them alone
	void mark_persistent() { if (p) DevRegistry::get().set_persistent(p); }
	size_t bytes() const { return n * sizeof(T); }
};

// RAII pinned host buffer (results land here so that D2H runs at PCIe rate and callers get zero-copy views)
struct PinnedRegistry {
	std::mutex mu;
	std::map<void *, size_t> live;
	static PinnedRegistry &get() { static PinnedRegistry r; return r; }
	size_t poison_all(uint64_t seed) {   // debug: staging buffers kept across passes hold what the previous pass left
		std::lock_guard<std::mutex> lk(mu);
		for (auto &kv : live) {
			uint32_t *w = static_cast<uint32_t *>(kv.first);
			for (size_t i = 0; i < kv.second / 4; ++i) w[i] = uint32_t(mix64(seed + i) >> 16);
		}
		return live.size();
	}
};
template <typename T>
struct PinnedBuf {
	T *p = nullptr;
	size_t n = 0;
	PinnedBuf() = default;
	PinnedBuf(const PinnedBuf &) = delete;
	PinnedBuf &operator=(const PinnedBuf &) = delete;
	~PinnedBuf() { release(); }
	void release() {
		if (!p) return;
		{ auto &r = PinnedRegistry::get(); std::lock_guard<std::mutex> lk(r.mu); r.live.erase(p); }
		(void)hipHostFree(p); p = nullptr; n = 0;
	}
	// exactly `count` elements, with hipHostMalloc flags of the caller's choice (hipHostMallocNonCoherent: memory the CPU caches -- staging that host
	// threads fill and the device's DMA engines read)
	void ensure_exact(size_t count, unsigned flags) {
		if (count <= n) return;
		release();
		HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p), count * sizeof(T), flags));
		n = count;
		{ auto &r = PinnedRegistry::get(); std::lock_guard<std::mutex> lk(r.mu); r.live[p] = count * sizeof(T); }
	}
	void ensure(size_t count) {
		if (count <= n) return;
		release();
		size_t cap = count + count / 8 + 16;
		HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p), cap * sizeof(T), hipHostMallocDefault));
		n = cap;
		{ auto &r = PinnedRegistry::get(); std::lock_guard<std::mutex> lk(r.mu); r.live[p] = cap * sizeof(T); }
	}
};

// ---- wave / block primitives (256- or 512-thread blocks, 64-lane waves) ----

__device__ inline uint32_t lane_id() { return threadIdx.x & 63u; }
// Streaming accesses (data read or written exactly once per pass): the non-temporal forms keep them from displacing the
// randomly accessed tables (barcode table, gene -> chromosome) in L2.  DROPEST_STREAM_NT=0 at compile time restores plain accesses.
#ifndef DROPEST_STREAM_NT
#define DROPEST_STREAM_NT 1
#endif
typedef uint32_t nt_v4u32 __attribute__((ext_vector_type(4)));
typedef unsigned long long nt_v2u64 __attribute__((ext_vector_type(2)));
__device__ inline uint4 stream_load_u32x4(const uint32_t *p) {
#if DROPEST_STREAM_NT
	const nt_v4u32 v = __builtin_nontemporal_load(reinterpret_cast<const nt_v4u32 *>(p));
#else
	const nt_v4u32 v = *reinterpret_cast<const nt_v4u32 *>(p);
#endif
	return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline ulonglong2 stream_load_u64x2(const unsigned long long *p) {
#if DROPEST_STREAM_NT
	const nt_v2u64 v = __builtin_nontemporal_load(reinterpret_cast<const nt_v2u64 *>(p));
#else
	const nt_v2u64 v = *reinterpret_cast<const nt_v2u64 *>(p);
#endif
	return make_ulonglong2(v.x, v.y);
}
__device__ inline void stream_store_u64x2(unsigned long long *p, unsigned long long a, unsigned long long b) {
	nt_v2u64 v; v.x = a; v.y = b;
#if DROPEST_STREAM_NT
	__builtin_nontemporal_store(v, reinterpret_cast<nt_v2u64 *>(p));
#else
	*reinterpret_cast<nt_v2u64 *>(p) = v;
#endif
}
__device__ inline void stream_store_u32x4(uint32_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
	nt_v4u32 v; v.x = a; v.y = b; v.z = c; v.w = d;
#if DROPEST_STREAM_NT
	__builtin_nontemporal_store(v, reinterpret_cast<nt_v4u32 *>(p));
#else
	*reinterpret_cast<nt_v4u32 *>(p) = v;
#endif
}
__device__ inline void stream_store_u32(uint32_t *p, uint32_t a) {
#if DROPEST_STREAM_NT
	__builtin_nontemporal_store(a, p);
#else
	*p = a;
#endif
}

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
