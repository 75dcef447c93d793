// probe: H2D / D2H rate of a hipHostMalloc'ed staging buffer right after host threads wrote it (sort_filtered's pattern)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
	const size_t n = size_t(58) << 20;
	void *h, *d; hipStream_t st;
	CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	CK(hipHostMalloc(&h, n, hipHostMallocDefault)); CK(hipMalloc(&d, n));
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
	for (int rep = 0; rep < 6; ++rep) {
		const int nt = rep < 3 ? 1 : 8;
		std::vector<std::thread> pool;
		auto t0 = now();
		for (int t = 0; t < nt; ++t) pool.emplace_back([&, t] { std::memset(static_cast<char *>(h) + n / nt * t, rep + 1, n / nt); });
		for (auto &t : pool) t.join();
		auto t1 = now();
		CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
		auto t2 = now();
		CK(hipMemcpyAsync(h, d, n / 6, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
		auto t3 = now();
		std::printf("threads %d: fill %.2f ms, H2D %.2f ms (%.1f GB/s), D2H of 1/6 %.2f ms (%.1f GB/s)\n", nt, ms(t0, t1), ms(t1, t2), n / ms(t1, t2) / 1e6,
		            ms(t2, t3), n / 6 / ms(t2, t3) / 1e6);
	}
	return 0;
}
