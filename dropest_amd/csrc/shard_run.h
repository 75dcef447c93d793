// shard_run.h -- the Estimation hot path sharded by cell barcode over several MI355X (included by dropest_amd.hip).
//
// The reference has ONE container fed by one thread (Estimation/CellsDataContainer.h:33-123, dropest.cpp:245-252); every
// per-read step of the path is independent per barcode, so the reads are sharded by owner(cb) = mix64(cb) mod n
// (SURVEY.md §8e).  One ShardRun = one shard = one dropest_ctx on one GPU, driven by one host thread -- a process per GPU
// (bench.py under torch.distributed.run: the communicator comes from an ncclUniqueId the launcher distributes) or the
// threads of one process (the C++ facade owning N GPUs; tests put several shards on ONE device).  Everything between the
// collectives is the single-GPU pipeline; the collectives go through a Transport:
//   RcclTransport   RCCL over xGMI: one grouped ncclSend / ncclRecv all-to-all(v) of the five read arrays (the only
//                   data-path collective), all-gathers of small tables, molecule rows of merged cells
//   LocalTransport  shards inside one process: device-to-device copies and a host barrier (several shards per device)
// Steps of one pass: partition by owner (stable) -> all-to-all -> barcode table + key-field agreement -> sort / reduce ->
// whitelist CB merge across shards (search / export / intersect / decide / apply / finish, merge_shard.h) -> UMI merge
// (N-UMIs: random fills follow ONE rand() sequence in global cell order) and filter -> global column order -> every
// shard writes ITS columns of both matrices into host memory shared by the node's shards (all PCIe links at once).
#pragma once

#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <sched.h>
#include <sys/stat.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <mutex>
#include <sys/mman.h>
#include <unistd.h>

namespace dropest {

// ---- RCCL, bound at run time (the library also works on hosts without it: single-GPU use never touches it) -----------
struct RcclApi {
	void *handle = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	static RcclApi &get() {
		static RcclApi api = [] {
			RcclApi a;
			// The RCCL that belongs to the HIP runtime this library runs on: a process may hold two ROCm trees (PyTorch ships its
			// own next to /opt/rocm), and an RCCL from the other tree opens ITS libhsa-runtime64 -- not initialised -- and fails
			// with "no ROCm-capable device".  So: the directory libamdhip64 was loaded from first, then the default search.
			Dl_info hip{};
			if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &hip) && hip.dli_fname) {
				std::string dir(hip.dli_fname);
				const size_t slash = dir.rfind('/');
				if (slash != std::string::npos) {
					dir.resize(slash + 1);
					for (const char *name : {"librccl.so.1", "librccl.so"}) if ((a.handle = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_GLOBAL))) break;
				}
			}
			if (!a.handle) for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
			if (!a.handle) return a;
			auto sym = [&](const char *n) { return dlsym(a.handle, n); };
			a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
			a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
			a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
			a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
			a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
			a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
			a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
			a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
			a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
			if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.GroupStart || !a.GroupEnd || !a.Send || !a.Recv || !a.AllGather) a.handle = nullptr;
			return a;
		}();
		if (!api.handle) throw DeviceError("RCCL (librccl.so) is not available: sharded runs over several GPUs need it");
		return api;
	}
	void check(ncclResult_t r, const char *what) const {
		if (r != ncclSuccess) throw DeviceError(std::string("RCCL ") + what + ": " + (GetErrorString ? GetErrorString(r) : "error"));
	}
};

// ---- transports --------------------------------------------------------------------------------------------------------
struct Transport {
	int rank = 0, world = 1;
	virtual ~Transport() {}
	virtual const char *name() const = 0;
	// all-to-all(v) of n_arrays device arrays at once: array a has elem[a]-byte elements; the block for peer p starts at
	// element sum(send_cnt[< p]) of d_send[a] and lands at element sum(recv_cnt[< p]) of d_recv[a].  Returns when done.
	// in_place: bit a set = the block of array a that stays on this shard is NOT copied (the caller reads it in d_send[a])
	virtual void exchange(int n_arrays, const void *const *d_send, void *const *d_recv, const size_t *elem, const uint64_t *send_cnt,
	                      const uint64_t *recv_cnt, hipStream_t st, uint32_t in_place = 0) = 0;
	// One chunk of such an all-to-all(v): the piece for peer p starts at element send_off[p] of d_send[a] (send_cnt[p] elements) and lands at
	// element recv_off[p] of d_recv[a] (recv_cnt[p] elements).  Enqueued on `st` and NOT waited for where the transport can (RCCL): the
	// caller orders its consumers with an event on `st` -- the next chunk is partitioned, the previous one hashed, while this one travels.
	virtual void exchange_at(int n_arrays, const void *const *d_send, void *const *d_recv, const size_t *elem, const uint64_t *send_off, const uint64_t *send_cnt,
	                         const uint64_t *recv_off, const uint64_t *recv_cnt, hipStream_t st, uint32_t in_place = 0) = 0;
	virtual void gather_host(const void *mine, size_t bytes, void *all) = 0;   // all: world x bytes, rank order
	// every shard's device block to every shard: block of rank p = bytes[p] at d_all + off[p]
	virtual void gather_dev(const void *d_mine, void *d_all, const size_t *off, const size_t *bytes, hipStream_t st) = 0;
	virtual void barrier() = 0;
	// host memory shared by the shards of the node (collective call; at least `bytes`); *d_ptr = this shard's device view
	virtual void *shared_host(int slot, size_t bytes, void **d_ptr) = 0;

	template <class T>
	void gather_vec(const std::vector<T> &mine, std::vector<T> &all, std::vector<size_t> &count) {   // variable-length host rows
		// ONE collective when every rank's rows fit a small fixed slot (the length travels in front of them) -- most of a pass's
		// host gathers are a few hundred bytes, and a collective of staged bytes costs its latency, not its size; a second one
		// with the padded rows only when somebody's did not fit.
		constexpr size_t SLOT = 16384;
		const uint64_t n = mine.size();
		const size_t fit = (SLOT - 8) / sizeof(T);
		std::vector<unsigned char> slot(SLOT, 0), slots(SLOT * size_t(world));
		std::memcpy(slot.data(), &n, 8);
		if (n && n <= fit) std::memcpy(slot.data() + 8, mine.data(), size_t(n) * sizeof(T));
		gather_host(slot.data(), SLOT, slots.data());
		std::vector<uint64_t> ns(size_t(world), 0);
		uint64_t mx = 0;
		for (int p = 0; p < world; ++p) { std::memcpy(&ns[size_t(p)], slots.data() + size_t(p) * SLOT, 8); mx = std::max(mx, ns[size_t(p)]); }
		count.assign(size_t(world), 0);
		all.clear();
		if (mx <= fit) {
			for (int p = 0; p < world; ++p) {
				count[size_t(p)] = size_t(ns[size_t(p)]);
				const T *rows = reinterpret_cast<const T *>(slots.data() + size_t(p) * SLOT + 8);
				all.insert(all.end(), rows, rows + ns[size_t(p)]);
			}
			return;
		}
		std::vector<T> pad(mx), buf(size_t(mx) * size_t(world));
		if (n) std::memcpy(static_cast<void *>(pad.data()), mine.data(), size_t(n) * sizeof(T));
		gather_host(pad.data(), size_t(mx) * sizeof(T), buf.data());
		for (int p = 0; p < world; ++p) {
			count[size_t(p)] = size_t(ns[size_t(p)]);
			all.insert(all.end(), buf.begin() + size_t(p) * mx, buf.begin() + size_t(p) * mx + ns[size_t(p)]);
		}
	}
};

// Shards inside one process.  One hub per group; every member runs on its own host thread.
struct LocalHub {
	int n;
	std::mutex m;
	std::condition_variable cv;
	int arrived = 0;
	uint64_t generation = 0;
	std::vector<const void *> p0, p1;   // per rank: pointers published for the current collective
	struct Shared { void *host = nullptr; size_t bytes = 0; } shared[4];
	bool failed = false;                // a member threw: wake the others instead of deadlocking
	explicit LocalHub(int n_) : n(n_), p0(size_t(n_), nullptr), p1(size_t(n_), nullptr) {}
	~LocalHub() { for (auto &s : shared) if (s.host) (void)hipHostFree(s.host); }
	void barrier() {
		std::unique_lock<std::mutex> lk(m);
		if (failed) throw DeviceError("another shard of the group failed");
		const uint64_t g = generation;
		if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); return; }
		cv.wait(lk, [&] { return generation != g || failed; });
		if (failed && generation == g) throw DeviceError("another shard of the group failed");
	}
	void fail() { std::lock_guard<std::mutex> lk(m); failed = true; cv.notify_all(); }
};

struct LocalTransport : Transport {
	std::shared_ptr<LocalHub> hub;
	LocalTransport(std::shared_ptr<LocalHub> h, int r) : hub(std::move(h)) { rank = r; world = hub->n; }
	const char *name() const override { return "local"; }
	struct Pub { const void *const *d_send; const size_t *elem; const uint64_t *send_cnt; };
	void exchange(int n_arrays, const void *const *d_send, void *const *d_recv, const size_t *elem, const uint64_t *send_cnt,
	              const uint64_t *recv_cnt, hipStream_t st, uint32_t in_place = 0) override {
		HIP_CHECK(stream_wait(st));   // the blocks the peers copy were written on this stream
		Pub pub{d_send, elem, send_cnt};
		hub->p0[size_t(rank)] = &pub;
		hub->barrier();
		uint64_t roff = 0;
		for (int p = 0; p < world; ++p) {
			const Pub *src = static_cast<const Pub *>(hub->p0[size_t(p)]);
			uint64_t soff = 0;
			for (int q = 0; q < rank; ++q) soff += src->send_cnt[q];
			for (int a = 0; a < n_arrays; ++a)
				if (recv_cnt[p] && !(p == rank && (in_place >> a & 1u)))
					HIP_CHECK(hipMemcpyAsync(static_cast<char *>(d_recv[a]) + roff * elem[a], static_cast<const char *>(src->d_send[a]) + soff * elem[a],
					                         size_t(recv_cnt[p]) * elem[a], hipMemcpyDefault, st));
			roff += recv_cnt[p];
		}
		HIP_CHECK(stream_wait(st));
		hub->barrier();   // the senders' buffers may be reused
	}
	struct PubAt { const void *const *d_send; const size_t *elem; const uint64_t *send_off, *send_cnt; };
	void exchange_at(int n_arrays, const void *const *d_send, void *const *d_recv, const size_t *elem, const uint64_t *send_off, const uint64_t *send_cnt,
	                 const uint64_t *recv_off, const uint64_t *recv_cnt, hipStream_t st, uint32_t in_place = 0) override {
		HIP_CHECK(stream_wait(st));   // the pieces the peers copy were written before this point of `st`
		PubAt pub{d_send, elem, send_off, send_cnt};
		hub->p0[size_t(rank)] = &pub;
		hub->barrier();
		for (int p = 0; p < world; ++p) {
			const PubAt *src = static_cast<const PubAt *>(hub->p0[size_t(p)]);
			if (src->send_cnt[rank] != recv_cnt[p]) throw DeviceError("chunked exchange: a peer's piece differs from the agreed size");
			for (int a = 0; a < n_arrays; ++a)
				if (recv_cnt[p] && !(p == rank && (in_place >> a & 1u)))
					HIP_CHECK(hipMemcpyAsync(static_cast<char *>(d_recv[a]) + recv_off[p] * elem[a], static_cast<const char *>(src->d_send[a]) + src->send_off[rank] * elem[a],
					                         size_t(recv_cnt[p]) * elem[a], hipMemcpyDefault, st));
		}
		HIP_CHECK(stream_wait(st));
		hub->barrier();   // the senders' buffers may be reused
	}
	void gather_host(const void *mine, size_t bytes, void *all) override {
		hub->p0[size_t(rank)] = mine;
		hub->barrier();
		for (int p = 0; p < world; ++p) std::memcpy(static_cast<char *>(all) + size_t(p) * bytes, hub->p0[size_t(p)], bytes);
		hub->barrier();
	}
	void gather_dev(const void *d_mine, void *d_all, const size_t *off, const size_t *bytes, hipStream_t st) override {
		HIP_CHECK(stream_wait(st));   // what the peers are about to read was produced on this stream
		hub->p1[size_t(rank)] = d_mine;
		hub->barrier();
		for (int p = 0; p < world; ++p)
			if (bytes[p]) HIP_CHECK(hipMemcpyAsync(static_cast<char *>(d_all) + off[p], hub->p1[size_t(p)], bytes[p], hipMemcpyDefault, st));
		HIP_CHECK(stream_wait(st));
		hub->barrier();
	}
	void barrier() override { hub->barrier(); }
	void *shared_host(int slot, size_t bytes, void **d_ptr) override {
		hub->barrier();
		if (rank == 0) {
			LocalHub::Shared &s = hub->shared[slot];
			if (s.bytes < bytes) {
				if (s.host) HIP_CHECK(hipHostFree(s.host));
				s.host = nullptr; s.bytes = 0;
				const size_t cap = bytes + bytes / 4 + 4096;
				HIP_CHECK(hipHostMalloc(&s.host, cap, hipHostMallocPortable | hipHostMallocMapped));
				s.bytes = cap;
			}
		}
		hub->barrier();
		void *host = hub->shared[slot].host;
		HIP_CHECK(hipHostGetDevicePointer(d_ptr, host, 0));
		return host;
	}
};

// The small host tables of the processes of one node meet in shared memory, not on the device: a pass makes a dozen host collectives of
// a few bytes to a few hundred KB, and each one staged through the GPU (copy in, ncclAllGather, copy out, stream wait) costs ~50-100 us
// of launch and synchronisation latency whatever its size.  HostMailbox: one POSIX shm object per run (named by the run's unique id;
// rank 0 creates it, the others attach, rank 0 unlinks it once all are in), `world` slots of SLOT bytes, twice (two halves used in
// turn), and a sense-reversing barrier on an atomic counter.  A gather = write my slot, barrier, read all slots: ONE barrier -- a rank
// can only be two collectives ahead of another when that one has passed the barrier in between, i.e. has finished reading the half
// about to be overwritten.  Payloads beyond SLOT bytes take the RCCL path.  A rank that does not arrive within TIMEOUT_S (a peer died)
// fails the collective instead of hanging.
struct HostMailbox {
	static constexpr size_t SLOT = size_t(1) << 20;
	// seconds a rank waits for the others (attach, barrier) before it fails the collective for everybody; DROPEST_MAILBOX_TIMEOUT_S (tests)
	static unsigned timeout_s() { static const unsigned t = [] { const char *e = getenv("DROPEST_MAILBOX_TIMEOUT_S"); return e ? unsigned(std::max(1, atoi(e))) : 300u; }(); return t; }
	struct Header { std::atomic<uint32_t> attached, arrived, generation, failed; std::atomic<uint64_t> magic; };
	static uint64_t magic_of(uint64_t token) { return mix64(token ^ 0x6d61696c626f7821ull) | 1ull; }   // never 0: a fresh object reads as zeros
	int rank = 0, world = 1;
	char *base = nullptr;
	size_t bytes = 0;
	uint64_t phase = 0;
	Header *hdr() const { return reinterpret_cast<Header *>(base); }
	char *slot(uint64_t ph, int p) const { return base + 4096 + ((ph & 1) * size_t(world) + size_t(p)) * SLOT; }
	HostMailbox(uint64_t token, int r, int w) : rank(r), world(w) {
		bytes = 4096 + 2 * size_t(w) * SLOT;
		char path[96];
		std::snprintf(path, sizeof(path), "/dropest_mb_%llx_%d", (unsigned long long)token, w);
		int fd = -1;
		const auto t0 = std::chrono::steady_clock::now();
		if (r == 0) {
			shm_unlink(path);   // a leftover of a crashed run with the same id cannot be ours
			fd = shm_open(path, O_CREAT | O_EXCL | O_RDWR, 0600);
			if (fd < 0 || ftruncate(fd, off_t(bytes)) != 0) { if (fd >= 0) { close(fd); shm_unlink(path); } throw DeviceError("cannot create the host mailbox in /dev/shm"); }
		} else {
			for (;;) {   // rank 0 may not be there yet; a file that exists but is not sized yet is not ready either
				fd = shm_open(path, O_RDWR, 0600);
				if (fd >= 0) { struct stat st; if (fstat(fd, &st) == 0 && size_t(st.st_size) >= bytes) break; close(fd); fd = -1; }
				if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s())) throw DeviceError("host mailbox: rank 0 did not create it");
				usleep(200);
			}
		}
		void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		close(fd);
		if (m == MAP_FAILED) { if (r == 0) shm_unlink(path); throw DeviceError("cannot map the host mailbox"); }
		base = static_cast<char *>(m);   // (a fresh shm object reads as zeros: the header starts at 0 / 0 / 0 / 0)
		// rank 0 signs the object it has just created; the others attach only to a signed one (an object of the same name that rank 0 is
		// about to unlink and replace -- a leftover -- never carries this run's word)
		// (a wait that times out leaves the constructor by an exception: no destructor runs, so the mapping -- and on rank 0 the name, which
		// is otherwise unlinked only once everybody is in -- are released here)
		try {
			if (r == 0) hdr()->magic.store(magic_of(token), std::memory_order_release);
			else wait_until([&] { return hdr()->magic.load(std::memory_order_acquire) == magic_of(token); }, "signature", false);
			hdr()->attached.fetch_add(1, std::memory_order_acq_rel);
			wait_until([&] { return hdr()->attached.load(std::memory_order_acquire) >= uint32_t(world); }, "attach");
		} catch (...) {
			munmap(base, bytes); base = nullptr;
			if (r == 0) shm_unlink(path);
			throw;
		}
		if (r == 0) shm_unlink(path);   // the mappings keep it alive; nothing is left behind on a crash
	}
	~HostMailbox() { if (base) munmap(base, bytes); }
	HostMailbox(const HostMailbox &) = delete;
	template <class F> void wait_until(F &&done, const char *what, bool heed_failed = true) {
		const auto t0 = std::chrono::steady_clock::now();
		for (uint32_t spins = 0; !done(); ++spins) {
			if (heed_failed && hdr()->failed.load(std::memory_order_acquire)) throw DeviceError("another shard of the run failed");
			if (spins < 2000) { dropest::host_cpu_relax(); continue; }
			sched_yield();
			if ((spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s())) {
				hdr()->failed.store(1, std::memory_order_release);
				throw DeviceError(std::string("host mailbox: a shard did not arrive (") + what + ")");
			}
		}
	}
	void fail() { if (base) hdr()->failed.store(1, std::memory_order_release); }
	void barrier() {
		Header *h = hdr();
		const uint32_t g = h->generation.load(std::memory_order_acquire);
		if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == uint32_t(world)) {
			h->arrived.store(0, std::memory_order_relaxed);
			h->generation.store(g + 1, std::memory_order_release);
			return;
		}
		wait_until([&] { return h->generation.load(std::memory_order_acquire) != g; }, "barrier");
	}
	bool fits(size_t n) const { return n <= SLOT; }
	void gather(const void *mine, size_t n, void *all) {   // n <= SLOT, the same on every rank
		const uint64_t ph = phase++;
		if (n) std::memcpy(slot(ph, rank), mine, n);
		barrier();
		for (int p = 0; p < world; ++p) if (n) std::memcpy(static_cast<char *>(all) + size_t(p) * n, slot(ph, p), n);
	}
};

// One process per GPU: RCCL for the device data; the small host tables through the HostMailbox above (RCCL all-gathers of staged bytes
// for payloads beyond its slots, or for everything with DROPEST_HOST_COLLECTIVES=rccl).  Shared host memory for the results: a POSIX
// shm object mapped and registered by every process of the node.
struct RcclTransport : Transport {
	ncclComm_t comm = nullptr;
	hipStream_t st0 = nullptr;          // the owning context's stream
	DevBuf<unsigned char> stage_in, stage_out;
	PinnedBuf<unsigned char> h_in, h_out;
	uint64_t token = 0;
	std::unique_ptr<HostMailbox> mailbox;
	struct Shm { void *host = nullptr; size_t bytes = 0; int gen = 0; } shm[4];
	// DROPEST_SHARD_DATAPLANE=shm: the device data of the collectives (exchange, gather_dev) crosses through POSIX shared memory instead of
	// RCCL -- every process stages what it sends in a segment of its own, the peers read it after a barrier.  For runs whose processes
	// share GPUs (RCCL refuses two ranks on one device): the multi-process tests on a one-GPU box run the whole step this way -- the
	// mailbox, the shared result buffers, the sequence of collectives are the real ones, only ncclSend / ncclRecv are stood in for.
	bool shm_plane = false;
	struct Stage { void *host = nullptr; size_t bytes = 0; uint64_t gen = 0; };
	Stage my_stage;
	std::vector<Stage> peer_stage;
	static void stage_path(char *out, size_t n, uint64_t token, int rank, uint64_t gen) { std::snprintf(out, n, "/dropest_dp_%llx_%d_%llu", (unsigned long long)token, rank, (unsigned long long)gen); }
	// collective: every rank makes sure its segment holds `need` bytes, then maps the peers' segments (again) where they were replaced
	void stage_prepare(size_t need) {
		uint64_t row[2] = {my_stage.gen, 0};
		if (need > my_stage.bytes) {
			if (my_stage.host) { char old[128]; stage_path(old, sizeof(old), token, rank, my_stage.gen); munmap(my_stage.host, my_stage.bytes); shm_unlink(old); my_stage.host = nullptr; }
			const size_t cap = need + need / 4 + 4096;
			char path[128];
			stage_path(path, sizeof(path), token, rank, my_stage.gen + 1);
			shm_unlink(path);
			const int fd = shm_open(path, O_CREAT | O_EXCL | O_RDWR, 0600);
			if (fd < 0 || posix_fallocate(fd, 0, off_t(cap)) != 0) { if (fd >= 0) { close(fd); shm_unlink(path); } throw DeviceError("cannot create the staging segment of the shm data plane"); }
			void *m = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
			close(fd);
			if (m == MAP_FAILED) { shm_unlink(path); throw DeviceError("cannot map the staging segment of the shm data plane"); }
			my_stage.host = m; my_stage.bytes = cap; ++my_stage.gen;
			row[0] = my_stage.gen;
		}
		row[1] = my_stage.bytes;
		std::vector<uint64_t> all(size_t(world) * 2);
		gather_host(row, 16, all.data());
		peer_stage.resize(size_t(world));
		for (int p = 0; p < world; ++p) {
			if (p == rank) continue;
			Stage &ps = peer_stage[size_t(p)];
			if (ps.gen == all[size_t(p) * 2] && ps.host) continue;
			if (ps.host) { munmap(ps.host, ps.bytes); ps.host = nullptr; }
			ps.gen = all[size_t(p) * 2]; ps.bytes = size_t(all[size_t(p) * 2 + 1]);
			if (!ps.gen) continue;
			char path[128];
			stage_path(path, sizeof(path), token, p, ps.gen);
			const int fd = shm_open(path, O_RDONLY, 0600);
			void *m = fd >= 0 ? mmap(nullptr, ps.bytes, PROT_READ, MAP_SHARED, fd, 0) : MAP_FAILED;
			if (fd >= 0) close(fd);
			if (m == MAP_FAILED) throw DeviceError("cannot map a peer's staging segment of the shm data plane");
			ps.host = m;
		}
	}
	RcclTransport(int r, int w, const uint8_t id_bytes[128], hipStream_t st) : st0(st) {
		rank = r; world = w;
		for (int i = 0; i < 128; ++i) token = token * 1099511628211ull + id_bytes[i];
		{ const char *dp = getenv("DROPEST_SHARD_DATAPLANE"); shm_plane = dp && std::string(dp) == "shm"; }
		if (shm_plane) { mailbox = std::make_unique<HostMailbox>(token, r, w); return; }   // one host by construction; no communicator
		const RcclApi &api = RcclApi::get();
		ncclUniqueId id;
		static_assert(sizeof(id) == 128, "ncclUniqueId");
		std::memcpy(&id, id_bytes, 128);
		api.check(api.CommInitRank(&comm, w, id, r), "ncclCommInitRank");
		const char *hc = getenv("DROPEST_HOST_COLLECTIVES");
		bool use_mailbox = !(hc && std::string(hc) == "rccl");
		if (use_mailbox && w > 1) {
			// The mailbox is POSIX shared memory: every rank must be on ONE host.  The ranks compare a word that names the host's running
			// kernel instance (boot id, else the host name) over the communicator first; ranks on several hosts keep the RCCL path for
			// their host collectives instead of waiting 300 s for a shm object that will never show up.
			uint64_t mine = 1469598103934665603ull;
			auto fold = [&](const char *s) { for (; *s; ++s) { mine ^= (unsigned char)*s; mine *= 1099511628211ull; } };
			char buf[256] = {0};
			if (FILE *f = fopen("/proc/sys/kernel/random/boot_id", "r")) { if (fgets(buf, sizeof(buf), f)) fold(buf); fclose(f); }
			if (!buf[0] && gethostname(buf, sizeof(buf) - 1) == 0) fold(buf);
			std::vector<uint64_t> all(size_t(w), 0);
			gather_host(&mine, 8, all.data());   // (no mailbox yet: the staged RCCL all-gather)
			for (int p = 0; p < w; ++p) use_mailbox &= all[size_t(p)] == mine;
		}
		if (use_mailbox) mailbox = std::make_unique<HostMailbox>(token, r, w);
	}
	~RcclTransport() override {
		for (auto &ps : peer_stage) if (ps.host) munmap(ps.host, ps.bytes);
		if (my_stage.host) { char path[128]; stage_path(path, sizeof(path), token, rank, my_stage.gen); munmap(my_stage.host, my_stage.bytes); shm_unlink(path); }
		for (auto &s : shm) if (s.host) { (void)hipHostUnregister(s.host); munmap(s.host, s.bytes); }
		if (comm) (void)RcclApi::get().CommDestroy(comm);
	}
	const char *name() const override { return shm_plane ? "shm" : "rccl"; }
	// the shm data plane's all-to-all(v): [header: my send counts][array 0, all destinations][array 1 ...] in my segment, a barrier, then
	// every rank copies the blocks addressed to it out of its peers' segments
	void exchange_shm(int n_arrays, const void *const *d_send, void *const *d_recv, const size_t *elem, const uint64_t *send_cnt,
	                  const uint64_t *recv_cnt, hipStream_t st) {
		uint64_t total = 0;
		for (int p = 0; p < world; ++p) total += send_cnt[p];
		size_t need = size_t(world) * 8;
		for (int a = 0; a < n_arrays; ++a) need += ((size_t(total) * elem[a] + 63) & ~size_t(63));
		stage_prepare(need);
		char *mine = static_cast<char *>(my_stage.host);
		std::memcpy(mine, send_cnt, size_t(world) * 8);
		size_t at = size_t(world) * 8;
		HIP_CHECK(stream_wait(st));   // what is sent was produced on this stream
		for (int a = 0; a < n_arrays; ++a) {
			if (total) HIP_CHECK(hipMemcpy(mine + at, d_send[a], size_t(total) * elem[a], hipMemcpyDeviceToHost));
			at += (size_t(total) * elem[a] + 63) & ~size_t(63);
		}
		barrier();
		uint64_t roff = 0;
		for (int p = 0; p < world; ++p) {
			if (p != rank && recv_cnt[p]) {
				const char *peer = static_cast<const char *>(peer_stage[size_t(p)].host);
				const uint64_t *pc = reinterpret_cast<const uint64_t *>(peer);
				uint64_t ptotal = 0, poff = 0;
				for (int q = 0; q < world; ++q) { if (q < rank) poff += pc[q]; ptotal += pc[q]; }
				if (pc[rank] != recv_cnt[p]) throw DeviceError("shm data plane: a peer's send count differs from the agreed one");
				size_t pat = size_t(world) * 8;
				for (int a = 0; a < n_arrays; ++a) {
					HIP_CHECK(hipMemcpy(static_cast<char *>(d_recv[a]) + roff * elem[a], peer + pat + size_t(poff) * elem[a], size_t(recv_cnt[p]) * elem[a], hipMemcpyHostToDevice));
					pat += (size_t(ptotal) * elem[a] + 63) & ~size_t(63);
				}
			}
			roff += recv_cnt[p];
		}
		barrier();   // nobody overwrites its segment while a peer still reads it
	}
	void exchange(int n_arrays, const void *const *d_send, void *const *d_recv, const size_t *elem, const uint64_t *send_cnt,
	              const uint64_t *recv_cnt, hipStream_t st, uint32_t in_place = 0) override {
		if (shm_plane) {
			uint64_t soff = 0, roff = 0;
			for (int p = 0; p < rank; ++p) { soff += send_cnt[p]; roff += recv_cnt[p]; }
			for (int a = 0; a < n_arrays; ++a)
				if (send_cnt[rank] && !(in_place >> a & 1u)) HIP_CHECK(hipMemcpyAsync(static_cast<char *>(d_recv[a]) + roff * elem[a], static_cast<const char *>(d_send[a]) + soff * elem[a],
				                                             size_t(send_cnt[rank]) * elem[a], hipMemcpyDeviceToDevice, st));
			exchange_shm(n_arrays, d_send, d_recv, elem, send_cnt, recv_cnt, st);
			HIP_CHECK(stream_wait(st));
			return;
		}
		const RcclApi &api = RcclApi::get();
		// the block a shard keeps never leaves the device: a plain copy (a self send / recv pair moves it at a fifth of the rate)
		{
			uint64_t soff = 0, roff = 0;
			for (int p = 0; p < rank; ++p) { soff += send_cnt[p]; roff += recv_cnt[p]; }
			for (int a = 0; a < n_arrays; ++a)
				if (send_cnt[rank] && !(in_place >> a & 1u)) HIP_CHECK(hipMemcpyAsync(static_cast<char *>(d_recv[a]) + roff * elem[a], static_cast<const char *>(d_send[a]) + soff * elem[a],
				                                             size_t(send_cnt[rank]) * elem[a], hipMemcpyDeviceToDevice, st));
		}
		api.check(api.GroupStart(), "ncclGroupStart");
		for (int a = 0; a < n_arrays; ++a) {
			uint64_t soff = 0, roff = 0;
			for (int p = 0; p < world; ++p) {
				if (p == rank) { soff += send_cnt[p]; roff += recv_cnt[p]; continue; }
				if (send_cnt[p]) api.check(api.Send(static_cast<const char *>(d_send[a]) + soff * elem[a], size_t(send_cnt[p]) * elem[a], ncclUint8, p, comm, st), "ncclSend");
				if (recv_cnt[p]) api.check(api.Recv(static_cast<char *>(d_recv[a]) + roff * elem[a], size_t(recv_cnt[p]) * elem[a], ncclUint8, p, comm, st), "ncclRecv");
				soff += send_cnt[p]; roff += recv_cnt[p];
			}
		}
		api.check(api.GroupEnd(), "ncclGroupEnd");
		// (no wait here: whatever reads the received blocks is queued behind them on this stream -- cb_sample, cb_insert --, and the send
		// buffers are not rewritten before the next pass's partition, on this stream too)
	}
	void exchange_at(int n_arrays, const void *const *d_send, void *const *d_recv, const size_t *elem, const uint64_t *send_off, const uint64_t *send_cnt,
	                 const uint64_t *recv_off, const uint64_t *recv_cnt, hipStream_t st, uint32_t in_place = 0) override {
		for (int a = 0; a < n_arrays; ++a)   // the piece a shard keeps never leaves the device
			if (send_cnt[rank] && !(in_place >> a & 1u))
				HIP_CHECK(hipMemcpyAsync(static_cast<char *>(d_recv[a]) + recv_off[rank] * elem[a], static_cast<const char *>(d_send[a]) + send_off[rank] * elem[a],
				                         size_t(send_cnt[rank]) * elem[a], hipMemcpyDeviceToDevice, st));
		if (shm_plane) {   // processes that share GPUs: the pieces cross through the staging segments, with the host in between
			uint64_t total = 0;
			for (int p = 0; p < world; ++p) total += p == rank ? 0 : send_cnt[p];
			size_t need = size_t(world) * 16;
			for (int a = 0; a < n_arrays; ++a) need += ((size_t(total) * elem[a] + 63) & ~size_t(63));
			stage_prepare(need);
			char *mine = static_cast<char *>(my_stage.host);
			// header: for every peer, where its piece starts in each array's staged run (in elements) and its length
			uint64_t *hdr = reinterpret_cast<uint64_t *>(mine);
			uint64_t run = 0;
			for (int p = 0; p < world; ++p) { hdr[2 * p] = run; hdr[2 * p + 1] = p == rank ? 0 : send_cnt[p]; run += hdr[2 * p + 1]; }
			HIP_CHECK(stream_wait(st));
			size_t at = size_t(world) * 16;
			for (int a = 0; a < n_arrays; ++a) {
				for (int p = 0; p < world; ++p)
					if (p != rank && send_cnt[p])
						HIP_CHECK(hipMemcpy(mine + at + size_t(hdr[2 * p]) * elem[a], static_cast<const char *>(d_send[a]) + send_off[p] * elem[a], size_t(send_cnt[p]) * elem[a], hipMemcpyDeviceToHost));
				at += (size_t(total) * elem[a] + 63) & ~size_t(63);
			}
			barrier();
			for (int p = 0; p < world; ++p) {
				if (p == rank || !recv_cnt[p]) continue;
				const char *peer = static_cast<const char *>(peer_stage[size_t(p)].host);
				const uint64_t *ph = reinterpret_cast<const uint64_t *>(peer);
				if (ph[2 * rank + 1] != recv_cnt[p]) throw DeviceError("shm data plane: a peer's piece differs from the agreed size");
				uint64_t ptotal = 0;
				for (int q = 0; q < world; ++q) ptotal += ph[2 * q + 1];
				size_t pat = size_t(world) * 16;
				for (int a = 0; a < n_arrays; ++a) {
					HIP_CHECK(hipMemcpy(static_cast<char *>(d_recv[a]) + recv_off[p] * elem[a], peer + pat + size_t(ph[2 * rank]) * elem[a], size_t(recv_cnt[p]) * elem[a], hipMemcpyHostToDevice));
					pat += (size_t(ptotal) * elem[a] + 63) & ~size_t(63);
				}
			}
			barrier();   // nobody overwrites its segment while a peer still reads it
			return;
		}
		const RcclApi &api = RcclApi::get();
		api.check(api.GroupStart(), "ncclGroupStart");
		for (int a = 0; a < n_arrays; ++a)
			for (int p = 0; p < world; ++p) {
				if (p == rank) continue;
				if (send_cnt[p]) api.check(api.Send(static_cast<const char *>(d_send[a]) + send_off[p] * elem[a], size_t(send_cnt[p]) * elem[a], ncclUint8, p, comm, st), "ncclSend");
				if (recv_cnt[p]) api.check(api.Recv(static_cast<char *>(d_recv[a]) + recv_off[p] * elem[a], size_t(recv_cnt[p]) * elem[a], ncclUint8, p, comm, st), "ncclRecv");
			}
		api.check(api.GroupEnd(), "ncclGroupEnd");
	}
	void gather_host(const void *mine, size_t bytes, void *all) override {
		if (mailbox && mailbox->fits(bytes)) { mailbox->gather(mine, bytes, all); return; }
		if (shm_plane) {   // no communicator: a payload beyond a slot goes through the mailbox slot by slot
			std::vector<unsigned char> part(HostMailbox::SLOT * size_t(world));
			for (size_t at = 0; at < bytes; at += HostMailbox::SLOT) {
				const size_t n = std::min(HostMailbox::SLOT, bytes - at);
				mailbox->gather(static_cast<const char *>(mine) + at, n, part.data());
				for (int p = 0; p < world; ++p) std::memcpy(static_cast<char *>(all) + size_t(p) * bytes + at, part.data() + size_t(p) * n, n);
			}
			return;
		}
		const RcclApi &api = RcclApi::get();
		const size_t b = std::max<size_t>(bytes, 1);
		stage_in.ensure(b); stage_out.ensure(b * size_t(world)); h_in.ensure(b); h_out.ensure(b * size_t(world));
		std::memcpy(h_in.p, mine, bytes);
		HIP_CHECK(hipMemcpyAsync(stage_in.p, h_in.p, b, hipMemcpyHostToDevice, st0));
		api.check(api.AllGather(stage_in.p, stage_out.p, b, ncclUint8, comm, st0), "ncclAllGather");
		HIP_CHECK(hipMemcpyAsync(h_out.p, stage_out.p, b * size_t(world), hipMemcpyDeviceToHost, st0));
		HIP_CHECK(stream_wait(st0));
		for (int p = 0; p < world; ++p) std::memcpy(static_cast<char *>(all) + size_t(p) * bytes, h_out.p + size_t(p) * b, bytes);
	}
	void gather_dev(const void *d_mine, void *d_all, const size_t *off, const size_t *bytes, hipStream_t st) override {
		if (shm_plane) {
			stage_prepare(bytes[rank] + 64);
			HIP_CHECK(stream_wait(st));
			if (bytes[rank]) HIP_CHECK(hipMemcpy(my_stage.host, d_mine, bytes[rank], hipMemcpyDeviceToHost));
			barrier();
			for (int p = 0; p < world; ++p) {
				if (!bytes[p]) continue;
				if (p == rank) HIP_CHECK(hipMemcpy(static_cast<char *>(d_all) + off[p], d_mine, bytes[p], hipMemcpyDeviceToDevice));
				else HIP_CHECK(hipMemcpy(static_cast<char *>(d_all) + off[p], peer_stage[size_t(p)].host, bytes[p], hipMemcpyHostToDevice));
			}
			barrier();
			return;
		}
		const RcclApi &api = RcclApi::get();
		api.check(api.GroupStart(), "ncclGroupStart");
		for (int p = 0; p < world; ++p) {
			if (bytes[rank]) api.check(api.Send(d_mine, bytes[rank], ncclUint8, p, comm, st), "ncclSend");
			if (bytes[p]) api.check(api.Recv(static_cast<char *>(d_all) + off[p], bytes[p], ncclUint8, p, comm, st), "ncclRecv");
		}
		api.check(api.GroupEnd(), "ncclGroupEnd");   // (stream-ordered like exchange(): the kernels that read d_all follow on `st`)
	}
	void barrier() override { unsigned char x = 0; std::vector<unsigned char> all(static_cast<size_t>(world)); gather_host(&x, 1, all.data()); }
	void *shared_host(int slot, size_t bytes, void **d_ptr) override {
		Shm &s = shm[slot];
		// every rank must take the same branch: agree on the capacity first
		uint64_t want = s.bytes >= bytes ? 0 : uint64_t(bytes + bytes / 4 + 4096);
		std::vector<uint64_t> wants(static_cast<size_t>(world));
		gather_host(&want, 8, wants.data());
		uint64_t cap = 0;
		for (uint64_t w : wants) cap = std::max(cap, w);
		if (cap) {
			cap = std::max<uint64_t>(cap, s.bytes);
			if (s.host) { HIP_CHECK(hipHostUnregister(s.host)); munmap(s.host, s.bytes); s.host = nullptr; s.bytes = 0; }
			++s.gen;
			uint64_t pid0 = uint64_t(getpid());
			std::vector<uint64_t> pids(static_cast<size_t>(world));
			gather_host(&pid0, 8, pids.data());
			char path[128];
			std::snprintf(path, sizeof(path), "/dropest_%llx_%llu_%d_%d", (unsigned long long)token, (unsigned long long)pids[0], slot, s.gen);
			uint64_t ok = 1;
			int fd = -1;
			if (rank == 0) {
				fd = shm_open(path, O_CREAT | O_EXCL | O_RDWR, 0600);
				// The pages are reserved now (a full tmpfs fails here, not with SIGBUS later) -- on the NUMA node of this rank's GPU, where ROCm
				// puts pinned host memory and where the threads that widen the matrices into this buffer run (matrix_decode.h): left to the
				// default policy they land on whatever node this thread happens to run on, and every slot is then written across the sockets.
				h_in.ensure(4096);
				const int node = dropest::numa_node_of(h_in.p);
				unsigned long mask[16] = {0};
				const bool bound = node >= 0 && node < 1024 && (mask[node / 64] |= 1ul << (node % 64), syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, 1024ul) == 0);
				if (fd < 0 || posix_fallocate(fd, 0, off_t(cap)) != 0) ok = 0;
				if (bound) (void)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0ul);
			}
			std::vector<uint64_t> oks(static_cast<size_t>(world));
			gather_host(&ok, 8, oks.data());
			if (!oks[0]) { if (fd >= 0) { close(fd); shm_unlink(path); } throw DeviceError("cannot reserve the shared result buffer in /dev/shm"); }
			if (rank != 0) fd = shm_open(path, O_RDWR, 0600);
			void *host = fd >= 0 ? mmap(nullptr, size_t(cap), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
			if (fd >= 0) close(fd);
			ok = host != MAP_FAILED && hipHostRegister(host, size_t(cap), hipHostRegisterMapped | hipHostRegisterPortable) == hipSuccess;
			gather_host(&ok, 8, oks.data());
			if (rank == 0) shm_unlink(path);   // the mappings keep it alive; nothing is left behind on a crash
			for (uint64_t o : oks) if (!o) throw DeviceError("cannot map / register the shared result buffer");
			s.host = host; s.bytes = size_t(cap);
		}
		HIP_CHECK(hipHostGetDevicePointer(d_ptr, s.host, 0));
		return s.host;
	}
};

// ---- small kernels ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void take_u32_kernel(const uint32_t *__restrict__ table, const uint32_t *__restrict__ pos, uint32_t n, uint32_t *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) out[i] = table[pos[i]];
}
__global__ __launch_bounds__(256) void gather_u32_rows_kernel(const uint32_t *__restrict__ in, const uint32_t *__restrict__ row, uint32_t n, uint32_t width,
                                                              uint32_t *__restrict__ out) {
	for (uint64_t k = uint64_t(blockIdx.x) * 256 + threadIdx.x; k < uint64_t(n) * width; k += uint64_t(gridDim.x) * 256)
		out[k] = in[uint64_t(row[k / width]) * width + k % width];
}
struct ImportGatherArgs { const uint32_t *row; uint32_t n; const unsigned long long *low_all; const uint32_t *col_all[4]; unsigned long long *o_low; uint32_t *o_col[4]; };
__global__ __launch_bounds__(256) void import_gather_kernel(ImportGatherArgs a) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= a.n) return;
	const uint32_t r = a.row[i];
	a.o_low[i] = a.low_all[r];
#pragma unroll
	for (int k = 0; k < 4; ++k) a.o_col[k][i] = a.col_all[k][r];
}
// desc[3c .. 3c+2] = (src_start, dst_start, len) of column c; rows / values go to their places in the global matrix
__global__ __launch_bounds__(256) void place_columns_kernel(const unsigned long long *__restrict__ desc, const uint32_t *__restrict__ src_rows,
                                                            const uint32_t *__restrict__ src_vals, uint32_t *__restrict__ dst_rows, uint32_t *__restrict__ dst_vals) {
	const unsigned long long s = desc[3ull * blockIdx.x], d = desc[3ull * blockIdx.x + 1], len = desc[3ull * blockIdx.x + 2];
	for (unsigned long long t = threadIdx.x; t < len; t += 256) { dst_rows[d + t] = src_rows[s + t]; dst_vals[d + t] = src_vals[s + t]; }
}
// The same, narrowing on the way out (16-bit row indices and values: half the bytes over this shard's PCIe link); a value beyond
// 65534 is stored as 0xFFFF and listed exactly with its GLOBAL entry index (ovf[0] = count, then (position lo, position hi,
// value) triples, at most ovf_cap of them).
__global__ __launch_bounds__(256) void place_columns_narrow_kernel(const unsigned long long *__restrict__ desc, const uint32_t *__restrict__ src_rows,
                                                                   const uint32_t *__restrict__ src_vals, uint16_t *__restrict__ dst_rows, uint16_t *__restrict__ dst_vals,
                                                                   uint32_t *__restrict__ ovf, uint32_t ovf_cap) {
	const unsigned long long s = desc[3ull * blockIdx.x], d = desc[3ull * blockIdx.x + 1], len = desc[3ull * blockIdx.x + 2];
	for (unsigned long long t = threadIdx.x; t < len; t += 256) {
		const uint32_t v = src_vals[s + t];
		dst_rows[d + t] = uint16_t(src_rows[s + t]);
		dst_vals[d + t] = v >= 0xFFFFu ? uint16_t(0xFFFFu) : uint16_t(v);
		if (v >= 0xFFFFu) {
			const uint32_t at = atomicAdd(ovf, 1u);
			if (at < ovf_cap) { ovf[1 + 3 * at] = uint32_t(d + t); ovf[2 + 3 * at] = uint32_t((d + t) >> 32); ovf[3 + 3 * at] = v; }
		}
	}
}
// The BYTE form (dropest_matrix_bytes, include/dropest_amd.h): u8 row delta + u8 value at the entry's GLOBAL place, two bytes per entry
// on the PCIe link.  A workgroup takes one column.  A lane takes SIXTEEN consecutive entries of an aligned group: a group that lies inside
// the column leaves as one 16-byte store per array (a wave writes 1 KB of each array in one piece: the link takes whole lines -- with
// 4-byte stores per lane the placing kernels of a C2 pass reached ~30 GB/s of the link's ~52), the ragged groups at a column's ends byte by
// byte (the neighbouring column may be another shard's).  Entries that do not fit a byte (255 = listed) go to this shard's segment of the
// shared buffer with their global position: a wave counts what its lanes list, takes its places with ONE atomic per kind on the shard's
// counters, and the lanes write their entries there (no LDS, no workgroup barrier); the readers concatenate the segments (the lists
// carry no order).
__global__ __launch_bounds__(256) void place_columns_bytes_kernel(const unsigned long long *__restrict__ desc, const uint32_t *__restrict__ src_rows,
                                                                  const uint32_t *__restrict__ src_vals, uint8_t *__restrict__ dst_delta,
                                                                  uint8_t *__restrict__ dst_val, uint32_t *__restrict__ counters /* [2] */,
                                                                  uint32_t *__restrict__ rl_pos, uint32_t *__restrict__ rl_row,
                                                                  uint32_t *__restrict__ vl_pos, uint32_t *__restrict__ vl_val, uint32_t cap,
                                                                  uint32_t *__restrict__ slot_rows = nullptr, uint32_t *__restrict__ slot_vals = nullptr,
                                                                  uint32_t *flag = nullptr, uint32_t epoch = 0) {
	// flag / epoch (matrix_decode.h): the arrival flag of the chunk of columns placed BEFORE this launch on the stream -- this kernel runs, so
	// that chunk is complete and visible to the host threads that widen it into the 32-bit slots.  slot_rows / slot_vals: the slots
	// themselves (node-shared host memory); a listed entry is written there directly, so the widening needs no list at all.
	if (flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	const unsigned long long s = desc[3ull * blockIdx.x], d = desc[3ull * blockIdx.x + 1], len = desc[3ull * blockIdx.x + 2];
	const uint32_t lane = threadIdx.x & 63u;
	const unsigned long long g0 = d >> 4, g1 = (d + len + 15) >> 4;
	for (unsigned long long gb = g0; gb < g1; gb += blockDim.x) {   // (256 threads for cm's long columns, one wave for cm_raw's mostly short ones)
		const unsigned long long gq = gb + threadIdx.x, first = 16 * gq;
		uint32_t dd[4] = {0, 0, 0, 0}, vv[4] = {0, 0, 0, 0}, inside = 0, n_listed[2] = {0, 0};
		if (gq < g1) {
			uint32_t prev = (first > d && first <= d + len) ? src_rows[s + (first - d) - 1] : 0xFFFFFFFFu;
#pragma unroll
			for (uint32_t b = 0; b < 16; ++b) {
				const unsigned long long g = first + b;
				if (g < d || g >= d + len) continue;
				const unsigned long long t = g - d;
				const uint32_t row = src_rows[s + t], v = src_vals[s + t];
				const uint32_t delta = row - prev;
				prev = row;
				inside |= 1u << b;
				dd[b >> 2] |= (delta >= 255u ? 255u : delta) << (8 * (b & 3u));
				vv[b >> 2] |= (v >= 255u ? 255u : v) << (8 * (b & 3u));
				if (delta >= 255u) { ++n_listed[0]; if (slot_rows) slot_rows[g] = row; }
				if (v >= 255u) { ++n_listed[1]; if (slot_vals) slot_vals[g] = v; }
			}
			if (inside == 0xFFFFu) {
				reinterpret_cast<uint4 *>(dst_delta)[gq] = make_uint4(dd[0], dd[1], dd[2], dd[3]);
				reinterpret_cast<uint4 *>(dst_val)[gq] = make_uint4(vv[0], vv[1], vv[2], vv[3]);
			} else {
#pragma unroll
				for (uint32_t b = 0; b < 16; ++b) if (inside >> b & 1u) { dst_delta[first + b] = uint8_t(dd[b >> 2] >> (8 * (b & 3u))); dst_val[first + b] = uint8_t(vv[b >> 2] >> (8 * (b & 3u))); }
			}
		}
#pragma unroll
		for (uint32_t k = 0; k < 2; ++k) {
			if (!__ballot(n_listed[k] != 0)) continue;
			const uint32_t incl = wave_incl_scan_u32(n_listed[k]);
			const uint32_t total = uint32_t(__shfl(int(incl), 63, 64));
			uint32_t base = 0;
			if (lane == 0) base = atomicAdd(&counters[k], total);
			base = uint32_t(__shfl(int(base), 0, 64));
			uint32_t at = base + incl - n_listed[k];
			if (n_listed[k]) {
				uint32_t *pos = k ? vl_pos : rl_pos, *x = k ? vl_val : rl_row;
				const uint32_t *src = k ? src_vals : src_rows;
#pragma unroll
				for (uint32_t b = 0; b < 16; ++b)
					if ((inside >> b & 1u) && (((k ? vv[b >> 2] : dd[b >> 2]) >> (8 * (b & 3u))) & 0xFFu) == 255u) {
						if (at < cap) { pos[at] = uint32_t(first + b); x[at] = src[s + (first + b - d)]; }
						++at;
					}
			}
		}
	}
}
// cm_raw planned on the device (assemble_raw_device): the shards' real cells, each list ascending by the stream ordinal of the cell's
// first read, merged by rank -- a cell's global column = its local place + the cells of every other shard that come before it.
struct RawPlanTable { uint32_t world, rank; uint64_t off[64]; uint32_t n[64]; };   // shard s: first[n] at all + off, nnz prefix[n + 1] behind it
struct QueryTable { uint32_t world; uint64_t in_off[65], part_off[64], out_off[65], first_ord[64]; };
__global__ __launch_bounds__(256) void answer_queries_kernel(QueryTable t, const uint32_t *__restrict__ q_in, uint32_t n, const uint32_t *__restrict__ part_idx,
                                                             uint32_t *__restrict__ out) {
	const uint32_t k = blockIdx.x * 256 + threadIdx.x;
	if (k >= n) return;
	uint32_t p = 0;
	while (p + 1 < t.world && k >= t.in_off[p + 1]) ++p;
	out[k] = part_idx[t.part_off[p] + q_in[k]];
}
__global__ __launch_bounds__(256) void ordinals_from_answers_kernel(QueryTable t, const uint32_t *__restrict__ back, uint32_t n, unsigned long long *__restrict__ first) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	uint32_t s = 0;
	while (s + 1 < t.world && i >= t.out_off[s + 1]) ++s;
	first[i] = t.first_ord[s] + back[i];
}
__global__ __launch_bounds__(256) void plan_raw_columns_kernel(RawPlanTable t, const unsigned long long *__restrict__ all, const uint32_t *__restrict__ col_cell,
                                                               const unsigned long long *__restrict__ cell_barcode, uint32_t n,
                                                               unsigned long long *__restrict__ desc, unsigned long long *__restrict__ colptr64,
                                                               uint32_t *__restrict__ colptr32, unsigned long long *__restrict__ barcodes,
                                                               uint32_t *__restrict__ h_begin = nullptr, uint32_t *__restrict__ h_end = nullptr) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const unsigned long long *mine = all + t.off[t.rank], *my_pre = mine + t.n[t.rank];
	const unsigned long long f = mine[i];
	unsigned long long col = i, at = my_pre[i];
	for (uint32_t s = 0; s < t.world; ++s) {
		if (s == t.rank) continue;
		const unsigned long long *fs = all + t.off[s];
		uint32_t lo = 0, hi = t.n[s];
		while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (fs[mid] < f) lo = mid + 1; else hi = mid; }
		col += lo; at += fs[t.n[s] + lo];
	}
	desc[3ull * i] = my_pre[i]; desc[3ull * i + 1] = at; desc[3ull * i + 2] = my_pre[i + 1] - my_pre[i];
	colptr64[col] = at; colptr32[col] = uint32_t(at); barcodes[col] = cell_barcode[col_cell[i]];
	if (h_begin) { h_begin[i] = uint32_t(at); h_end[i] = uint32_t(at + my_pre[i + 1] - my_pre[i]); }   // (pinned: what the host threads that widen this shard's columns walk by)
}
__global__ void copy_words_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, uint32_t n) { if (threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x]; }
// local read position -> global stream ordinal: position p came from source rank s = the block [recv_off[s], recv_off[s+1])
// it lies in, as that rank's idx[p]-th resident read
// (every shard's resident reads are ONE contiguous range of the stream and the ranges ascend with the rank: the blocks a
// shard receives, concatenated in source-rank order, are then in stream order, which is what makes a read's POSITION
// on its shard a valid first-seen ordinal there)
struct OrdinalMap { const uint32_t *idx; uint32_t world; uint64_t recv_off[65]; uint64_t first_ord[64]; };
__device__ inline unsigned long long to_global_ordinal(const OrdinalMap &m, uint32_t p) {
	uint32_t s = 0;
	while (s + 1 < m.world && p >= m.recv_off[s + 1]) ++s;
	return m.first_ord[s] + (m.idx ? m.idx[p] : p - uint32_t(m.recv_off[s]));
}
__global__ __launch_bounds__(256) void ordinals_kernel(OrdinalMap m, const uint32_t *__restrict__ pos, uint32_t n, unsigned long long *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) out[i] = pos[i] == 0xFFFFFFFFu ? ~0ull : to_global_ordinal(m, pos[i]);
}

// -u across shards: smallest global ordinal of every UMI over the shards' tables; then the table becomes a rank table
__global__ __launch_bounds__(256) void min_rows_kernel(const unsigned long long *__restrict__ all, uint32_t rows, uint64_t n, unsigned long long *__restrict__ out) {
	for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
		unsigned long long m = ~0ull;
		for (uint32_t r = 0; r < rows; ++r) { const unsigned long long v = all[uint64_t(r) * n + i]; m = v < m ? v : m; }
		out[i] = m;
	}
}
// rows of `len` bytes gathered by index: out[i] = in[idx[i]] (the quality strings follow the stable partition of their reads)
__global__ __launch_bounds__(256) void gather_byte_rows_kernel(const uint8_t *__restrict__ in, const uint32_t *__restrict__ idx, uint32_t n, uint32_t len,
                                                               uint8_t *__restrict__ out) {
	for (uint64_t k = uint64_t(blockIdx.x) * 256 + threadIdx.x; k < uint64_t(n) * len; k += uint64_t(gridDim.x) * 256) {
		const uint32_t i = uint32_t(k / len), b = uint32_t(k % len);
		out[k] = in[uint64_t(idx[i]) * len + b];
	}
}
// -M across shards: the UMI histograms of the shards added up
__global__ __launch_bounds__(256) void sum_rows_kernel(const uint32_t *__restrict__ all, uint32_t rows, uint64_t n, uint32_t *__restrict__ out) {
	for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
		uint32_t sum = 0;
		for (uint32_t r = 0; r < rows; ++r) sum += all[uint64_t(r) * n + i];
		out[i] = sum;
	}
}
__global__ __launch_bounds__(256) void ranks_to_table_kernel(const unsigned long long *__restrict__ sorted_ord, const uint32_t *__restrict__ code, uint32_t n,
                                                             uint32_t *__restrict__ table) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) table[code[i]] = sorted_ord[i] == ~0ull ? 0xFFFFFFFFu : i;
}

}  // namespace dropest

// CellsDataContainer::compare_cells (CellsDataContainer.cpp:329-344) on table rows: (requested_genes, requested_umis,
// umis_number, barcode STRING) ascending.  Clean codes of one length order like their strings; anything else is decoded.
template <class Row>
static bool compare_cells_rows(const Row &p, const Row &q, const std::vector<std::string> &side) {
	if (p.req_genes != q.req_genes) return p.req_genes < q.req_genes;
	if (p.req_umis != q.req_umis) return p.req_umis < q.req_umis;
	const dropest::u64 pu = dropest::u64(size_t(p.total_umis)), qu = dropest::u64(size_t(q.total_umis));   // Cell::umis_number casts the int stat to size_t
	if (pu != qu) return pu < qu;
	const bool plain = !((p.barcode | q.barcode) & dropest::ESCAPE_BIT) && dropest::bit_length(p.barcode) == dropest::bit_length(q.barcode);
	if (plain) return p.barcode < q.barcode;
	return dropest::decode_code(p.barcode, side) < dropest::decode_code(q.barcode, side);
}

// ---- one shard ---------------------------------------------------------------------------------------------------------
struct dropest_shard {
	using u64 = dropest::u64;
	using u32 = dropest::u32;
	std::unique_ptr<dropest_ctx> ctx;
	std::unique_ptr<dropest::Transport> tr;
	int rank = 0, world = 1;
	bool force_exchange = false, trace = false;
	// this shard's contiguous range of the stream (resident, adopted)
	const u64 *r_cb = nullptr, *r_umi = nullptr;
	const u32 *r_gene = nullptr, *r_aux = nullptr;
	uint64_t n_res = 0, first_ordinal = 0;
	dropest::ReadStore pushed;       // reads pushed from host memory (dropest_shard_push_reads)
	// UMI quality strings of the resident reads (dropest_shard_set_umi_qualities): they travel with their reads in the exchange
	dropest::DevBuf<uint8_t> r_qual, p_qual, x_qual;
	dropest::DevBuf<uint8_t> r_qual_lens, p_qual_lens, x_qual_lens;   // (dropest_shard_set_umi_qualities_var) one length per read, <= r_qlen
	bool r_qual_var = false;
	u32 r_qlen = 0;
	uint64_t r_qual_reads = 0;
	bool r_have_qual = false;
	// partition / exchange buffers
	dropest::DevBuf<u64> p_cb, p_umi, x_cb, x_umi;
	dropest::DevBuf<u32> p_gene, p_aux, p_idx, x_gene, x_aux, x_idx;
	dropest::DevBuf<unsigned char> part_scratch;
	std::vector<uint64_t> send_cnt, recv_cnt, recv_off, first_ord;   // first_ord[p]: first stream ordinal of rank p's range
	bool exchanged = false;
	// the exchange record: 12 bytes per read (w0 u64 + w1 u32, k_misc.h ExchangePack) when the field widths of ALL shards allow
	// it, the five arrays (28 bytes) otherwise; the read's position in its source range (idx) only travels when a whole-table
	// conversion needs it on the device (-u) -- the few positions the pass asks about otherwise are resolved by asking the source
	dropest::DevBuf<u64> p_w0, x_w0;
	dropest::DevBuf<u32> p_w1, x_w1;
	bool packed = false, idx_exchanged = false, allow_packed = true, unpacked = false;
	bool exact_widths = false;   // option "exact_widths" / after a sampled pass missed a wide field: the histogram pass reads all four columns
	// The all-to-all in chunks under the partition and the table build (option "exchange_chunks": 0 = 4 chunks from 2^22 resident reads on, 1 =
	// one piece, k = k chunks): the blocks of every owner are written chunk by chunk; chunk k leaves on the exchange stream as soon as it is
	// written, while chunk k + 1 is partitioned, and the barcode table takes chunk k of every source's block (dropest_ctx::recv_chunks)
	// while chunk k + 1 travels.  For packed records without the index column and without quality strings (C2 / C5).
	int exchange_chunks = 0;
	bool chunked_now = false;
	hipStream_t xchg_stream = nullptr;
	std::vector<hipEvent_t> ev_scat, ev_recv;
	dropest::DevBuf<u32> d_chunk_bounds, d_chunk_cnt;
	dropest::ExchangePack exch_pack{};
	void unpack_exchanged();
	int rec_bytes = 28;
	std::vector<uint64_t> send_off;                                  // first read of every destination's block in p_idx
	// global table of the real cells (identical on every shard after a step)
	struct GRow {
		u64 barcode, first_global;
		u32 n_genes, req_genes, req_umis, local_id;
		int32_t total_umis, total_reads;
		u32 rank, pad;
	};
	std::vector<GRow> G;
	// results: global CSC of both matrices (rows / values in the node-shared host buffer)
	struct Mat {
		uint64_t ncols = 0, nnz = 0; std::vector<u64> colptr, col_barcode;
		// what the accessors hand out: the vectors above, or (cm_raw planned on the device) arrays in the shared buffer
		const u64 *colptr_p = nullptr, *barcode_p = nullptr; const u32 *colptr32_p = nullptr;
		const u32 *rows = nullptr, *vals = nullptr;               // 32-bit form (in the shared buffer, or widened on demand)
		bool narrow = false;                                      // the shared buffer holds the 16-bit form
		const uint16_t *rows16 = nullptr, *vals16 = nullptr;
		std::vector<u64> ovf_pos; std::vector<u32> ovf_val;       // entries beyond 65534, ascending position
		std::vector<u32> wide_rows, wide_vals; bool widened = false;
		// the byte form (dropest_matrix_bytes): deltas / values in the shared buffer, every shard's lists in its segment behind them
		bool bytes = false, lists_ready = false;
		const uint8_t *delta8 = nullptr, *vals8 = nullptr; const char *segments = nullptr; size_t seg_bytes = 0; u32 list_cap = 0;
		std::vector<u32> colptr32, rl_pos, rl_row, vl_pos, vl_val;
		// the 32-bit dgCMatrix slots i / x of the GLOBAL matrix in the shared buffer (option "slots_matrix", on): every shard's host threads
		// widen ITS columns from the byte form as the chunks of columns land (matrix_decode.h) -- the step ends where the plain context's does
		// (ResultsPrinter::create_matrix, Estimation/ResultsPrinter.cpp:433-442)
		bool slots = false;                                       // this step's shared buffer holds the slots (and no byte form)
		u32 *slot_rows = nullptr, *slot_vals = nullptr;           // host views
		u32 *d_slot_rows = nullptr, *d_slot_vals = nullptr;       // this shard's device views of them (the 32-bit fall-back places there)
		bool shipped = false;                                     // this shard has columns on their way (the context's widening job)
		std::vector<u32> dec_begin, dec_end;                      // this shard's columns: global begin / end of each (cm: planned on the host)
		std::vector<u32> col_cell, col_start;                     // ... their cells and local offsets (cm; cm_raw's are in raw_plan)
		dropest::PinnedBuf<u32> h_begin, h_end;                   // ... written by the device when the columns are planned there (cm_raw)
		const unsigned long long *d_descr = nullptr; u32 nc = 0; uint64_t local_nnz = 0;   // what the fall-back needs again
		// the byte form of a slots step, made on demand (dropest_shard_matrix_bytes)
		std::vector<uint8_t> enc_delta, enc_vals; bool encoded = false;
	} mat[2];
	bool slots_matrix = true;                                     // option "slots_matrix": the step ends with the 32-bit slots (needs byte_matrix)
	void finish_slots(int slot);
	void encode_bytes(Mat &M);
	bool narrow_matrix = true;                                    // option "narrow_matrix": 16-bit matrices when every gene id fits
	bool byte_matrix = true;                                      // option "byte_matrix": the byte form (any gene id); wins over narrow_matrix
	uint64_t byte_list_cap = 0;                                   // option "byte_list_cap": entries a shard may list per kind (0: 2^20)
	bool wide_now[2] = {false, false};                            // this step: a shard's lists of the matrix overflowed, its columns are placed again without the byte form
	bool lists_overflowed(const Mat &M) const;
	dropest::DevBuf<u32> d_list_count;
	void collect_lists(Mat &M);
	// cm_raw without a host table of all the real cells (option "raw_on_device", on): see assemble_raw_device
	bool raw_on_device = true, raw_device_now = false;
	struct RawPlan { std::vector<u32> col_cell, col_start, query; std::vector<u64> pre; std::vector<uint64_t> counts, q_out, q_in; uint64_t ncols = 0, nnz = 0; } raw_plan;
	bool plan_raw();
	void assemble_raw_device();
	struct SharedLayout { char *host; void *dev; size_t base, off_val, off_seg, seg_bytes, off_slots, slots_stride; dropest::u32 list_cap; bool bytes, narrow, slots; };
	SharedLayout open_shared(Mat &M, int slot, size_t head_bytes);
	// local_start[nc]: every column's offset among this shard's entries (what the placing launches and the widening are cut by)
	void place_columns(Mat &M, int slot, bool filtered_m, const SharedLayout &L, const unsigned long long *d_descr, dropest::u32 nc, uint64_t local_nnz, hipStream_t st = nullptr,
	                   const dropest::u32 *local_start = nullptr);
	// cm_raw's columns leave on a stream of their own, under the host collectives and the assembly of cm (the plain context's prefetch,
	// for shards): the placing kernel is the longest piece of the end of a pass and needs nothing of what follows it
	hipStream_t place_stream = nullptr;
	hipEvent_t ev_place = nullptr;
	~dropest_shard() {
		if (xchg_stream) { (void)dropest::stream_wait(xchg_stream); (void)hipStreamDestroy(xchg_stream); }
		for (hipEvent_t e : ev_scat) (void)hipEventDestroy(e);
		for (hipEvent_t e : ev_recv) (void)hipEventDestroy(e);
		if (ctx) for (auto &R : ctx->mat) R.settle();   // (the widening threads write into the transport's shared buffer: they leave before it is unmapped)
		if (place_stream) { (void)dropest::stream_wait(place_stream); (void)hipStreamDestroy(place_stream); } if (ev_place) (void)hipEventDestroy(ev_place); }
	dropest::DevBuf<u64> d_desc_raw, d_ord_out64;
	dropest::DevBuf<u32> d_ord_pos, d_ord_out;
	dropest::DevBuf<u64> d_plan_mine, d_plan_all;
	dropest::DevBuf<u32> d_q, d_q_in, d_q_ans, d_q_back, d_col_cell;
	dropest::DevBuf<u32> d_ovf;
	std::vector<std::pair<u64, u64>> merged_barcodes;   // (source, target) barcode of every merged cell, ascending source
	bool merged_pending = false;
	void name_merged_pairs() {   // world == 1: (source id, target id) of the context -> barcodes
		if (!merged_pending) return;
		merged_pending = false;
		dropest_ctx &c = *ctx;
		HIP_CHECK(hipSetDevice(c.cfg.device));
		std::vector<dropest::u64> bc(c.n_cells);
		if (c.n_cells) c.fetch(bc.data(), c.cell_cb.p, size_t(c.n_cells) * 8);
		for (auto const &pr : c.merge_pairs) merged_barcodes.emplace_back(bc[size_t(pr.first)], bc[size_t(pr.second)]);
		std::sort(merged_barcodes.begin(), merged_barcodes.end());
	}
	// column descriptors of assemble_matrix, one set per matrix: cm_raw's (when the host plans it) are still on their way to the device --
	// and read by its placing kernel, and kept for finish_slots -- while cm's are written (nothing waits in between since the emit does not)
	dropest::DevBuf<u64> d_desc_m[2];
	dropest::PinnedBuf<u64> h_desc_m[2], h_plan;
	bool desc_in_flight[2] = {false, false};
	dropest::DevBuf<u32> d_tmp32;
	std::map<std::string, dropest::KernelStat> phases;

	struct Phase {
		dropest_shard *s; const char *name; std::chrono::steady_clock::time_point t0;
		Phase(dropest_shard *sh, const char *n) : s(sh), name(n), t0(std::chrono::steady_clock::now()) {}
		~Phase() {
			if (s->trace) (void)stream_wait(s->ctx->stream);
			auto &st = s->phases[name];
			st.launches++;
			st.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		}
	};

	void step();
	void partition_and_exchange();
	void agree_on_key_fields();
	void global_umi_dictionary();
	// barcode merges across shards
	struct MergeWorld {   // the cells that take part: every shard's rows in rank order, identical on every shard
		struct LRow { u64 barcode; u32 n_genes, req_genes, req_umis, local_id; int32_t total_umis, total_reads; u32 first_read; };
		std::vector<LRow> Gm; std::vector<size_t> goff; u32 nG = 0, lo = 0, hi = 0;
	};
	struct TravelRows { dropest::DevBuf<u64> low_all; dropest::DevBuf<u32> col_all[4], q_all; std::vector<uint64_t> beg, end; };
	struct MergeApplied { std::vector<u32> final_t, mrank; std::vector<uint8_t> excl; std::vector<int32_t> reads, umis; };
	void merge_gather_cells(MergeWorld &W);
	void merge_gather_rows(const MergeWorld &W, TravelRows &T);
	std::vector<u32> merge_order(const MergeWorld &W);
	void merge_apply(const MergeWorld &W, const std::vector<int64_t> &my_tgt, const std::vector<u32> &order, MergeApplied &A);
	void merge_finish(const MergeWorld &W, const MergeApplied &A, TravelRows &T);
	void merge_umi_distribution();
	void cb_merge();
	void cb_merge_free();                        // shard_merge_free.h: Simple / PoissonSimple / merge-all
	void free_simple_targets(const MergeWorld &W, const std::vector<u32> &order, const std::vector<u32> &pos_of, std::vector<int64_t> &my_tgt);
	std::vector<double> free_expected(const MergeWorld &W, const std::vector<u32> &pair_base, const std::vector<u32> &pair_other);
	std::unique_ptr<TravelRows> free_rows;      // PoissonSimple: molecule rows gathered for the estimator, reused when the merge is applied
	void build_global_table();
	void assemble_matrix(bool filtered_m);
	std::vector<u32> order_rows(const std::vector<u32> &sel, bool by_first);
	std::vector<u64> global_ordinals(const std::vector<u32> &local_pos);
	dropest::OrdinalMap ordinal_map() const;
	void install_umi_hooks();
};

dropest::OrdinalMap dropest_shard::ordinal_map() const {
	using namespace dropest;
	OrdinalMap m{};
	if (exchanged && !idx_exchanged) throw InvalidError("internal: device ordinals asked for, but the read positions did not travel with this exchange");
	m.idx = exchanged ? x_idx.p : nullptr;
	m.world = exchanged ? u32(world) : 1u;
	if (exchanged) { for (int p = 0; p <= world; ++p) m.recv_off[p] = recv_off[size_t(p)]; for (int p = 0; p < world; ++p) m.first_ord[p] = first_ord[size_t(p)]; }
	else { m.recv_off[0] = 0; m.recv_off[1] = n_res; m.first_ord[0] = first_ordinal; }
	return m;
}

std::vector<dropest::u64> dropest_shard::global_ordinals(const std::vector<u32> &local_pos) {
	using namespace dropest;
	std::vector<u64> out(local_pos.size());
	dropest_ctx &c = *ctx;
	if (exchanged && !idx_exchanged) {
		// COLLECTIVE.  The positions did not travel: a read at local position p came from source s = the block [recv_off[s],
		// recv_off[s + 1]) it lies in, as the (p - recv_off[s])-th read s sent here -- s still holds the index array of its stable
		// partition and answers.  A pass asks for a few thousand positions (first reads of the real cells, tie candidates of the
		// N-UMI merge), against 4 bytes for every read of the stream if the array travelled.
		struct Query { u32 src, dst, q, tag; };
		struct Answer { u32 dst, tag, idx, pad; };
		std::vector<Query> mine, all;
		for (size_t i = 0; i < local_pos.size(); ++i) {
			const u32 p = local_pos[i];
			if (p == 0xFFFFFFFFu) { out[i] = ~0ull; continue; }
			u32 src = 0;
			while (src + 1 < u32(world) && p >= recv_off[size_t(src) + 1]) ++src;
			mine.push_back(Query{src, u32(rank), u32(p - recv_off[src]), u32(i)});
		}
		std::vector<size_t> cnt;
		tr->gather_vec(mine, all, cnt);
		std::vector<u32> pos; std::vector<Answer> my_ans, all_ans;
		for (const Query &q : all) if (int(q.src) == rank) { pos.push_back(u32(send_off[q.dst] + q.q)); my_ans.push_back(Answer{q.dst, q.tag, 0, 0}); }
		if (!pos.empty()) {
			const u32 n = u32(pos.size());
			DevBuf<u32> &d_pos = d_ord_pos, &d_out = d_ord_out;   // kept: a hipFree synchronises the whole device -- cm_raw's columns are on their way on another stream
			d_pos.ensure(n); d_out.ensure(n);
			HIP_CHECK(hipMemcpyAsync(d_pos.p, pos.data(), size_t(n) * 4, hipMemcpyHostToDevice, c.stream));
			hipLaunchKernelGGL(take_u32_kernel, dim3((n + 255) / 256), dim3(256), 0, c.stream, p_idx.p, d_pos.p, n, d_out.p);
			HIP_CHECK(hipGetLastError());
			std::vector<u32> got(n);
			c.fetch(got.data(), d_out.p, size_t(n) * 4);
			for (u32 k = 0; k < n; ++k) my_ans[k].idx = got[k];
		}
		tr->gather_vec(my_ans, all_ans, cnt);
		// the answers of source s come back in the order its queries stood in the gathered list: rank blocks in rank order
		size_t at = 0;
		for (int s = 0; s < world; ++s)
			for (size_t k = 0; k < cnt[size_t(s)]; ++k, ++at)
				if (int(all_ans[at].dst) == rank) out[all_ans[at].tag] = first_ord[size_t(s)] + all_ans[at].idx;
		return out;
	}
	if (local_pos.empty()) return out;
	const OrdinalMap m = ordinal_map();
	const u32 n = u32(local_pos.size());
	DevBuf<u32> &d_pos = d_ord_pos; DevBuf<u64> &d_out = d_ord_out64;
	d_pos.ensure(n); d_out.ensure(n);
	HIP_CHECK(hipMemcpyAsync(d_pos.p, local_pos.data(), size_t(n) * 4, hipMemcpyHostToDevice, c.stream));
	hipLaunchKernelGGL(ordinals_kernel, dim3((n + 255) / 256), dim3(256), 0, c.stream, m, d_pos.p, n, d_out.p);
	HIP_CHECK(hipGetLastError());
	c.fetch(out.data(), d_out.p, size_t(n) * 8);
	return out;
}

// 1-2. partition by owner (stable: reads of one owner keep stream order), one all-to-all(v).  The histogram pass also measures
// the field widths of the resident reads; the shards agree on them (the same collective that exchanges the block sizes) and, when
// barcode + UMI fit 64 bits and gene + mark + chromosome fit 32, every read crosses the links as 12 bytes instead of 28.
void dropest_shard::partition_and_exchange() {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	const u32 n = u32(n_res);
	send_cnt.assign(size_t(world), 0);
	// -u turns a whole table of positions into ordinals on the device; so do the tie replays of the whitelist-free merges (global cell
	// indices, global UMI order: shard_merge_free.h)
	const bool want_idx = c.cfg.umi_merge_kind == DROPEST_UMI_MERGE_DIRECTIONAL || c.cfg.merge_kind == DROPEST_MERGE_SIMPLE || c.cfg.merge_kind == DROPEST_MERGE_POISSON_SIMPLE;
	std::vector<uint64_t> all_cnt(size_t(world) * size_t(world));
	ExchangePack pack{};
	// The field widths of the packed record come from a SAMPLE of the reads (every 64th row of the histogram pass: that pass then reads
	// the barcodes only, 8 bytes per read instead of 24); the scatter, which reads every field anyway, reports a read that does not fit.
	// Then -- on every shard alike -- the partition is repeated with exact widths, and this shard object stays with those.
	const int C_plan = exchange_chunks == 1 ? 1 : (exchange_chunks ? exchange_chunks : 4);   // (the same option on every shard, like the others)
	std::vector<u32> chunk_bounds(size_t(C_plan) + 1, 0), my_chunk_cnt(size_t(world) * size_t(C_plan), 0);
	std::vector<uint64_t> chunk_cnt_all;   // [source][destination][chunk]
	chunked_now = false;
	c.recv_chunks.clear();
	auto partition = [&](bool sampled) -> bool {
		Phase ph(this, "partition");
		uint64_t need = 0;
		if (dropest_partition_scratch_bytes(n, &need) != DROPEST_OK) throw UnsupportedError("more than 2^32-2 reads per GPU");
		part_scratch.ensure(size_t(need) + 64);
		u32 nblocks, tpb; size_t off_k1, off_hist, off_row, off_base, total;
		partition_plan(n, nblocks, tpb, off_k1, off_hist, off_row, off_base, total);
		char *base = reinterpret_cast<char *>(part_scratch.p);
		u32 *hist = reinterpret_cast<u32 *>(base + off_hist), *row_total = reinterpret_cast<u32 *>(base + off_row), *digit_base = reinterpret_cast<u32 *>(base + off_base);
		u64 *d_stats = reinterpret_cast<u64 *>(base + ((total + 7) & ~size_t(7)));
		uint64_t stats[5] = {0, 0, 0, 0, 0};
		(void)sampled;
		std::vector<u32> totals(RS_RADIX, 0);
		if (n) {
			HIP_CHECK(hipMemsetAsync(d_stats, 0, 48, c.stream));
			if (sampled) hipLaunchKernelGGL(owner_hist_stats_kernel<64>, dim3(nblocks), dim3(OP_T), 0, c.stream, r_cb, r_umi, r_gene, r_aux, n, u32(world), tpb, hist, d_stats);
			else hipLaunchKernelGGL(owner_hist_stats_kernel<1>, dim3(nblocks), dim3(OP_T), 0, c.stream, r_cb, r_umi, r_gene, r_aux, n, u32(world), tpb, hist, d_stats);
			hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(RS_RADIX), dim3(256), 0, c.stream, hist, nblocks, row_total);
			hipLaunchKernelGGL(rs_scan_totals_kernel<256>, dim3(1), dim3(256), 0, c.stream, row_total, digit_base);
			HIP_CHECK(hipGetLastError());
			HIP_CHECK(hipMemcpyAsync(totals.data(), row_total, RS_RADIX * 4, hipMemcpyDeviceToHost, c.stream));
			if (C_plan > 1) {   // reads of every owner inside every chunk of blocks (the all-to-all may leave chunk by chunk)
				for (int k = 0; k <= C_plan; ++k) chunk_bounds[size_t(k)] = u32(uint64_t(nblocks) * uint64_t(k) / uint64_t(C_plan));
				d_chunk_bounds.ensure(size_t(C_plan) + 1); d_chunk_cnt.ensure(size_t(world) * size_t(C_plan));
				HIP_CHECK(hipMemcpyAsync(d_chunk_bounds.p, chunk_bounds.data(), (size_t(C_plan) + 1) * 4, hipMemcpyHostToDevice, c.stream));
				hipLaunchKernelGGL(owner_chunk_counts_kernel, dim3(div_up(u32(world) * u32(C_plan), 256u)), dim3(256), 0, c.stream, hist, row_total, nblocks, u32(world), u32(C_plan),
				                   d_chunk_bounds.p, d_chunk_cnt.p);
				HIP_CHECK(hipGetLastError());
				HIP_CHECK(hipMemcpyAsync(my_chunk_cnt.data(), d_chunk_cnt.p, size_t(world) * size_t(C_plan) * 4, hipMemcpyDeviceToHost, c.stream));
			}
			c.fetch(stats, d_stats, 40);
		}
		for (int p = 0; p < world; ++p) send_cnt[size_t(p)] = totals[size_t(p)];
		// block sizes and field widths of everybody, in one collective
		// (the sixth word: did this shard measure its widths on a sample?  A shard that took the exact path -- its exact_widths option, an
		// earlier pass that did not fit -- must still lay the record out like the others and join their 'did everything fit' collective:
		// the layout below follows what ANY shard did, never this shard's own option alone -- ADVICE r4)
		const size_t WC = C_plan > 1 ? size_t(world) * size_t(C_plan) : 0, W = size_t(world) + 6 + WC;
		std::vector<uint64_t> mine(W), every(W * size_t(world));
		for (int p = 0; p < world; ++p) mine[size_t(p)] = send_cnt[size_t(p)];
		for (int k = 0; k < 5; ++k) mine[size_t(world) + size_t(k)] = stats[k];
		mine[size_t(world) + 5] = sampled ? 1u : 0u;
		for (size_t i = 0; i < WC; ++i) mine[size_t(world) + 6 + i] = my_chunk_cnt[i];
		tr->gather_host(mine.data(), mine.size() * 8, every.data());
		chunk_cnt_all.assign(WC * size_t(world), 0);
		for (int p = 0; p < world; ++p) for (size_t i = 0; i < WC; ++i) chunk_cnt_all[size_t(p) * WC + i] = every[size_t(p) * W + size_t(world) + 6 + i];
		uint64_t g[5] = {0, 0, 0, 0, 0};
		bool any_sampled = false;
		for (int p = 0; p < world; ++p) {
			for (int q = 0; q < world; ++q) all_cnt[size_t(p) * size_t(world) + size_t(q)] = every[size_t(p) * W + size_t(q)];
			for (int k = 0; k < 4; ++k) g[k] = std::max(g[k], every[size_t(p) * W + size_t(world) + size_t(k)]);
			g[4] |= every[size_t(p) * W + size_t(world) + 4];
			any_sampled |= every[size_t(p) * W + size_t(world) + 5] != 0;
		}
		const int cb_bits = std::max(1, bit_length(g[0])), umi_bits = std::max(1, bit_length(g[1]));
		const int gene_bits = std::max(1, bit_length(g[2])), chr_bits = std::max(1, bit_length(g[3]));
		packed = allow_packed && !g[4] && cb_bits + umi_bits <= 64 && gene_bits + 3 + chr_bits <= 32;   // (a code with N has bit 63 set: never packed)
		idx_exchanged = want_idx || !packed;
		{
			uint64_t everybody = 0;
			for (uint64_t x : all_cnt) everybody += x;
			chunked_now = C_plan > 1 && packed && !idx_exchanged && !r_have_qual && (exchange_chunks > 1 || everybody >= (uint64_t(1) << 22) * uint64_t(world));
		}
		pack.cb_bits = cb_bits;
		// sampled widths: the gene field takes every bit the chromosome does not need (a wider field costs nothing), the chromosome one bit of slack
		pack.gene_bits = any_sampled && packed ? std::max(gene_bits, 32 - 3 - std::min(chr_bits + 1, 32 - 3 - gene_bits)) : gene_bits;
		rec_bytes = packed ? (idx_exchanged ? 16 : 12) : 28;
		send_off.assign(size_t(world) + 1, 0);
		for (int p = 0; p < world; ++p) send_off[size_t(p) + 1] = send_off[size_t(p)] + send_cnt[size_t(p)];
		// what this shard will receive is known from the same collective: the receive arrays exist before the scatter, which writes the
		// block the shard keeps straight into them
		recv_cnt.assign(size_t(world), 0); recv_off.assign(size_t(world) + 1, 0);
		for (int p = 0; p < world; ++p) { recv_cnt[size_t(p)] = all_cnt[size_t(p) * size_t(world) + size_t(rank)]; recv_off[size_t(p) + 1] = recv_off[size_t(p)] + recv_cnt[size_t(p)]; }
		const uint64_t n_recv0 = recv_off[size_t(world)];
		if (n_recv0 >= 0xFFFFFFFEull) throw UnsupportedError("more than 2^32-2 reads on one shard after the exchange");
		p_idx.ensure(std::max<size_t>(n, 1));
		if (packed) { p_w0.ensure(std::max<size_t>(n, 1)); p_w1.ensure(std::max<size_t>(n, 1)); x_w0.ensure(std::max<size_t>(n_recv0, 1)); x_w1.ensure(std::max<size_t>(n_recv0, 1)); }
		else {
			for (DevBuf<u64> *b : {&p_cb, &p_umi}) b->ensure(std::max<size_t>(n, 1));
			for (DevBuf<u32> *b : {&p_gene, &p_aux}) b->ensure(std::max<size_t>(n, 1));
		}
		if (n) {
			const int owner_bits = std::max(1, bit_length(uint64_t(world - 1)));
			OwnerSelf self{};
			if (any_sampled && packed) { HIP_CHECK(hipMemsetAsync(d_stats + 5, 0, 8, c.stream)); self.bad = reinterpret_cast<u32 *>(d_stats + 5); }
			self.owner = u32(rank);
			self.w0 = x_w0.p + recv_off[size_t(rank)] - send_off[size_t(rank)];     // (index = position in the partition's output)
			self.w1 = x_w1.p + recv_off[size_t(rank)] - send_off[size_t(rank)];
			if (packed && chunked_now) {
				// (launched chunk by chunk below, each chunk's pieces leaving as soon as they are written)
			} else if (packed) hipLaunchKernelGGL(owner_scatter_kernel<true>, dim3(nblocks), dim3(OP_T), 0, c.stream, r_cb, r_umi, r_gene, r_aux, n, u32(world), owner_bits, tpb, hist, digit_base,
			                               p_w0.p, static_cast<u64 *>(nullptr), p_w1.p, static_cast<u32 *>(nullptr), p_idx.p, pack, self);
			else hipLaunchKernelGGL(owner_scatter_kernel<false>, dim3(nblocks), dim3(OP_T), 0, c.stream, r_cb, r_umi, r_gene, r_aux, n, u32(world), owner_bits, tpb, hist, digit_base,
			                        p_cb.p, p_umi.p, p_gene.p, p_aux.p, p_idx.p, pack);
			HIP_CHECK(hipGetLastError());
		}
		if (chunked_now) {
			// chunk k: partition its blocks, then send its pieces (exchange stream) while chunk k + 1 is partitioned; ev_recv[k] = chunk k of every
			// source's block has landed
			if (!xchg_stream) HIP_CHECK(hipStreamCreateWithFlags(&xchg_stream, hipStreamNonBlocking));
			while (ev_scat.size() < size_t(C_plan)) { hipEvent_t e; HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev_scat.push_back(e); }
			while (ev_recv.size() < size_t(C_plan)) { hipEvent_t e; HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev_recv.push_back(e); }
			const size_t WC = size_t(world) * size_t(C_plan);
			const int owner_bits = std::max(1, bit_length(uint64_t(world - 1)));
			OwnerSelf self{};
			if (any_sampled) self.bad = reinterpret_cast<u32 *>(d_stats + 5);   // (cleared above)
			self.owner = u32(rank);
			self.w0 = x_w0.p + recv_off[size_t(rank)] - send_off[size_t(rank)];
			self.w1 = x_w1.p + recv_off[size_t(rank)] - send_off[size_t(rank)];
			std::vector<uint64_t> s_off(static_cast<size_t>(world)), s_cnt(static_cast<size_t>(world)), r_off(static_cast<size_t>(world)), r_cnt(static_cast<size_t>(world));
			std::vector<uint64_t> s_run(send_off.begin(), send_off.begin() + world), r_run(recv_off.begin(), recv_off.begin() + world);
			c.recv_chunks.clear();
			for (int k = 0; k < C_plan; ++k) {
				const u32 b0 = chunk_bounds[size_t(k)], b1 = chunk_bounds[size_t(k) + 1];
				if (n && b1 > b0) {
					hipLaunchKernelGGL(owner_scatter_kernel<true>, dim3(b1 - b0), dim3(OP_T), 0, c.stream, r_cb, r_umi, r_gene, r_aux, n, u32(world), owner_bits, tpb, hist, digit_base,
					                   p_w0.p, static_cast<u64 *>(nullptr), p_w1.p, static_cast<u32 *>(nullptr), p_idx.p, pack, self, b0, nblocks);
					HIP_CHECK(hipGetLastError());
				}
				HIP_CHECK(hipEventRecord(ev_scat[size_t(k)], c.stream));
				HIP_CHECK(hipStreamWaitEvent(xchg_stream, ev_scat[size_t(k)], 0));
				dropest_ctx::RecvChunk rc{};
				rc.ev = ev_recv[size_t(k)];
				rc.rg.n = u32(world);
				for (int p = 0; p < world; ++p) {
					s_off[size_t(p)] = s_run[size_t(p)]; s_cnt[size_t(p)] = chunk_cnt_all[size_t(rank) * WC + size_t(p) * size_t(C_plan) + size_t(k)];
					r_off[size_t(p)] = r_run[size_t(p)]; r_cnt[size_t(p)] = chunk_cnt_all[size_t(p) * WC + size_t(rank) * size_t(C_plan) + size_t(k)];
					s_run[size_t(p)] += s_cnt[size_t(p)]; r_run[size_t(p)] += r_cnt[size_t(p)];
					rc.rg.off[p] = u32(r_off[size_t(p)]); rc.rg.cnt[p] = u32(r_cnt[size_t(p)]);
				}
				const void *snd[2] = {p_w0.p, p_w1.p};
				void *rcv[2] = {x_w0.p, x_w1.p};
				const size_t elem[2] = {8, 4};
				tr->exchange_at(2, snd, rcv, elem, s_off.data(), s_cnt.data(), r_off.data(), r_cnt.data(), xchg_stream, 3u);   // the kept pieces: already in place
				HIP_CHECK(hipEventRecord(ev_recv[size_t(k)], xchg_stream));
				c.recv_chunks.push_back(rc);
			}
			for (int p = 0; p < world; ++p)
				if (s_run[size_t(p)] != send_off[size_t(p)] + send_cnt[size_t(p)] || r_run[size_t(p)] != recv_off[size_t(p)] + recv_cnt[size_t(p)])
					throw InvalidError("internal: the chunks of the exchange do not add up to its blocks");
		}
		if (!(any_sampled && packed)) return true;
		// did every read of every shard fit?  (one word from the device, one small collective)
		uint64_t bad = 0;
		if (n) { u32 b32 = 0; c.fetch(&b32, d_stats + 5, 4); bad = b32; }
		std::vector<uint64_t> bad_of(static_cast<size_t>(world));
		tr->gather_host(&bad, 8, bad_of.data());
		for (uint64_t x : bad_of) if (x) { if (chunked_now) { HIP_CHECK(stream_wait(xchg_stream)); c.recv_chunks.clear(); } return false; }
		return true;
	};
	if (!partition(!exact_widths)) { exact_widths = true; phases["partition:exact_again"].launches++; partition(false); }
	{
		Phase ph(this, "all_to_all");
		const uint64_t n_recv = recv_off[size_t(world)];
		if (!packed) {
			for (DevBuf<u64> *b : {&x_cb, &x_umi}) b->ensure(std::max<size_t>(n_recv, 1));
			for (DevBuf<u32> *b : {&x_gene, &x_aux}) b->ensure(std::max<size_t>(n_recv, 1));
		}
		if (idx_exchanged) x_idx.ensure(std::max<size_t>(n_recv, 1));
		if (packed && chunked_now) {
			phases["all_to_all:chunks"].launches += u32(C_plan);   // (it left chunk by chunk, inside the partition)
		} else if (packed) {
			const void *snd[3] = {p_w0.p, p_w1.p, p_idx.p};
			void *rcv[3] = {x_w0.p, x_w1.p, x_idx.p};
			const size_t elem[3] = {8, 4, 4};
			tr->exchange(idx_exchanged ? 3 : 2, snd, rcv, elem, send_cnt.data(), recv_cnt.data(), c.stream, 3u);   // w0 / w1 of the own block: already in place
		} else {
			const void *snd[5] = {p_cb.p, p_umi.p, p_gene.p, p_aux.p, p_idx.p};
			void *rcv[5] = {x_cb.p, x_umi.p, x_gene.p, x_aux.p, x_idx.p};
			const size_t elem[5] = {8, 8, 4, 4, 4};
			tr->exchange(5, snd, rcv, elem, send_cnt.data(), recv_cnt.data(), c.stream);
		}
		auto &st = phases["all_to_all"];
		uint64_t out = 0;
		for (int p = 0; p < world; ++p) if (p != rank) out += send_cnt[size_t(p)];
		st.bytes += double(out) * rec_bytes;   // bytes this shard put on the links (self block excluded)
		phases["exchange_record_bytes"].bytes = double(rec_bytes);
		if (r_have_qual && r_qlen) {   // the quality strings: gathered into the order of the partition, exchanged with the same counts
			p_qual.ensure(std::max<size_t>(size_t(n) * r_qlen, 1)); x_qual.ensure(std::max<size_t>(size_t(n_recv) * r_qlen, 1));
			if (n) {
				hipLaunchKernelGGL(gather_byte_rows_kernel, dim3(u32(std::min<uint64_t>((uint64_t(n) * r_qlen + 255) / 256, 16384))), dim3(256), 0, c.stream,
				                   r_qual.p, p_idx.p, n, r_qlen, p_qual.p);
				HIP_CHECK(hipGetLastError());
			}
			const void *snd[1] = {p_qual.p};
			void *rcv[1] = {x_qual.p};
			const size_t elem[1] = {r_qlen};
			tr->exchange(1, snd, rcv, elem, send_cnt.data(), recv_cnt.data(), c.stream);
			uint64_t out_q = 0;
			for (int p = 0; p < world; ++p) if (p != rank) out_q += send_cnt[size_t(p)];
			st.bytes += double(out_q) * r_qlen;
			if (r_qual_var) {   // strings of several lengths: the length of every read's string travels the same way (1 byte per read)
				p_qual_lens.ensure(std::max<size_t>(n, 1)); x_qual_lens.ensure(std::max<size_t>(n_recv, 1));
				if (n) {
					hipLaunchKernelGGL(gather_byte_rows_kernel, dim3(u32(std::min<uint64_t>((uint64_t(n) + 255) / 256, 16384))), dim3(256), 0, c.stream,
					                   r_qual_lens.p, p_idx.p, n, 1u, p_qual_lens.p);
					HIP_CHECK(hipGetLastError());
				}
				const void *snd1[1] = {p_qual_lens.p};
				void *rcv1[1] = {x_qual_lens.p};
				const size_t elem1[1] = {1};
				tr->exchange(1, snd1, rcv1, elem1, send_cnt.data(), recv_cnt.data(), c.stream);
				st.bytes += double(out_q);
			}
		}
		// The records stay packed: cb_sample, cb_insert, the sampled statistics and build_keys read them as they are (k_cbhash.h: ReadPack);
		// whatever else needs the four columns (UMI first-occurrence tables, quality sums) asks the context, which calls back here.
		// DROPEST_SHARD_UNPACK_FIRST=1: the columns at once, as before.
		exch_pack = pack;
		if (packed && n_recv && getenv("DROPEST_SHARD_UNPACK_FIRST")) unpack_exchanged();
	}
	exchanged = true;
}

// the four columns of the received reads from their packed records (x_w0 / x_w1 hold every block, the kept one included)
void dropest_shard::unpack_exchanged() {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	const uint64_t n_recv = recv_off[size_t(world)];
	if (!packed || !n_recv) return;
	Phase ph2(this, "unpack");
	for (DevBuf<u64> *b : {&x_cb, &x_umi}) b->ensure(std::max<size_t>(n_recv, 1));
	for (DevBuf<u32> *b : {&x_gene, &x_aux}) b->ensure(std::max<size_t>(n_recv, 1));
	hipLaunchKernelGGL(exchange_unpack_kernel, dim3(u32(std::min<uint64_t>((n_recv + 255) / 256, 8192))), dim3(256), 0, c.stream, x_w0.p, x_w1.p, u32(n_recv), exch_pack,
	                   x_cb.p, x_umi.p, x_gene.p, x_aux.p, 0u, 0u, static_cast<const unsigned long long *>(nullptr), static_cast<const uint32_t *>(nullptr));
	HIP_CHECK(hipGetLastError());
	unpacked = true;
}

// 3b. all shards must lay out the gene / UMI fields of the sort key identically (molecule rows move between shards in a
// merge) and agree on whether a gene determines its chromosome
void dropest_shard::agree_on_key_fields() {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	uint64_t mine[6] = {c.ingest.umi_clean_min, c.ingest.umi_clean_max, c.ingest.umi_escape_max_plus1, c.ingest.gene_max_plus1,
	                    c.ingest.chr_max_plus1, c.ingest.gene_chr_conflict};
	std::vector<uint64_t> all(size_t(world) * 6);
	tr->gather_host(mine, sizeof(mine), all.data());
	uint64_t g[6] = {~0ull, 0, 0, 0, 0, 0};
	for (int p = 0; p < world; ++p) {
		g[0] = std::min(g[0], all[size_t(p) * 6]);
		for (int k = 1; k < 6; ++k) g[k] = std::max(g[k], all[size_t(p) * 6 + size_t(k)]);
	}
	const u32 n_genes = u32(std::min<uint64_t>(g[3], GENE_CHR_CAP));
	if (n_genes && !g[5]) {
		if (!c.gene_chr.p) {   // a shard without reads: an all-unset table
			c.gene_chr.ensure(GENE_CHR_CAP);
			HIP_CHECK(hipMemsetAsync(c.gene_chr.p, 0xFF, size_t(GENE_CHR_CAP) * 4, c.stream));
		}
		std::vector<u32> tab(n_genes), tabs(size_t(n_genes) * size_t(world));
		c.fetch(tab.data(), c.gene_chr.p, size_t(n_genes) * 4);
		tr->gather_host(tab.data(), size_t(n_genes) * 4, tabs.data());
		bool conflict = false;
		for (u32 i = 0; i < n_genes; ++i) {
			u32 v = GENE_CHR_UNSET;
			for (int p = 0; p < world; ++p) {
				const u32 x = tabs[size_t(p) * n_genes + i];
				if (x == GENE_CHR_UNSET) continue;
				if (v == GENE_CHR_UNSET) v = x; else if (v != x) conflict = true;   // one gene on two chromosomes, seen by different shards
			}
			tab[i] = v;
		}
		if (conflict) g[5] = 1;
		else {
			HIP_CHECK(hipMemcpyAsync(c.gene_chr.p, tab.data(), size_t(n_genes) * 4, hipMemcpyHostToDevice, c.stream));
			HIP_CHECK(stream_wait(c.stream));
		}
	}
	if (g[3] > GENE_CHR_CAP) g[5] = 1;
	c.ingest.umi_clean_min = g[0]; c.ingest.umi_clean_max = g[1]; c.ingest.umi_escape_max_plus1 = g[2];
	c.ingest.gene_max_plus1 = u32(g[3]); c.ingest.chr_max_plus1 = u32(g[4]); c.ingest.gene_chr_conflict = u32(g[5]);
}

// The UMI dictionary of a sharded run (round 6; StringIndexer::add has no width limit, Estimation/StringIndexer.cpp:10-18, Gene.cpp:17-24): when the
// gene and UMI fields alone reach 64 bits the key carries the UMI's rank among the distinct clean UMIs of the WHOLE stream.  Every shard
// makes the sorted distinct UMIs of its own reads (k_umidict.h), all of them gather every shard's list (Transport::gather_dev: RCCL all-gather
// across GPUs, peer copies inside a process, the shm plane across processes on one GPU), and each sorts the union and keeps the distinct
// values: one dictionary, the same on every shard, ranks ascending with the codes -- so every order that hangs on the UMI field is the plain
// layout's, and the molecule rows that cross between shards in a barcode merge carry ranks every shard reads alike.  (An all-to-all by
// mix64(UMI) mod n with per-owner ranks would move 1 / n of the bytes but give ranks that do NOT ascend with the codes.)  The decision comes
// from the agreed ingest statistics, so every shard takes it alike.
void dropest_shard::global_umi_dictionary() {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	int gene_bits = bit_length(uint64_t(c.ingest.gene_max_plus1));
	if ((1ull << gene_bits) - 1 < c.ingest.gene_max_plus1) gene_bits++;
	if (gene_bits == 0) gene_bits = 1;
	uint64_t cells_mine = c.n_cells;
	std::vector<uint64_t> cells_all(static_cast<size_t>(world));
	tr->gather_host(&cells_mine, 8, cells_all.data());
	uint64_t cells_max = 0;
	for (uint64_t x : cells_all) cells_max = std::max(cells_max, x);
	const int cell_bits = std::max(1, bit_length(cells_max ? cells_max - 1 : 0));
	if (!c.umi_dict_wanted(gene_bits, cell_bits)) return;
	Phase ph(this, "umi_dictionary");
	const std::function<void(DevBuf<u64> &, u32 &)> across = [&](DevBuf<u64> &dict, u32 &n) {
		uint64_t mine = n;
		std::vector<uint64_t> counts(static_cast<size_t>(world));
		tr->gather_host(&mine, 8, counts.data());
		std::vector<size_t> off(static_cast<size_t>(world)), bytes(static_cast<size_t>(world));
		uint64_t total = 0;
		for (int p = 0; p < world; ++p) { off[size_t(p)] = size_t(total) * 8; bytes[size_t(p)] = size_t(counts[size_t(p)]) * 8; total += counts[size_t(p)]; }
		if (total >= 0xFFFFFFF0ull) throw UnsupportedError("the shards' UMI dictionaries hold more than 2^32 entries together");
		DevBuf<u64> all;
		all.alloc(std::max<uint64_t>(total, 1));
		tr->gather_dev(dict.p, all.p, off.data(), bytes.data(), c.stream);
		u32 m = u32(total);
		c.sort_unique_u64(all, m);
		dict = std::move(all);
		n = m;
		phases["umi_dictionary"].bytes += double(total) * 8;
	};
	c.build_umi_dict(&across);
}

// Order of table rows `sel` (indices into G): by_first = ascending global first ordinal (cell-id order of ONE container);
// else CellsDataContainer::compare_cells (CellsDataContainer.cpp:329-344): (requested_genes, requested_umis, umis_number,
// barcode string) ascending.  Large tables of clean equal-length barcodes are sorted on the device (three stable radix
// sorts, as dropest_ctx::sort_filtered does); anything else on the host with the strings decoded.
std::vector<dropest::u32> dropest_shard::order_rows(const std::vector<u32> &sel, bool by_first) {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	const size_t m = sel.size();
	std::vector<u32> out(sel);
	if (m < 2) return out;
	size_t device_min = 50000;
	if (const char *e = getenv("DROPEST_DEVICE_SORT_MIN")) device_min = size_t(std::max(1, atoi(e)));
	if (by_first) {
		// every shard orders the real cells of ALL shards: at 8 GPUs that is ~10^6 rows -- one device radix sort, not a host sort
		if (m >= device_min && m < 0xFFFFFFF0ull) {
			const u32 mm = u32(m);
			c.sort_stage.ensure(size_t(mm)); c.keys_a.ensure(mm); c.keys_b.ensure(mm); c.vals_a.ensure(mm); c.vals_b.ensure(mm);
			u64 o = 0, a = ~0ull;
			for (u32 k = 0; k < mm; ++k) { const u64 f = G[sel[k]].first_global; c.sort_stage.p[k] = f; o |= f; a &= f; }
			u64 *k = c.keys_a.p, *k_alt = c.keys_b.p;
			u32 *v = c.vals_a.p, *v_alt = c.vals_b.p;
			HIP_CHECK(hipMemcpyAsync(k, c.sort_stage.p, size_t(mm) * 8, hipMemcpyHostToDevice, c.stream));
			hipLaunchKernelGGL(iota_kernel, dim3(div_up(mm, 256)), dim3(256), 0, c.stream, v, mm);
			c.radix_sort(k, v, k_alt, v_alt, mm, o ^ a);
			u32 *perm = reinterpret_cast<u32 *>(c.sort_stage.p);
			HIP_CHECK(hipMemcpyAsync(perm, v, size_t(mm) * 4, hipMemcpyDeviceToHost, c.stream));
			HIP_CHECK(stream_wait(c.stream));
			for (u32 kk = 0; kk < mm; ++kk) out[kk] = sel[perm[kk]];
			return out;
		}
		std::sort(out.begin(), out.end(), [&](u32 a, u32 b) { return G[a].first_global < G[b].first_global; });
		return out;
	}
	bool uniform = true; u64 any = 0;
	const int bl0 = bit_length(G[sel[0]].barcode);
	for (u32 i : sel) { any |= G[i].barcode; uniform &= bit_length(G[i].barcode) == bl0; }
	if (m >= device_min && m < 0xFFFFFFF0ull && uniform && !(any & ESCAPE_BIT)) {
		const u32 mm = u32(m);
		c.sort_stage.ensure(size_t(mm) * 3); c.sort_cols.ensure(size_t(mm) * 3);
		c.keys_a.ensure(mm); c.keys_b.ensure(mm); c.vals_a.ensure(mm); c.vals_b.ensure(mm);
		u64 *h_code = c.sort_stage.p, *h_umis = c.sort_stage.p + mm, *h_sizes = c.sort_stage.p + 2 * size_t(mm);
		u64 o[3] = {0, 0, 0}, a[3] = {~0ull, ~0ull, ~0ull};
		for (u32 k = 0; k < mm; ++k) {
			const GRow &r = G[sel[k]];
			h_code[k] = r.barcode; h_umis[k] = u64(size_t(r.total_umis)); h_sizes[k] = (u64(r.req_genes) << 32) | r.req_umis;
			o[0] |= h_code[k]; a[0] &= h_code[k]; o[1] |= h_umis[k]; a[1] &= h_umis[k]; o[2] |= h_sizes[k]; a[2] &= h_sizes[k];
		}
		HIP_CHECK(hipMemcpyAsync(c.sort_cols.p, c.sort_stage.p, size_t(mm) * 3 * 8, hipMemcpyHostToDevice, c.stream));
		const u64 *d_code = c.sort_cols.p, *d_umis = c.sort_cols.p + mm, *d_sizes = c.sort_cols.p + 2 * size_t(mm);
		u64 *k = c.keys_a.p, *k_alt = c.keys_b.p;
		u32 *v = c.vals_a.p, *v_alt = c.vals_b.p;
		hipLaunchKernelGGL(iota_kernel, dim3(div_up(mm, 256)), dim3(256), 0, c.stream, v, mm);
		HIP_CHECK(hipMemcpyAsync(k, d_code, size_t(mm) * 8, hipMemcpyDeviceToDevice, c.stream));
		c.radix_sort(k, v, k_alt, v_alt, mm, o[0] ^ a[0]);
		const std::pair<const u64 *, u64> more[2] = {{d_umis, o[1] ^ a[1]}, {d_sizes, o[2] ^ a[2]}};
		for (auto const &nx : more) {
			hipLaunchKernelGGL(gather_u64_kernel, dim3(div_up(mm, 256)), dim3(256), 0, c.stream, nx.first, v, mm, k);
			HIP_CHECK(hipGetLastError());
			c.radix_sort(k, v, k_alt, v_alt, mm, nx.second);
		}
		u32 *perm = reinterpret_cast<u32 *>(c.sort_stage.p);
		HIP_CHECK(hipMemcpyAsync(perm, v, size_t(mm) * 4, hipMemcpyDeviceToHost, c.stream));
		HIP_CHECK(stream_wait(c.stream));
		for (u32 kk = 0; kk < mm; ++kk) out[kk] = sel[perm[kk]];
		return out;
	}
	// (requested genes, requested UMIs) decide almost every comparison: a stable LSD counting sort on their varying bytes, then only the
	// runs of equal sizes by the full comparison (as dropest_ctx::sort_filtered does)
	auto less = [&](u32 x, u32 y) { return compare_cells_rows(G[x], G[y], c.side); };
	if (m > 64) {
		auto sizes_of = [&](u32 i) { return (u64(G[i].req_genes) << 32) | G[i].req_umis; };
		u64 o = 0, a = ~0ull;
		for (u32 i : out) { o |= sizes_of(i); a &= sizes_of(i); }
		const u64 vary = o ^ a;
		std::vector<u32> other(m);
		for (int shift = 0; shift < 64; shift += 8) {
			if (!((vary >> shift) & 0xFFull)) continue;
			size_t cnt[257] = {0};
			for (u32 i : out) ++cnt[((sizes_of(i) >> shift) & 0xFFull) + 1];
			for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
			for (u32 i : out) other[cnt[(sizes_of(i) >> shift) & 0xFFull]++] = i;
			out.swap(other);
		}
		for (size_t i = 0; i < m;) {
			size_t j = i + 1;
			while (j < m && sizes_of(out[j]) == sizes_of(out[i])) ++j;
			if (j - i > 1) std::sort(out.begin() + long(i), out.begin() + long(j), less);
			i = j;
		}
	} else
		std::sort(out.begin(), out.end(), less);
	return out;
}

// 5. CB merge over all shards (MergeStrategyBase::merge_inited, MergeStrategyBase.cpp:11-57).  Every shard ends with the same global
// picture and applies the part that concerns its cells.  Common to every strategy: the cells that take part are all-gathered
// (merge_gather_cells), every shard decides the targets of ITS cells, the sequential smallest-first application runs identically
// everywhere (merge_apply), molecule rows of cells whose target lives elsewhere travel (merge_gather_rows) and are appended where the
// target lives (merge_finish).
void dropest_shard::merge_gather_cells(MergeWorld &W) {
	dropest_ctx &c = *ctx;
	std::vector<MergeWorld::LRow> local;
	for (const HostCell &h : c.real) {
		if (h.merged || h.excluded || h.row.n_genes < c.min_before) continue;
		local.push_back(MergeWorld::LRow{h.row.barcode, h.row.n_genes, h.row.requested_genes, h.row.requested_umis, h.id, h.row.total_umis, h.row.total_reads, h.row.first_read});
	}
	std::vector<size_t> cnt;
	{ Phase ph(this, "cbm:gather_cells"); tr->gather_vec(local, W.Gm, cnt); }
	if (W.Gm.size() >= 0x7FFFFFFFull) throw UnsupportedError("more than 2^31 cells in a sharded merge");
	W.nG = u32(W.Gm.size());
	W.goff.assign(size_t(world) + 1, 0);
	for (int p = 0; p < world; ++p) W.goff[size_t(p) + 1] = W.goff[size_t(p)] + cnt[size_t(p)];
	W.lo = u32(W.goff[size_t(rank)]); W.hi = u32(W.goff[size_t(rank) + 1]);
}

// the export buffers of every shard's ShardMerge (rows of its listed cells), all-gathered; beg / end = row range of a listed cell
void dropest_shard::merge_gather_rows(const MergeWorld &W, TravelRows &T) {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	dropest_ctx::ShardMerge &M = *c.shard;
	Phase ph(this, "cbm:gather_rows");
	struct Listed { u32 g; uint64_t b, e; };
	std::vector<Listed> my_listed(M.listed_g.size()), all_listed;
	for (size_t i = 0; i < M.listed_g.size(); ++i) my_listed[i] = Listed{M.listed_g[i], M.row_offset[i], M.row_offset[i + 1]};
	std::vector<size_t> lcnt;
	tr->gather_vec(my_listed, all_listed, lcnt);
	const bool with_qual = c.have_qual && c.qual_len;
	T.beg.assign(W.nG, ~0ull); T.end.assign(W.nG, ~0ull);
	uint64_t my_rows = M.row_offset.empty() ? 0 : M.row_offset.back();
	std::vector<uint64_t> rows(static_cast<size_t>(world));
	tr->gather_host(&my_rows, 8, rows.data());
	std::vector<size_t> off8(static_cast<size_t>(world)), b8(static_cast<size_t>(world)), off4(static_cast<size_t>(world)), b4(static_cast<size_t>(world));
	uint64_t total = 0;
	std::vector<uint64_t> row_base(size_t(world) + 1, 0);
	for (int p = 0; p < world; ++p) { row_base[size_t(p) + 1] = row_base[size_t(p)] + rows[size_t(p)]; total += rows[size_t(p)]; }
	if (total > 0xFFFFFFF0ull) throw UnsupportedError("more than 2^32 molecule rows travel in the sharded merge");
	for (int p = 0; p < world; ++p) { off8[size_t(p)] = size_t(row_base[size_t(p)]) * 8; b8[size_t(p)] = size_t(rows[size_t(p)]) * 8; off4[size_t(p)] = size_t(row_base[size_t(p)]) * 4; b4[size_t(p)] = size_t(rows[size_t(p)]) * 4; }
	T.low_all.alloc(std::max<size_t>(total, 1));
	for (auto &b : T.col_all) b.alloc(std::max<size_t>(total, 1));
	if (!M.x_low.p) { M.x_low.alloc(1); for (auto &b : M.x_col) b.alloc(1); }
	tr->gather_dev(M.x_low.p, T.low_all.p, off8.data(), b8.data(), c.stream);
	for (int k = 0; k < 4; ++k) tr->gather_dev(M.x_col[k].p, T.col_all[k].p, off4.data(), b4.data(), c.stream);
	if (with_qual) {   // the sums rows of the travelling molecules
		const size_t qs = c.qual_stride();
		std::vector<size_t> offq(static_cast<size_t>(world)), bq(static_cast<size_t>(world));
		for (int p = 0; p < world; ++p) { offq[size_t(p)] = off4[size_t(p)] * qs; bq[size_t(p)] = b4[size_t(p)] * qs; }
		T.q_all.alloc(std::max<size_t>(total * qs, 1));
		if (!M.x_q.p) M.x_q.alloc(1);
		tr->gather_dev(M.x_q.p, T.q_all.p, offq.data(), bq.data(), c.stream);
	}
	size_t at = 0;
	for (int p = 0; p < world; ++p)
		for (size_t i = 0; i < lcnt[size_t(p)]; ++i, ++at) { T.beg[all_listed[at].g] = all_listed[at].b + row_base[size_t(p)]; T.end[all_listed[at].g] = all_listed[at].e + row_base[size_t(p)]; }
}

// my_tgt[i] = target (global place, or -1) of cell lo + i; the same sequential application on every shard.  `order` = the cells in
// ascending compare_cells order (the reference's filtered order).
void dropest_shard::merge_apply(const MergeWorld &W, const std::vector<int64_t> &my_tgt, const std::vector<u32> &order, MergeApplied &A) {
	dropest_ctx &c = *ctx;
	const u32 nG = W.nG;
	std::vector<int64_t> target;
	{ Phase ph(this, "cbm:gather_targets"); std::vector<size_t> tcnt; tr->gather_vec(my_tgt, target, tcnt); }
	if (target.size() != nG) throw InvalidError("internal: the shards' merge targets do not cover the cell list");
	const bool with_qual = c.have_qual && c.qual_len;
	A.final_t.assign(nG, 0); A.excl.assign(nG, 0); A.reads.assign(nG, 0); A.umis.assign(nG, 0);
	A.mrank.assign(with_qual ? nG : 0u, 0);   // place of every cell in its target's merge order (0: not merged away)
	Phase ph(this, "cbm:order+apply");
	std::vector<int64_t> tgt_in_order(nG);
	for (u32 i = 0; i < nG; ++i) { tgt_in_order[i] = target[order[i]]; A.reads[i] = W.Gm[i].total_reads; A.umis[i] = W.Gm[i].total_umis; }
	apply_merge_order(nG, nG, order.data(), tgt_in_order.data(), A.reads.data(), A.umis.data(), A.final_t.data(), A.excl.data(), with_qual ? A.mrank.data() : nullptr);
}

std::vector<dropest::u32> dropest_shard::merge_order(const MergeWorld &W) {
	const u32 nG = W.nG;
	G.resize(nG);
	for (u32 i = 0; i < nG; ++i) { GRow r{}; r.barcode = W.Gm[i].barcode; r.req_genes = W.Gm[i].req_genes; r.req_umis = W.Gm[i].req_umis; r.total_umis = W.Gm[i].total_umis; G[i] = r; }
	std::vector<u32> sel(nG);
	for (u32 i = 0; i < nG; ++i) sel[i] = i;
	return order_rows(sel, false);
}

void dropest_shard::merge_finish(const MergeWorld &W, const MergeApplied &A, TravelRows &T) {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	const u32 nG = W.nG, lo = W.lo, hi = W.hi;
	const auto &Gm = W.Gm;
	const bool with_qual = c.have_qual && c.qual_len;
	Phase ph(this, "cbm:finish");
	std::vector<u32> local_id, move_src, move_tgt, import_row, import_cell, local_rank, import_rank;
	std::vector<uint8_t> l_excl, l_merged; std::vector<int32_t> l_reads, l_umis;
	merged_barcodes.clear();
	for (u32 i = 0; i < nG; ++i) {
		const bool mine = i >= lo && i < hi;
		if (mine) { local_id.push_back(Gm[i].local_id); l_excl.push_back(A.excl[i]); l_merged.push_back(A.final_t[i] != i); l_reads.push_back(A.reads[i]); l_umis.push_back(A.umis[i]);
		            if (with_qual) local_rank.push_back(A.mrank[i]); }
		if (A.final_t[i] == i) continue;
		merged_barcodes.emplace_back(Gm[i].barcode, Gm[A.final_t[i]].barcode);
		const u32 t = A.final_t[i];
		if (!(t >= lo && t < hi)) continue;
		if (mine) { move_src.push_back(Gm[i].local_id); move_tgt.push_back(Gm[t].local_id); continue; }
		if (T.beg[i] == ~0ull) throw InvalidError("internal: a merged cell's molecule rows were not exported");
		for (uint64_t r = T.beg[i]; r < T.end[i]; ++r) { import_row.push_back(u32(r)); import_cell.push_back(Gm[t].local_id); if (with_qual) import_rank.push_back(A.mrank[i]); }
	}
	std::sort(merged_barcodes.begin(), merged_barcodes.end());
	const u32 ni = u32(import_row.size());
	DevBuf<u32> d_row, d_cell, d_col[4]; DevBuf<u64> d_low;
	d_row.alloc(std::max<u32>(ni, 1)); d_cell.alloc(std::max<u32>(ni, 1)); d_low.alloc(std::max<u32>(ni, 1));
	for (auto &b : d_col) b.alloc(std::max<u32>(ni, 1));
	if (ni) {
		HIP_CHECK(hipMemcpyAsync(d_row.p, import_row.data(), size_t(ni) * 4, hipMemcpyHostToDevice, c.stream));
		HIP_CHECK(hipMemcpyAsync(d_cell.p, import_cell.data(), size_t(ni) * 4, hipMemcpyHostToDevice, c.stream));
		ImportGatherArgs a{};
		a.row = d_row.p; a.n = ni; a.low_all = T.low_all.p; a.o_low = d_low.p;
		for (int k = 0; k < 4; ++k) { a.col_all[k] = T.col_all[k].p; a.o_col[k] = d_col[k].p; }
		hipLaunchKernelGGL(import_gather_kernel, dim3(div_up(ni, 256)), dim3(256), 0, c.stream, a);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(stream_wait(c.stream));
	}
	DevBuf<u32> d_irank, d_iq;
	if (with_qual) {
		const u32 qs = c.qual_stride();
		d_irank.alloc(std::max<u32>(ni, 1)); d_iq.alloc(std::max<size_t>(size_t(ni) * qs, 1));
		if (ni) {
			HIP_CHECK(hipMemcpyAsync(d_irank.p, import_rank.data(), size_t(ni) * 4, hipMemcpyHostToDevice, c.stream));
			hipLaunchKernelGGL(gather_u32_rows_kernel, dim3(u32(std::min<uint64_t>((uint64_t(ni) * qs + 255) / 256, 16384))), dim3(256), 0, c.stream, T.q_all.p, d_row.p, ni, qs, d_iq.p);
			HIP_CHECK(hipGetLastError());
			HIP_CHECK(stream_wait(c.stream));
		}
		c.shard_merge_quality_import(local_id.size(), local_rank.data(), d_irank.p, d_iq.p);
	}
	const u32 *cols[4] = {d_col[0].p, d_col[1].p, d_col[2].p, d_col[3].p};
	c.shard_merge_finish(local_id.size(), local_id.data(), l_excl.data(), l_merged.data(), l_reads.data(), l_umis.data(), move_src.size(),
	                     move_src.data(), move_tgt.data(), ni, d_cell.p, reinterpret_cast<const uint64_t *>(d_low.p), cols);
}

// RealBarcodes / PoissonRealBarcodes: phases in merge_shard.h
void dropest_shard::cb_merge() {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	MergeWorld W;
	merge_gather_cells(W);
	const auto &Gm = W.Gm;
	const u32 nG = W.nG, lo = W.lo, hi = W.hi;
	// search: my real cells against everybody's
	std::vector<uint64_t> g_bc(nG); std::vector<u32> g_ng(nG); std::vector<int32_t> g_tu(nG);
	for (u32 i = 0; i < nG; ++i) { g_bc[i] = Gm[i].barcode; g_ng[i] = Gm[i].n_genes; g_tu[i] = Gm[i].total_umis; }
	std::vector<u32> base_g(hi - lo), base_l(hi - lo);
	for (u32 i = lo; i < hi; ++i) { base_g[i - lo] = i; base_l[i - lo] = Gm[i].local_id; }
	uint64_t n_pairs = 0;
	{ Phase ph(this, "cbm:search"); c.shard_merge_search(nG, g_bc.data(), g_ng.data(), g_tu.data(), hi - lo, base_g.data(), base_l.data(), &n_pairs); }
	dropest_ctx::ShardMerge &M = *c.shard;
	struct Pair { u32 base, cand; };
	std::vector<Pair> my_pairs(static_cast<size_t>(n_pairs)), allp;
	for (size_t p = 0; p < n_pairs; ++p) my_pairs[p] = Pair{M.base_g[M.S.pair_base[p]], M.S.pair_cand[p]};
	std::vector<size_t> pcnt;
	{ Phase ph(this, "cbm:gather_pairs"); tr->gather_vec(my_pairs, allp, pcnt); }
	TravelRows T;
	merge_gather_rows(W, T);
	DevBuf<u64> &low_all = T.low_all;
	std::vector<uint64_t> &beg = T.beg, &end = T.end;
	const bool poisson = c.cfg.merge_kind == DROPEST_MERGE_POISSON_REAL;
	if (poisson) {
		// -M: the estimator works on the UMI distribution of ALL filtered cells and on the largest gene of any cell (PoissonTargetEstimator::init,
		// PoissonTargetEstimator.cpp:46-59): dense histograms over the UMI field, all-gathered and added; every shard builds the same tables
		merge_umi_distribution();
	}
	// intersect: the pairs whose candidate is mine
	std::vector<size_t> poff(size_t(world) + 1, 0);
	for (int p = 0; p < world; ++p) poff[size_t(p) + 1] = poff[size_t(p)] + pcnt[size_t(p)];
	struct Ans { uint64_t pair; u32 inter, pad; double expected; };
	std::vector<Ans> my_ans, all_ans;
	{
		Phase ph(this, "cbm:intersect");
		std::vector<u32> cand_local; std::vector<uint64_t> bb, be;
		std::vector<uint64_t> which;
		for (size_t i = 0; i < allp.size(); ++i)
			if (allp[i].cand >= lo && allp[i].cand < hi) {
				which.push_back(i); cand_local.push_back(Gm[allp[i].cand].local_id);
				if (beg[allp[i].base] == ~0ull) throw InvalidError("internal: a base's molecule rows were not exported");
				bb.push_back(beg[allp[i].base]); be.push_back(end[allp[i].base]);
			}
		std::vector<u32> inter(which.size());
		std::vector<double> expected(which.size(), 0.0);
		c.shard_merge_intersect(which.size(), cand_local.data(), bb.data(), be.data(), reinterpret_cast<const uint64_t *>(low_all.p), inter.data());
		if (poisson) c.shard_merge_expected(which.size(), cand_local.data(), bb.data(), be.data(), reinterpret_cast<const uint64_t *>(low_all.p), expected.data());
		my_ans.resize(which.size());
		for (size_t i = 0; i < which.size(); ++i) my_ans[i] = Ans{which[i], inter[i], 0, expected[i]};
		std::vector<size_t> acnt;
		tr->gather_vec(my_ans, all_ans, acnt);
	}
	std::vector<u32> inter_all(allp.size(), 0);
	std::vector<double> expected_all(poisson ? allp.size() : 0, 0.0);
	for (const Ans &a : all_ans) { inter_all[size_t(a.pair)] = a.inter; if (poisson) expected_all[size_t(a.pair)] = a.expected; }
	// decide: targets of my bases; then the same sequential application everywhere
	std::vector<int64_t> my_tgt(hi - lo, -1);
	{
		Phase ph(this, "cbm:decide");
		if (poisson) c.shard_merge_decide_poisson(inter_all.data() + poff[size_t(rank)], expected_all.data() + poff[size_t(rank)], my_tgt.data());
		else c.shard_merge_decide(inter_all.data() + poff[size_t(rank)], my_tgt.data());
	}
	MergeApplied A;
	merge_apply(W, my_tgt, merge_order(W), A);   // all real cells are "filtered" before the merge (threshold 0)
	merge_finish(W, A, T);
}

// -M: the UMI distribution of ALL shards' filtered cells -> the estimator's tables, identical on every shard
void dropest_shard::merge_umi_distribution() {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	Phase ph(this, "cbm:umi_distribution");
	DevBuf<u32> hist, all, sum;
	uint64_t kept = 0; u32 max_size = 0;
	c.shard_merge_umi_histogram(hist, kept, max_size);
	const size_t n = size_t(1) << c.layout.umi_bits;
	uint64_t mine[3] = {kept, max_size, n};
	std::vector<uint64_t> every(size_t(world) * 3);
	tr->gather_host(mine, sizeof(mine), every.data());
	uint64_t kept_total = 0; u32 max_all = 0;
	for (int p = 0; p < world; ++p) {
		if (every[size_t(p) * 3 + 2] != n) throw InvalidError("internal: the shards disagree on the width of the UMI field");
		kept_total += every[size_t(p) * 3]; max_all = std::max<u32>(max_all, u32(every[size_t(p) * 3 + 1]));
	}
	all.alloc(n * size_t(world)); sum.alloc(n);
	std::vector<size_t> off(static_cast<size_t>(world)), bytes(static_cast<size_t>(world), n * 4);
	for (int p = 0; p < world; ++p) off[size_t(p)] = size_t(p) * n * 4;
	tr->gather_dev(hist.p, all.p, off.data(), bytes.data(), c.stream);
	hipLaunchKernelGGL(sum_rows_kernel, dim3(u32(std::min<size_t>((n + 255) / 256, 4096))), dim3(256), 0, c.stream, all.p, u32(world), uint64_t(n), sum.p);
	HIP_CHECK(hipGetLastError());
	c.shard_merge_set_umi_distribution(sum.p, n, kept_total, max_all);
}

#include "shard_merge_free.h"

// 7. global view of the real cells: every shard's rows, all-gathered; first_global = the stream ordinal of the cell's first read
void dropest_shard::build_global_table() {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	Phase ph(this, "cells_allgather");
	std::vector<GRow> mine;
	std::vector<u32> pos;
	auto take = [&](const HostCell &h) {
		GRow r{};
		r.barcode = h.row.barcode; r.n_genes = h.row.n_genes; r.req_genes = h.row.requested_genes; r.req_umis = h.row.requested_umis;
		r.local_id = h.id; r.total_umis = h.row.total_umis; r.total_reads = h.row.total_reads; r.rank = u32(rank);
		mine.push_back(r); pos.push_back(h.row.first_read);
	};
	if (raw_device_now && c.cfg.max_cells <= 0) {
		// cm_raw is planned on the device: the table serves cm only, and its rows are the context's own filtered cells (collected on a few
		// threads by sort_filtered; without -C nothing is cut from them locally) -- not another walk over every real cell
		c.filtered_cells();
		std::vector<u32> ridx(c.filtered_ridx);
		std::sort(ridx.begin(), ridx.end());   // cell-id order, as the walk below gives it
		for (u32 ri : ridx) take(c.real[ri]);
	} else
	for (const HostCell &h : c.real) {
		if (h.merged || h.excluded || h.row.n_genes < c.min_before) continue;
		if (raw_device_now && h.row.requested_genes < c.min_after) continue;   // cm_raw is planned on the device: the table serves cm only
		take(h);
	}
	// first_global orders cm_raw's columns when the HOST plans them; cm is ordered by compare_cells: no device round trip for it
	if (!raw_device_now) { Phase p2(this, "cells:ordinals");
	const std::vector<u64> ord = global_ordinals(pos);
	for (size_t i = 0; i < mine.size(); ++i) mine[i].first_global = ord[i]; }
	std::vector<size_t> cnt;
	{ Phase p3(this, "cells:gather"); tr->gather_vec(mine, G, cnt); }
	if (std::getenv("DROPEST_SHARD_TRACE")) fprintf(stderr, "[shard %d] real rows %zu of %zu, G %zu\n", rank, mine.size(), c.real.size(), G.size());
}

// 8. one matrix: global column order (identical on every shard), this shard's columns emitted on the device and written to
// their places in the node-shared host buffer
void dropest_shard::assemble_matrix(bool filtered_m) {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	const int slot = filtered_m ? 0 : 1;
	Mat &M = mat[slot];
	ctx->mat[slot].settle();   // (a straggler of the previous widening reads dec_begin / dec_end)
	dropest::DevBuf<u64> &d_desc = d_desc_m[slot];
	dropest::PinnedBuf<u64> &h_desc = h_desc_m[slot];
	std::vector<u32> sel;
	for (u32 i = 0; i < G.size(); ++i) if (!filtered_m || G[i].req_genes >= c.min_after) sel.push_back(i);
	std::vector<u32> order;
	{
		Phase ph(this, filtered_m ? "order:cm" : "order:cm_raw");
		order = order_rows(sel, !filtered_m);
		if (filtered_m && c.cfg.max_cells > 0 && size_t(c.cfg.max_cells) < order.size())   // -C: the largest max_cells cells (CellsDataContainer.cpp:269-273)
			order.erase(order.begin(), order.end() - c.cfg.max_cells);
	}
	Phase ph(this, filtered_m ? "matrix:cm" : "matrix:cm_raw");
	const size_t ncols = order.size();
	M.ncols = ncols; M.colptr.assign(ncols + 1, 0); M.col_barcode.resize(ncols);
	std::vector<u32> &col_cell = M.col_cell, &col_start = M.col_start;   // (members: their copies to the device are not waited for here)
	col_cell.clear(); col_start.clear();
	std::vector<u64> desc;
	uint64_t local_nnz = 0;
	for (size_t j = 0; j < ncols; ++j) {
		const GRow &r = G[order[j]];
		const u32 len = filtered_m ? r.req_genes : r.n_genes;
		M.col_barcode[j] = r.barcode;
		M.colptr[j + 1] = M.colptr[j] + len;
		if (int(r.rank) != rank) continue;
		col_cell.push_back(r.local_id); col_start.push_back(u32(local_nnz));
		desc.push_back(local_nnz); desc.push_back(M.colptr[j]); desc.push_back(len);
		local_nnz += len;
	}
	M.dec_begin.clear(); M.dec_end.clear();
	for (size_t k = 0; k < col_cell.size(); ++k) { M.dec_begin.push_back(u32(desc[3 * k + 1])); M.dec_end.push_back(u32(desc[3 * k + 1] + desc[3 * k + 2])); }
	M.nnz = M.colptr[ncols];
	if (M.nnz > 0xFFFFFFF0ull || local_nnz > 0xFFFFFFF0ull) throw UnsupportedError("count matrix with more than 2^32 non-zeros");
	M.colptr_p = M.colptr.data(); M.barcode_p = M.col_barcode.data();
	M.colptr32.resize(ncols + 1);
	for (size_t j = 0; j <= ncols; ++j) M.colptr32[j] = u32(M.colptr[j]);
	M.colptr32_p = M.colptr32.data();
	SharedLayout L;
	{ Phase p2(this, filtered_m ? "cm:open_shared" : "raw:open_shared"); L = open_shared(M, slot, 0); }
	const u32 nc = u32(col_cell.size());
	if (nc && local_nnz) {
		if (!L.slots) c.emit_columns_device(filtered_m, false, col_cell, col_start, local_nnz, false);   // (a slots step emits the byte form of the local matrix: place_columns)
		// (the same matrix assembled again inside one step -- the overflow fall-back -- or in the next step: the previous copy has left the
		// staging buffer by then in every run seen, but nothing said so: wait, it costs nothing on a drained stream)
		if (desc_in_flight[slot]) HIP_CHECK(stream_wait(c.stream));
		d_desc.ensure(desc.size()); h_desc.ensure(desc.size());
		std::memcpy(h_desc.p, desc.data(), desc.size() * 8);
		HIP_CHECK(hipMemcpyAsync(d_desc.p, h_desc.p, desc.size() * 8, hipMemcpyHostToDevice, c.stream));
		desc_in_flight[slot] = true;
	}
	Phase p3(this, filtered_m ? "cm:place" : "raw:place");
	place_columns(M, slot, filtered_m, L, d_desc.p, nc, local_nnz, nullptr, col_start.data());
}

// The node-shared host buffer of one matrix (collective: every shard takes the same decisions -- the gene ids are the agreed ones).
// [head_bytes of the caller][payload][byte form: one segment per shard].  Payload: 32-bit rows | values; 16-bit rows | values; byte form:
// deltas | values (each padded to 16 bytes).  A segment = 4 words (rows listed, values listed, 0, 0) + 4 arrays of list_cap words (row
// positions, rows, value positions, values); list_cap comes from the GLOBAL nnz: the same on every shard.
dropest_shard::SharedLayout dropest_shard::open_shared(Mat &M, int slot, size_t head_bytes) {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	SharedLayout L{};
	L.bytes = byte_matrix && !wide_now[slot];
	L.narrow = !L.bytes && narrow_matrix && c.narrow_possible();
	L.base = (head_bytes + 15) & ~size_t(15);
	L.list_cap = u32(std::min<uint64_t>(byte_list_cap ? byte_list_cap : (1u << 20), (M.nnz + 15) & ~15ull));
	L.off_val = (size_t(M.nnz) + 15) & ~size_t(15); L.off_seg = 2 * L.off_val; L.seg_bytes = 16 + 16 * size_t(L.list_cap);
	// slots step: the buffer holds the 32-bit slots only -- rows | values, each padded to whole 64-byte lines (the widening stores whole
	// lines); every shard's bytes cross its link as the byte form of its LOCAL matrix and never stand in the shared buffer
	L.slots = L.bytes && slots_matrix;
	if (L.slots) L.bytes = false;
	L.off_slots = (L.base + 63) & ~size_t(63);   // (from the buffer's start: that is page-aligned)
	L.slots_stride = ((size_t(M.nnz) + 15) & ~size_t(15)) * 4;
	const size_t payload = L.slots ? L.off_slots - L.base + std::max<size_t>(2 * L.slots_stride, 64)   // (an empty matrix still gets a buffer: the transports map what they are asked for)
	                               : (L.bytes ? L.off_seg + L.seg_bytes * size_t(world) : std::max<size_t>(size_t(M.nnz), 1) * (L.narrow ? 4 : 8));
	ctx->mat[slot].settle();   // (a straggler of the previous step's widening may still be leaving: the buffer is about to be rewritten or replaced)
	L.host = static_cast<char *>(tr->shared_host(slot, L.base + payload, &L.dev));
	char *host = L.host + L.base;
	M.narrow = L.narrow; M.widened = false; M.ovf_pos.clear(); M.ovf_val.clear();
	M.bytes = L.bytes; M.lists_ready = false;
	M.rows = M.vals = nullptr; M.rows16 = M.vals16 = nullptr; M.delta8 = M.vals8 = nullptr;
	M.slots = L.slots; M.slot_rows = M.slot_vals = nullptr; M.shipped = false; M.encoded = false;
	if (L.slots) {
		M.slot_rows = reinterpret_cast<u32 *>(L.host + L.off_slots); M.slot_vals = reinterpret_cast<u32 *>(L.host + L.off_slots + L.slots_stride);
		M.d_slot_rows = reinterpret_cast<u32 *>(static_cast<char *>(L.dev) + L.off_slots); M.d_slot_vals = reinterpret_cast<u32 *>(static_cast<char *>(L.dev) + L.off_slots + L.slots_stride);
		M.rows = M.slot_rows; M.vals = M.slot_vals;
		return L;
	}
	if (L.bytes) {
		M.delta8 = reinterpret_cast<const uint8_t *>(host); M.vals8 = reinterpret_cast<const uint8_t *>(host) + L.off_val;
		M.segments = host + L.off_seg; M.seg_bytes = L.seg_bytes; M.list_cap = L.list_cap;
	} else if (L.narrow) { M.rows16 = reinterpret_cast<const uint16_t *>(host); M.vals16 = reinterpret_cast<const uint16_t *>(host) + M.nnz; }
	else { M.rows = reinterpret_cast<const u32 *>(host); M.vals = reinterpret_cast<const u32 *>(host) + M.nnz; }
	return L;
}

// this shard's columns (emitted into c.mat[slot] on the device; d_descr: local offset, global offset, length of each) -> their global
// places in the shared buffer, in the form open_shared chose
void dropest_shard::place_columns(Mat &M, int slot, bool filtered_m, const SharedLayout &L, const unsigned long long *d_descr, dropest::u32 nc, uint64_t local_nnz, hipStream_t st,
                                  const dropest::u32 *local_start) {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	if (!st) st = c.stream;
	char *d_payload = static_cast<char *>(L.dev) + L.base;
	const dropest_ctx::MatrixResult &R = c.mat[slot];
	const bool work = nc && local_nnz;
	if (L.slots) {
		// The columns leave as the byte form of this shard's LOCAL matrix (the emit and the chunked copies of one context: whole lines at the
		// link's streaming rate, whatever the columns' lengths -- placing bytes at their GLOBAL places straight from a kernel met ragged column
		// ends with byte stores over PCIe: 22 GB/s on cm_raw's short columns), and the pool's host threads widen every chunk of columns into the
		// shared 32-bit slots at the columns' global places as soon as its arrival flag is up (matrix_decode.h).
		M.d_descr = d_descr; M.nc = nc; M.local_nnz = local_nnz;
		if (!work) return;
		dropest_ctx::WireTarget T;
		T.rows = M.slot_rows; T.vals = M.slot_vals; T.global_nnz = M.nnz; T.d_descr = d_descr;
		T.begin = M.dec_begin.empty() ? M.h_begin.p : M.dec_begin.data();
		T.end = M.dec_end.empty() ? M.h_end.p : M.dec_end.data();
		const std::vector<u32> &cells = filtered_m ? M.col_cell : raw_plan.col_cell, &starts = filtered_m ? M.col_start : raw_plan.col_start;
		c.ship_columns_to_slots(filtered_m, false, cells, starts, local_nnz, T, st);
		M.shipped = true;
		return;
	}
	if (L.bytes) {
		u32 *seg_words = reinterpret_cast<u32 *>(d_payload + L.off_seg + L.seg_bytes * size_t(rank));
		d_list_count.ensure(8);
		u32 *cnt = d_list_count.p + 4 * slot;
		HIP_CHECK(hipMemsetAsync(cnt, 0, 16, st));
		if (work) {
			u32 *lists = seg_words + 4;
			const size_t cap = L.list_cap;
			{
			auto launch = [&] {
				hipLaunchKernelGGL(place_columns_bytes_kernel, dim3(nc), dim3(filtered_m ? 256 : 64), 0, st, d_descr, R.d_row.p, R.d_val.p,
				                   reinterpret_cast<uint8_t *>(d_payload), reinterpret_cast<uint8_t *>(d_payload) + L.off_val, cnt,
				                   lists, lists + cap, lists + 2 * cap, lists + 3 * cap, L.list_cap);
			};
			if (st == c.stream) c.timed(filtered_m ? "place_columns:cm" : "place_columns:cm_raw", double(local_nnz) * 10, launch);
			else launch();   // (the context's launch timer brackets its own stream)
			}
		}
		hipLaunchKernelGGL(copy_words_kernel, dim3(1), dim3(64), 0, st, cnt, seg_words, 4u);   // the counts, behind the lists on the stream
		HIP_CHECK(hipGetLastError());
		return;
	}
	constexpr u32 OVF_CAP = 1u << 16;
	if (work) {
		if (L.narrow) {
			d_ovf.ensure(1 + 3 * size_t(OVF_CAP));
			HIP_CHECK(hipMemsetAsync(d_ovf.p, 0, 4, c.stream));
			hipLaunchKernelGGL(place_columns_narrow_kernel, dim3(nc), dim3(256), 0, c.stream, d_descr, R.d_row.p, R.d_val.p,
			                   reinterpret_cast<uint16_t *>(d_payload), reinterpret_cast<uint16_t *>(d_payload) + M.nnz, d_ovf.p, OVF_CAP);
		} else
			hipLaunchKernelGGL(place_columns_kernel, dim3(nc), dim3(256), 0, c.stream, d_descr, R.d_row.p, R.d_val.p,
			                   reinterpret_cast<u32 *>(d_payload), reinterpret_cast<u32 *>(d_payload) + M.nnz);
		HIP_CHECK(hipGetLastError());
	}
	if (L.narrow) {   // the (rare) entries beyond 16 bits of every shard, gathered on the host
		struct Ovf { u64 pos; u32 val, pad; };
		std::vector<Ovf> mine, all;
		if (work) {
			u32 count = 0;
			c.fetch(&count, d_ovf.p, 4);
			if (count > OVF_CAP) throw UnsupportedError("more than 2^16 matrix entries beyond 65534 on one shard: set the shard option narrow_matrix to 0");
			if (count) {
				std::vector<u32> raw(3 * size_t(count));
				c.fetch(raw.data(), d_ovf.p + 1, raw.size() * 4);
				for (u32 i = 0; i < count; ++i) mine.push_back(Ovf{u64(raw[3 * i]) | (u64(raw[3 * i + 1]) << 32), raw[3 * i + 2], 0});
			}
		}
		std::vector<size_t> cnt;
		tr->gather_vec(mine, all, cnt);
		std::sort(all.begin(), all.end(), [](const Ovf &a, const Ovf &b) { return a.pos < b.pos; });
		for (const Ovf &o : all) { M.ovf_pos.push_back(o.pos); M.ovf_val.push_back(o.val); }
	}
}

// cm_raw without a host table of all the real cells.  Its columns are the real cells of ALL shards in the order of ONE container's cell
// ids = the stream ordinal of every cell's first read; on a shard the real cells stand in that order already (local ids are first-seen
// ranks, and the blocks a shard received stand in stream order).  So the global matrix is a merge of `world` ascending lists:
//   plan_raw           (host) this shard's columns, their nnz prefix; ONE small collective: columns, nnz and ordinal queries per shard
//   assemble_raw_device        first-read ordinals on the device (the positions did not travel with a packed exchange: two small
//                              device all-to-alls ask the sources), all-gather of (ordinal, nnz prefix) of every shard's columns --
//                              16 bytes per column over the links --, then one thread per local column: binary searches in the other
//                              shards' lists give its global place and nnz offset; it writes colptr / barcode of its column into
//                              the shared buffer and the descriptor the placing kernel reads.
// The host table (build_global_table) then holds the filtered candidates only.  Falls back to the host plan when a shard's cells are
// not in ordinal order (never seen: a guard) or the option raw_on_device is off.
bool dropest_shard::plan_raw() {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	Phase ph(this, "raw:plan");
	RawPlan &P = raw_plan;
	P.col_cell.clear(); P.col_start.clear(); P.query.clear(); P.pre.assign(1, 0);
	const bool ask = exchanged && !idx_exchanged;
	P.q_out.assign(size_t(world), 0);
	uint64_t ok = raw_on_device ? 1 : 0;
	u32 last = 0, src = 0;
	for (const HostCell &h : c.real) {
		if (h.merged || h.excluded || h.row.n_genes < c.min_before) continue;
		const u32 pos = h.row.first_read;
		if (pos == 0xFFFFFFFFu || (!P.col_cell.empty() && pos <= last)) { ok = 0; break; }
		last = pos;
		if (P.pre.back() + h.row.n_genes > 0xFFFFFFF0ull) throw UnsupportedError("count matrix with more than 2^32 non-zeros");
		P.col_cell.push_back(h.id); P.col_start.push_back(u32(P.pre.back())); P.pre.push_back(P.pre.back() + h.row.n_genes);
		if (ask) {
			while (src + 1 < u32(world) && pos >= recv_off[size_t(src) + 1]) ++src;
			P.query.push_back(u32(pos - recv_off[src])); ++P.q_out[src];
		} else
			P.query.push_back(pos);
	}
	// [ok, columns, nnz, queries to shard 0 .. world-1] of every shard
	const size_t w = 3 + size_t(world);
	std::vector<uint64_t> mine(w, 0);
	mine[0] = ok; mine[1] = P.col_cell.size(); mine[2] = P.pre.back();
	for (int p = 0; p < world; ++p) mine[3 + size_t(p)] = P.q_out[size_t(p)];
	P.counts.assign(w * size_t(world), 0);
	tr->gather_host(mine.data(), w * 8, P.counts.data());
	P.ncols = P.nnz = 0;
	P.q_in.assign(size_t(world), 0);
	for (int p = 0; p < world; ++p) {
		const uint64_t *row = P.counts.data() + w * size_t(p);
		ok &= row[0]; P.ncols += row[1]; P.nnz += row[2]; P.q_in[size_t(p)] = row[3 + size_t(rank)];
	}
	if (P.nnz > 0xFFFFFFF0ull) throw UnsupportedError("count matrix with more than 2^32 non-zeros");
	return ok != 0;
}

void dropest_shard::assemble_raw_device() {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	Phase ph(this, "matrix:cm_raw");
	RawPlan &P = raw_plan;
	Mat &M = mat[1];
	ctx->mat[1].settle();
	const size_t w = 3 + size_t(world);
	const u32 nl = u32(P.col_cell.size());
	const uint64_t ncols = P.ncols;
	M.ncols = ncols; M.nnz = P.nnz; M.colptr.clear(); M.col_barcode.clear(); M.colptr32.clear();
	// head of the shared buffer: colptr64[ncols + 1] | colptr32[ncols + 1] | barcodes[ncols]
	const size_t off_c32 = (ncols + 1) * 8, off_bc = (off_c32 + (ncols + 1) * 4 + 15) & ~size_t(15), head = off_bc + ncols * 8;
	const SharedLayout L = open_shared(M, 1, head);
	M.colptr_p = reinterpret_cast<const u64 *>(L.host); M.colptr32_p = reinterpret_cast<const u32 *>(L.host + off_c32);
	M.barcode_p = reinterpret_cast<const u64 *>(L.host + off_bc);
	if (rank == 0) {   // the closing entry is nobody's column
		reinterpret_cast<u64 *>(L.host)[ncols] = P.nnz;
		reinterpret_cast<u32 *>(L.host + off_c32)[ncols] = u32(P.nnz);
	}
	// this shard's (first-read ordinal [nl] | nnz prefix [nl + 1])
	d_plan_mine.ensure(2 * size_t(nl) + 1); d_q.ensure(std::max<u32>(nl, 1)); d_col_cell.ensure(std::max<u32>(nl, 1));
	h_plan.ensure(size_t(nl) + 1 + size_t(nl));   // staging of its own (cm's descriptors may still be on their way from their staging buffer): prefix (u64) | queries, cells (u32)
	std::memcpy(h_plan.p, P.pre.data(), (size_t(nl) + 1) * 8);
	u32 *h32 = reinterpret_cast<u32 *>(h_plan.p + nl + 1);
	if (nl) { std::memcpy(h32, P.query.data(), size_t(nl) * 4); std::memcpy(h32 + nl, P.col_cell.data(), size_t(nl) * 4); }
	HIP_CHECK(hipMemcpyAsync(d_plan_mine.p + nl, h_plan.p, (size_t(nl) + 1) * 8, hipMemcpyHostToDevice, c.stream));
	if (nl) {
		HIP_CHECK(hipMemcpyAsync(d_q.p, h32, size_t(nl) * 4, hipMemcpyHostToDevice, c.stream));
		HIP_CHECK(hipMemcpyAsync(d_col_cell.p, h32 + nl, size_t(nl) * 4, hipMemcpyHostToDevice, c.stream));
	}
	if (exchanged && !idx_exchanged) {
		QueryTable t{};
		t.world = u32(world);
		uint64_t n_in = 0;
		for (int p = 0; p < world; ++p) {
			t.in_off[p] = n_in; n_in += P.q_in[size_t(p)]; t.part_off[p] = send_off[size_t(p)]; t.first_ord[p] = first_ord[size_t(p)];
			t.out_off[p + 1] = t.out_off[p] + P.q_out[size_t(p)];
		}
		t.in_off[world] = n_in;
		d_q_in.ensure(std::max<uint64_t>(n_in, 1)); d_q_ans.ensure(std::max<uint64_t>(n_in, 1)); d_q_back.ensure(std::max<u32>(nl, 1));
		{ const void *snd[1] = {d_q.p}; void *rcv[1] = {d_q_in.p}; const size_t elem[1] = {4};
		  tr->exchange(1, snd, rcv, elem, P.q_out.data(), P.q_in.data(), c.stream); }
		if (n_in) {
			hipLaunchKernelGGL(answer_queries_kernel, dim3(u32((n_in + 255) / 256)), dim3(256), 0, c.stream, t, d_q_in.p, u32(n_in), p_idx.p, d_q_ans.p);
			HIP_CHECK(hipGetLastError());
		}
		{ const void *snd[1] = {d_q_ans.p}; void *rcv[1] = {d_q_back.p}; const size_t elem[1] = {4};
		  tr->exchange(1, snd, rcv, elem, P.q_in.data(), P.q_out.data(), c.stream); }
		if (nl) hipLaunchKernelGGL(ordinals_from_answers_kernel, dim3((nl + 255) / 256), dim3(256), 0, c.stream, t, d_q_back.p, nl, reinterpret_cast<unsigned long long *>(d_plan_mine.p));
	} else if (nl)
		hipLaunchKernelGGL(ordinals_kernel, dim3((nl + 255) / 256), dim3(256), 0, c.stream, ordinal_map(), d_q.p, nl, reinterpret_cast<unsigned long long *>(d_plan_mine.p));
	HIP_CHECK(hipGetLastError());
	// every shard's block to every shard
	RawPlanTable pt{};
	pt.world = u32(world); pt.rank = u32(rank);
	std::vector<size_t> off(static_cast<size_t>(world), 0), bytes(static_cast<size_t>(world), 0);
	size_t total = 0;
	for (int p = 0; p < world; ++p) {
		const uint64_t n_p = P.counts[w * size_t(p) + 1];
		pt.off[p] = total; pt.n[p] = u32(n_p);
		off[size_t(p)] = total * 8; bytes[size_t(p)] = (2 * size_t(n_p) + 1) * 8;
		total += 2 * size_t(n_p) + 1;
	}
	d_plan_all.ensure(total);
	tr->gather_dev(d_plan_mine.p, d_plan_all.p, off.data(), bytes.data(), c.stream);
	d_desc_raw.ensure(std::max<size_t>(3 * size_t(nl), 1));
	M.dec_begin.clear(); M.dec_end.clear();
	if (L.slots) { M.h_begin.ensure(std::max<u32>(nl, 1)); M.h_end.ensure(std::max<u32>(nl, 1)); }
	if (nl) {
		char *d_head = static_cast<char *>(L.dev);
		hipLaunchKernelGGL(plan_raw_columns_kernel, dim3((nl + 255) / 256), dim3(256), 0, c.stream, pt, reinterpret_cast<const unsigned long long *>(d_plan_all.p),
		                   d_col_cell.p, reinterpret_cast<const unsigned long long *>(c.cell_cb.p), nl, reinterpret_cast<unsigned long long *>(d_desc_raw.p),
		                   reinterpret_cast<unsigned long long *>(d_head), reinterpret_cast<uint32_t *>(d_head + off_c32),
		                   reinterpret_cast<unsigned long long *>(d_head + off_bc), L.slots ? M.h_begin.p : static_cast<u32 *>(nullptr),
		                   L.slots ? M.h_end.p : static_cast<u32 *>(nullptr));
		HIP_CHECK(hipGetLastError());
	}
	const uint64_t local_nnz = P.pre.back();
	if (nl && local_nnz && !L.slots) c.emit_columns_device(false, false, P.col_cell, P.col_start, local_nnz, false);
	// the placing kernel on its own stream (byte form: nothing on the host waits for it before the end of the step)
	hipStream_t st = c.stream;
	if ((L.bytes || L.slots) && !getenv("DROPEST_SHARD_NO_PLACE_STREAM")) {
		if (!place_stream) { HIP_CHECK(hipStreamCreateWithFlags(&place_stream, hipStreamNonBlocking)); HIP_CHECK(hipEventCreateWithFlags(&ev_place, hipEventDisableTiming)); }
		HIP_CHECK(hipEventRecord(ev_place, c.stream));
		HIP_CHECK(hipStreamWaitEvent(place_stream, ev_place, 0));
		st = place_stream;
	}
	place_columns(M, 1, false, L, reinterpret_cast<const unsigned long long *>(d_desc_raw.p), nl, local_nnz, st, P.col_start.data());
}

void dropest_shard::step() {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	HIP_CHECK(hipSetDevice(c.cfg.device));
	Phase whole(this, "step");
	// the ordinal ranges of all shards (ordinals name reads across shards: first-seen order, N-UMI tie breaks)
	if (pushed.n) {   // reads pushed from host memory: wait for the last batch, use the store in place
		pushed.wait();
		r_cb = pushed.cb.p; r_umi = pushed.umi.p; r_gene = pushed.gene.p; r_aux = pushed.aux.p; n_res = pushed.n;
	}
	first_ord.assign(size_t(world), 0);
	{
		uint64_t mine[2] = {first_ordinal, n_res};
		std::vector<uint64_t> all(size_t(world) * 2);
		tr->gather_host(mine, sizeof(mine), all.data());
		uint64_t reach = 0;
		for (int p = 0; p < world; ++p) {
			first_ord[size_t(p)] = all[size_t(p) * 2];
			if (all[size_t(p) * 2 + 1] == 0) continue;   // an empty shard has no place in the order
			if (first_ord[size_t(p)] < reach) throw InvalidError("the shards' ordinal ranges must ascend with the rank and must not overlap");
			reach = first_ord[size_t(p)] + all[size_t(p) * 2 + 1];
		}
	}
	{   // UMI qualities: all shards or none, one length, one string per resident read
		uint64_t mine[2] = {(r_have_qual ? 1ull : 0ull) | (r_have_qual && r_qual_var ? 2ull : 0ull), r_qlen};
		std::vector<uint64_t> all(size_t(world) * 2);
		tr->gather_host(mine, sizeof(mine), all.data());
		for (int p = 0; p < world; ++p)
			if (all[size_t(p) * 2] != mine[0] || all[size_t(p) * 2 + 1] != mine[1])
				throw InvalidError("UMI qualities must be given to every shard of the run, in one way (with or without per-read lengths) and with one row width (dropest_shard_set_umi_qualities)");
		if (r_have_qual && r_qual_reads != n_res)
			throw InvalidError("UMI qualities were given for " + std::to_string(r_qual_reads) + " reads, the shard holds " + std::to_string(n_res));
	}
	// forget the previous pass
	c.free_results(); c.chunks.clear(); c.n_reads = 0; c.store.clear(); c.store_chunk = -1;
	c.d_cb = c.d_umi = nullptr; c.d_gene = c.d_aux = nullptr;
	exchanged = false;
	ReadChunk ch;
	if (world > 1 || force_exchange) {
		unpacked = false;
		c.rpack = dropest::ReadPack{}; c.unpack_reads = nullptr;
		partition_and_exchange();
		ch.n = recv_off[size_t(world)];
		if (packed && !unpacked && ch.n) {
			// the context reads the 12-byte records as they are; the four columns are restored only if something asks for them
			ch.p_cb = x_w0.p; ch.p_umi = x_w0.p; ch.p_gene = x_w1.p; ch.p_aux = x_w1.p;
			c.rpack.cb_bits = exch_pack.cb_bits; c.rpack.gene_bits = exch_pack.gene_bits;
			c.unpack_reads = [this] {
				dropest_ctx &cc = *ctx;
				unpack_exchanged();
				cc.d_cb = x_cb.p; cc.d_umi = x_umi.p; cc.d_gene = x_gene.p; cc.d_aux = x_aux.p;
				if (!cc.chunks.empty()) { cc.chunks[0].p_cb = x_cb.p; cc.chunks[0].p_umi = x_umi.p; cc.chunks[0].p_gene = x_gene.p; cc.chunks[0].p_aux = x_aux.p; }
			};
		} else { ch.p_cb = x_cb.p; ch.p_umi = x_umi.p; ch.p_gene = x_gene.p; ch.p_aux = x_aux.p; }
	} else {   // one shard: every read is at home already
		c.rpack = dropest::ReadPack{}; c.unpack_reads = nullptr;
		ch.p_cb = r_cb; ch.p_umi = r_umi; ch.p_gene = r_gene; ch.p_aux = r_aux; ch.n = n_res;
	}
	if (ch.n) { c.n_reads = ch.n; c.chunks.push_back(std::move(ch)); }
	// UMI qualities: one string per read of this shard AFTER the exchange, in the order of its reads
	c.have_qual = false; c.qual_var = false; c.qual_len = 0; c.qual_reads = 0;
	if (r_have_qual) {
		c.have_qual = true; c.qual_len = r_qlen; c.qual_reads = c.n_reads;
		const size_t bytes = size_t(c.n_reads) * r_qlen;
		if (bytes) {
			c.umi_qual.ensure(bytes);
			HIP_CHECK(hipMemcpyAsync(c.umi_qual.p, exchanged ? x_qual.p : r_qual.p, bytes, hipMemcpyDeviceToDevice, c.stream));
		}
		if (r_qual_var) {
			c.qual_var = true;
			if (c.n_reads) {
				c.umi_qual_lens.ensure(c.n_reads);
				HIP_CHECK(hipMemcpyAsync(c.umi_qual_lens.p, exchanged ? x_qual_lens.p : r_qual_lens.p, c.n_reads, hipMemcpyDeviceToDevice, c.stream));
			}
		}
	}
	// (one shard has nobody to agree with: its pass runs in one piece and may plan the key layout from a sample like any context)
	if (world > 1) { Phase ph(this, "ingest"); c.run_ingest(); agree_on_key_fields(); global_umi_dictionary(); }
	install_umi_hooks();
	{ Phase ph(this, "pipeline"); c.run_set_initialized(); }
	merged_barcodes.clear();
	if ((c.cfg.merge_kind == DROPEST_MERGE_REAL_BARCODES || c.cfg.merge_kind == DROPEST_MERGE_POISSON_REAL) && world > 1) { Phase ph(this, "cb_merge"); cb_merge(); }
	else if (c.cfg.merge_kind != DROPEST_MERGE_NONE && world > 1) { Phase ph(this, "cb_merge"); cb_merge_free(); }
	{ Phase ph(this, "finalize"); c.run_merge_and_filter(); }
	merged_pending = world == 1 && c.cfg.merge_kind != DROPEST_MERGE_NONE;   // one shard: the context's own pairs, named when asked for
	raw_device_now = plan_raw();
	// cm_raw first when it is planned on the device: its columns are on their way to the host (own stream) while the filtered
	// candidates are gathered and ordered and cm is assembled -- every shard issues its collectives in this same order.  (Measured at
	// C2 through one shard: 12.2 ms with everything on one stream, 11.75 this way, 11.9 with the table before cm_raw -- cm's emit then
	// shares the CUs with the placing kernel.)
	wide_now[0] = wide_now[1] = false;
	if (raw_device_now) assemble_raw_device();
	build_global_table();
	assemble_matrix(true);
	if (!raw_device_now) assemble_matrix(false);
	auto wait_all = [&] {
		// this shard's columns widened (its thread takes part: what is unclaimed, then what a straggler holds), its streams drained, then everybody's
		{ Phase ph(this, "matrix:decode_wait"); finish_slots(1); finish_slots(0); }
		Phase ph(this, "matrix:wait"); HIP_CHECK(stream_wait(c.stream)); if (place_stream) HIP_CHECK(stream_wait(place_stream)); tr->barrier();
	};
	wait_all();
	if (byte_matrix) {
		// A shard that had more to list than its segment holds (very sparse columns: a small cell lists nearly every row) shows in the
		// segment counts, which every shard reads after the barrier: all take the same decision and place that matrix's columns again in
		// the 16- / 32-bit form.  The step cannot fail for the shape of the data.
		bool again = false;
		for (int k = 0; k < 2; ++k) if (lists_overflowed(mat[k])) { wide_now[k] = true; again = true; }
		if (again) {
			Phase ph(this, "matrix:overflow");
			tr->barrier();   // (nobody reopens a shared buffer while another shard still reads its counts)
			if (wide_now[1] && raw_device_now) assemble_raw_device();
			if (wide_now[0]) assemble_matrix(true);
			if (wide_now[1] && !raw_device_now) assemble_matrix(false);
			wait_all();
		}
		// (a matrix whose slots were widened inside the step needs its lists only if somebody asks for the byte form: collected then)
		Phase ph(this, "matrix:lists"); for (int k = 0; k < 2; ++k) collect_lists(mat[k]);   // (a slots step has no byte form in the buffer: nothing to collect)
	}
	c.collect_timings();
}

void dropest_shard::finish_slots(int slot) {
	using namespace dropest;
	Mat &M = mat[slot];
	if (!M.slots || !M.shipped) return;
	M.shipped = false;
	dropest_ctx &c = *ctx;
	if (c.wire_finish(c.mat[slot])) return;
	// The lists of this shard's byte form overflowed (very sparse columns: a small cell lists nearly every row).  The slots ARE the 32-bit
	// form: this shard emits its columns as 32-bit arrays and places them there directly -- its own decision, no collective (ADVICE r3:
	// the step cannot fail for the shape of the data).
	phases["matrix:overflow"].launches++;
	const bool filtered_m = slot == 0;
	c.emit_columns_device(filtered_m, false, filtered_m ? M.col_cell : raw_plan.col_cell, filtered_m ? M.col_start : raw_plan.col_start, M.local_nnz, false);
	const dropest_ctx::MatrixResult &R = c.mat[slot];
	hipLaunchKernelGGL(place_columns_kernel, dim3(M.nc), dim3(256), 0, c.stream, M.d_descr, R.d_row.p, R.d_val.p, M.d_slot_rows, M.d_slot_vals);
	HIP_CHECK(hipGetLastError());
	HIP_CHECK(stream_wait(c.stream));
}

// The byte form of a matrix whose step ended with the slots (dropest_shard_matrix_bytes asks for it): encoded from the slots, once.
void dropest_shard::encode_bytes(Mat &M) {
	using namespace dropest;
	if (M.encoded) return;
	M.enc_delta.assign(size_t(M.nnz), 0); M.enc_vals.assign(size_t(M.nnz), 0);
	M.rl_pos.clear(); M.rl_row.clear(); M.vl_pos.clear(); M.vl_val.clear();
	const u32 *cp = M.colptr32_p;
	for (uint64_t c = 0; c < M.ncols; ++c) {
		uint32_t prev1 = 0;
		for (uint32_t k = cp[c]; k < cp[c + 1]; ++k) {
			const uint32_t row = M.slot_rows[k], v = M.slot_vals[k], d = row + 1u - prev1;
			prev1 = row + 1u;
			M.enc_delta[k] = uint8_t(d >= 255u ? 255u : d); M.enc_vals[k] = uint8_t(v >= 255u ? 255u : v);
			if (d >= 255u) { M.rl_pos.push_back(k); M.rl_row.push_back(row); }
			if (v >= 255u) { M.vl_pos.push_back(k); M.vl_val.push_back(v); }
		}
	}
	M.delta8 = M.enc_delta.data(); M.vals8 = M.enc_vals.data(); M.lists_ready = true; M.encoded = true;
}

// the byte form's lists: every shard's segment of the shared buffer (written through its PCIe link, complete after the barrier)
// -> one list of a kind (a matrix whose lists overflowed on some shard was placed again without the byte form before this: step())
bool dropest_shard::lists_overflowed(const Mat &M) const {
	if (!M.bytes) return false;
	for (int p = 0; p < world; ++p) {
		const u32 *w = reinterpret_cast<const u32 *>(M.segments + M.seg_bytes * size_t(p));
		if (w[0] > M.list_cap || w[1] > M.list_cap) return true;
	}
	return false;
}

void dropest_shard::collect_lists(Mat &M) {
	using namespace dropest;
	if (!M.bytes || M.lists_ready) return;
	size_t nr = 0, nv = 0;
	for (int p = 0; p < world; ++p) {
		const u32 *w = reinterpret_cast<const u32 *>(M.segments + M.seg_bytes * size_t(p));
		if (w[0] > M.list_cap || w[1] > M.list_cap) throw InvalidError("internal: a byte-form matrix kept overflowed lists");
		nr += w[0]; nv += w[1];
	}
	M.rl_pos.resize(nr); M.rl_row.resize(nr); M.vl_pos.resize(nv); M.vl_val.resize(nv);
	size_t ar = 0, av = 0;
	for (int p = 0; p < world; ++p) {
		const u32 *w = reinterpret_cast<const u32 *>(M.segments + M.seg_bytes * size_t(p));
		const u32 *l = w + 4;
		if (w[0]) { std::memcpy(M.rl_pos.data() + ar, l, size_t(w[0]) * 4); std::memcpy(M.rl_row.data() + ar, l + M.list_cap, size_t(w[0]) * 4); ar += w[0]; }
		if (w[1]) { std::memcpy(M.vl_pos.data() + av, l + 2 * size_t(M.list_cap), size_t(w[1]) * 4); std::memcpy(M.vl_val.data() + av, l + 3 * size_t(M.list_cap), size_t(w[1]) * 4); av += w[1]; }
	}
	M.lists_ready = true;
}

// N-UMI merge across shards (umi_merge_host.h): global first occurrences of the tie candidates, offsets into the ONE rand()
// sequence the reference draws the random fills from
void dropest_shard::install_umi_hooks() {
	using namespace dropest;
	if (world == 1) { ctx->hooks.reset(); return; }
	auto h = std::make_shared<dropest_ctx::ShardHooks>();
	h->first_seen_global = [this](const std::vector<u64> &codes) {
		Phase ph(this, "umi:first_seen");
		std::vector<u64> all;
		std::vector<size_t> cnt;
		tr->gather_vec(codes, all, cnt);
		std::sort(all.begin(), all.end());
		all.erase(std::unique(all.begin(), all.end()), all.end());
		const std::vector<u32> pos = ctx->umi_first_positions(all);
		std::vector<u64> ord = global_ordinals(pos);
		std::vector<u64> every(std::max<size_t>(all.size(), 1) * size_t(world), ~0ull);
		ord.resize(std::max<size_t>(all.size(), 1), ~0ull);
		tr->gather_host(ord.data(), ord.size() * 8, every.data());
		for (size_t i = 0; i < all.size(); ++i)
			for (int p = 0; p < world; ++p) ord[i] = std::min(ord[i], every[size_t(p) * ord.size() + i]);
		std::vector<u64> out(codes.size());
		for (size_t i = 0; i < codes.size(); ++i) out[i] = ord[size_t(std::lower_bound(all.begin(), all.end(), codes[i]) - all.begin())];
		return out;
	};
	h->rng_offsets = [this](const std::vector<u32> &cell_first, const std::vector<u32> &gene, const std::vector<u32> &draws) {
		Phase ph(this, "umi:rng_offsets");
		struct Key { u64 first_global; u32 gene, draws, rank, idx; };
		const std::vector<u64> ord = global_ordinals(cell_first);
		std::vector<Key> mine, all;
		for (size_t g = 0; g < draws.size(); ++g) if (draws[g]) mine.push_back(Key{ord[g], gene[g], draws[g], u32(rank), u32(g)});
		std::vector<size_t> cnt;
		tr->gather_vec(mine, all, cnt);
		std::sort(all.begin(), all.end(), [](const Key &a, const Key &b) { return a.first_global != b.first_global ? a.first_global < b.first_global : a.gene < b.gene; });
		std::vector<u64> out(draws.size(), 0);
		u64 at = 0;
		for (const Key &k : all) { if (int(k.rank) == rank) out[k.idx] = at; at += k.draws; }
		// groups without draws: wherever the sequence stands when they are reached (skip_to never goes back)
		return out;
	};
	h->globalize_umi_first = [this](u32 *d_table, size_t n) {
		Phase ph(this, "umi:first_table");
		dropest_ctx &c = *ctx;
		if (n >= 0xFFFFFFF0ull) throw UnsupportedError("UMI table too large");
		{   // the tables are all-gathered with ONE size: a shard that laid its UMI field out differently must not get this far
			uint64_t mine_n = n;
			std::vector<uint64_t> every(static_cast<size_t>(world));
			tr->gather_host(&mine_n, 8, every.data());
			for (uint64_t x : every) if (x != n) throw InvalidError("internal: the shards disagree on the size of the UMI first-occurrence table (" + std::to_string(x) + " vs " + std::to_string(n) + ")");
		}
		DevBuf<u64> mine, all;
		mine.alloc(n); all.alloc(n * size_t(world));
		hipLaunchKernelGGL(ordinals_kernel, dim3(u32((n + 255) / 256)), dim3(256), 0, c.stream, ordinal_map(), d_table, u32(n), mine.p);
		HIP_CHECK(hipGetLastError());
		std::vector<size_t> off(static_cast<size_t>(world)), bytes(static_cast<size_t>(world), n * 8);
		for (int p = 0; p < world; ++p) off[size_t(p)] = size_t(p) * n * 8;
		tr->gather_dev(mine.p, all.p, off.data(), bytes.data(), c.stream);
		hipLaunchKernelGGL(min_rows_kernel, dim3(u32(std::min<size_t>((n + 255) / 256, 4096))), dim3(256), 0, c.stream, all.p, u32(world), uint64_t(n), mine.p);
		HIP_CHECK(hipGetLastError());
		// ranks: sort (ordinal, code) on the bits an ordinal can have; codes nobody saw (all ones) keep 0xFFFFFFFF
		uint64_t reach = 1;
		{
			uint64_t mine2[2] = {first_ordinal, n_res};
			std::vector<uint64_t> every(size_t(world) * 2);
			tr->gather_host(mine2, sizeof(mine2), every.data());
			for (int p = 0; p < world; ++p) reach = std::max<uint64_t>(reach, every[size_t(p) * 2] + every[size_t(p) * 2 + 1]);
		}
		const u64 mask = reach >= (1ull << 63) ? ~0ull : ((1ull << bit_length(reach)) - 1ull);
		DevBuf<u64> k_alt; DevBuf<u32> v, v_alt;
		k_alt.alloc(n); v.alloc(n); v_alt.alloc(n);
		hipLaunchKernelGGL(iota_kernel, dim3(u32((n + 255) / 256)), dim3(256), 0, c.stream, v.p, u32(n));
		u64 *k = mine.p, *ka = k_alt.p; u32 *vv = v.p, *va = v_alt.p;
		c.radix_sort(k, vv, ka, va, u32(n), mask);
		hipLaunchKernelGGL(ranks_to_table_kernel, dim3(u32((n + 255) / 256)), dim3(256), 0, c.stream, k, vv, u32(n), d_table);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(stream_wait(c.stream));   // the buffers above die with this scope
	};
	ctx->hooks = h;
}

// ---- C-ABI of the sharded runner (include/dropest_amd.h) ---------------------------------------------------------------
static dropest_shard *make_shard(const dropest_cfg *cfg, int rank, int world) {
	std::unique_ptr<dropest_shard> s(new dropest_shard());
	s->ctx.reset(new dropest_ctx());
	s->ctx->init_from_cfg(*cfg);
	s->rank = rank; s->world = world;
	if (const char *e = getenv("DROPEST_SHARD_TRACE")) s->trace = atoi(e) != 0;
	if (const char *e = getenv("DROPEST_SHARD_FORCE_EXCHANGE")) s->force_exchange = atoi(e) != 0;
	return s.release();
}

extern "C" {

dropest_status dropest_shard_unique_id(uint8_t id[128]) {
	return guarded([&] {
		if (!id) throw InvalidError("null argument");
		const dropest::RcclApi &api = dropest::RcclApi::get();
		ncclUniqueId u;
		api.check(api.GetUniqueId(&u), "ncclGetUniqueId");
		std::memcpy(id, &u, 128);
	});
}

dropest_status dropest_shard_create(const dropest_cfg *cfg, int32_t rank, int32_t world, const uint8_t id[128], dropest_shard **out) {
	if (!cfg || !out) { g_last_error = "null argument"; return DROPEST_ERR_INVALID; }
	*out = nullptr;
	return guarded([&] {
		if (world < 1 || world > 64 || rank < 0 || rank >= world) throw InvalidError("rank / world out of range (1..64 shards)");
		if (!id) throw InvalidError("the RCCL unique id of the run is needed (dropest_shard_unique_id on rank 0, distributed by the launcher)");
		std::unique_ptr<dropest_shard> s(make_shard(cfg, rank, world));
		s->tr.reset(new dropest::RcclTransport(rank, world, id, s->ctx->stream));
		*out = s.release();
	});
}

dropest_status dropest_shard_group_create(const dropest_cfg *cfg, int32_t n, const int32_t *devices, dropest_shard **out) {
	if (!cfg || !out || !devices) { g_last_error = "null argument"; return DROPEST_ERR_INVALID; }
	return guarded([&] {
		if (n < 1 || n > 64) throw InvalidError("1..64 shards");
		auto hub = std::make_shared<dropest::LocalHub>(n);
		std::vector<std::unique_ptr<dropest_shard>> made;
		for (int i = 0; i < n; ++i) {
			dropest_cfg c = *cfg;
			c.device = devices[i];
			made.emplace_back(make_shard(&c, i, n));
			made.back()->tr.reset(new dropest::LocalTransport(hub, i));
		}
		// shards on different devices copy to each other directly (xGMI peer access)
		for (int i = 0; i < n; ++i)
			for (int j = 0; j < n; ++j)
				if (devices[i] != devices[j]) {
					int can = 0;
					HIP_CHECK(hipDeviceCanAccessPeer(&can, devices[i], devices[j]));
					if (!can) continue;
					HIP_CHECK(hipSetDevice(devices[i]));
					const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
					if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(e);
					(void)hipGetLastError();
				}
		for (int i = 0; i < n; ++i) out[i] = made[size_t(i)].release();
	});
}

// Keys wider than 64 bits.  Of the three key fields only the cell field depends on how much of the stream a context sees:
// split by barcode owner over `parts` shards ON THE CONTEXT'S OWN DEVICE, every shard numbers 1/parts of the cells.  The
// shards read the context's resident reads in place (contiguous ordinal ranges, ascending with the shard).
dropest_status dropest_ctx_split(dropest_ctx *ctx, int32_t parts, dropest_shard **out) {
	if (!ctx || !out) { g_last_error = "null argument"; return DROPEST_ERR_INVALID; }
	return guarded([&] {
		if (parts < 2 || parts > 64) throw InvalidError("2..64 parts");
		HIP_CHECK(hipSetDevice(ctx->cfg.device));
		if (ctx->n_reads == 0) throw InvalidError("no reads to split");
		if (ctx->have_qual && ctx->qual_reads != ctx->n_reads) throw InvalidError("UMI qualities were given for another number of reads than the context holds");
		ctx->concat_chunks();
		ctx->free_results();
		ctx->release_tables();   // the shards need the room; the reads stay
		auto hub = std::make_shared<dropest::LocalHub>(parts);
		std::vector<std::unique_ptr<dropest_shard>> made;
		const uint64_t n = ctx->n_reads;
		for (int i = 0; i < parts; ++i) {
			made.emplace_back(make_shard(&ctx->cfg, i, parts));
			dropest_shard &s = *made.back();
			s.tr.reset(new dropest::LocalTransport(hub, i));
			s.ctx->side = ctx->side;
			const uint64_t a = n * uint64_t(i) / uint64_t(parts), b = n * uint64_t(i + 1) / uint64_t(parts);
			s.r_cb = ctx->d_cb + a; s.r_umi = ctx->d_umi + a; s.r_gene = ctx->d_gene + a; s.r_aux = ctx->d_aux + a;
			s.n_res = b - a; s.first_ordinal = a;
			if (ctx->have_qual) {   // the UMI quality strings of the shard's range (and their lengths) follow their reads: a copy on the device
				s.r_have_qual = true; s.r_qlen = ctx->qual_len; s.r_qual_reads = b - a; s.r_qual_var = ctx->qual_var;
				const size_t bytes = size_t(b - a) * ctx->qual_len;
				if (bytes) { s.r_qual.ensure(bytes); HIP_CHECK(hipMemcpyAsync(s.r_qual.p, ctx->umi_qual.p + size_t(a) * ctx->qual_len, bytes, hipMemcpyDeviceToDevice, ctx->stream)); }
				if (ctx->qual_var && b > a) { s.r_qual_lens.ensure(size_t(b - a)); HIP_CHECK(hipMemcpyAsync(s.r_qual_lens.p, ctx->umi_qual_lens.p + a, size_t(b - a), hipMemcpyDeviceToDevice, ctx->stream)); }
			}
		}
		if (ctx->have_qual) HIP_CHECK(stream_wait(ctx->stream));
		for (int i = 0; i < parts; ++i) out[i] = made[size_t(i)].release();
	});
}

dropest_status dropest_key_width(dropest_ctx *ctx, uint32_t *cell_bits, uint32_t *gene_bits, uint32_t *umi_bits) {
	return guarded([&] {
		if (!ctx) throw InvalidError("null context");
		if (cell_bits) *cell_bits = ctx->wanted_bits[0];
		if (gene_bits) *gene_bits = ctx->wanted_bits[1];
		if (umi_bits) *umi_bits = ctx->wanted_bits[2];
	});
}

void dropest_shard_destroy(dropest_shard *s) {
	if (!s) return;
	(void)hipSetDevice(s->ctx->cfg.device);
	delete s;
}

dropest_ctx *dropest_shard_ctx(dropest_shard *s) { return s ? s->ctx.get() : nullptr; }

dropest_status dropest_shard_set_reads_device(dropest_shard *s, const uint64_t *d_cb, const uint64_t *d_umi, const uint32_t *d_gene,
                                              const uint32_t *d_aux, uint64_t n, uint64_t first_ordinal) {
	return guarded([&] {
		if (!s) throw InvalidError("null shard");
		if (n && (!d_cb || !d_umi || !d_gene || !d_aux)) throw InvalidError("null read array");
		if (n >= 0xFFFFFFFEull) throw UnsupportedError("more than 2^32-2 reads per shard");
		s->r_cb = reinterpret_cast<const dropest::u64 *>(d_cb); s->r_umi = reinterpret_cast<const dropest::u64 *>(d_umi);
		s->r_gene = d_gene; s->r_aux = d_aux; s->n_res = n; s->first_ordinal = first_ordinal;
		s->pushed.clear();
	});
}

dropest_status dropest_shard_push_reads(dropest_shard *s, const uint64_t *cb, const uint64_t *umi, const uint32_t *gene, const uint32_t *aux,
                                        uint64_t n, uint64_t first_ordinal) {
	return guarded([&] {
		if (!s) throw InvalidError("null shard");
		if (!n) return;
		if (!cb || !umi || !gene || !aux) throw InvalidError("null read array");
		if (s->pushed.n + n >= 0xFFFFFFFEull) throw UnsupportedError("more than 2^32-2 reads per shard");
		HIP_CHECK(hipSetDevice(s->ctx->cfg.device));
		if (s->pushed.n == 0) s->first_ordinal = first_ordinal;
		else if (first_ordinal != s->first_ordinal + s->pushed.n) throw InvalidError("the batches pushed to a shard must continue its ordinal range without a gap");
		s->pushed.push(cb, umi, gene, aux, size_t(n));
	});
}

dropest_status dropest_shard_set_umi_qualities(dropest_shard *s, const uint8_t *qualities, uint32_t quality_length, uint64_t n_reads) {
	return guarded([&] {
		if (!s) throw InvalidError("null shard");
		if (quality_length > 255) throw UnsupportedError("UMI quality strings longer than 255");
		if (n_reads && quality_length && !qualities) throw InvalidError("null quality array");
		HIP_CHECK(hipSetDevice(s->ctx->cfg.device));
		s->r_have_qual = true; s->r_qlen = quality_length; s->r_qual_reads = n_reads; s->r_qual_var = false;
		const size_t bytes = size_t(n_reads) * quality_length;
		if (bytes) {
			s->r_qual.ensure(bytes);
			HIP_CHECK(hipMemcpy(s->r_qual.p, qualities, bytes, hipMemcpyHostToDevice));
		}
	});
}

dropest_status dropest_shard_set_umi_qualities_var(dropest_shard *s, const uint8_t *qualities, uint32_t row_bytes, const uint8_t *lengths, uint64_t n_reads) {
	const dropest_status st = dropest_shard_set_umi_qualities(s, qualities, row_bytes, n_reads);
	if (st != DROPEST_OK) return st;
	return guarded([&] {
		if (n_reads && !lengths) throw InvalidError("null length array");
		for (uint64_t i = 0; i < n_reads; ++i)
			if (lengths[i] > row_bytes) throw InvalidError("a quality length beyond the row width (" + std::to_string(row_bytes) + ")");
		if (n_reads) {
			s->r_qual_lens.ensure(n_reads);
			HIP_CHECK(hipMemcpy(s->r_qual_lens.p, lengths, n_reads, hipMemcpyHostToDevice));
		}
		s->r_qual_var = true;
	});
}

dropest_status dropest_shard_step(dropest_shard *s) {
	return guarded([&] {
		if (!s) throw InvalidError("null shard");
		try { s->step(); }
		catch (...) {   // the other members of an in-process group must not wait for this one forever
			if (auto *lt = dynamic_cast<dropest::LocalTransport *>(s->tr.get())) lt->hub->fail();
			if (auto *rt = dynamic_cast<dropest::RcclTransport *>(s->tr.get())) if (rt->mailbox) rt->mailbox->fail();   // (the peers' next host collective throws)
			throw;
		}
	});
}

dropest_status dropest_shard_group_step(dropest_shard *const *shards, int32_t n) {
	return guarded([&] {
		if (!shards || n < 1) throw InvalidError("null argument");
		std::vector<dropest_status> rc(size_t(n), DROPEST_OK);
		std::vector<std::string> msg(static_cast<size_t>(n));
		std::vector<std::thread> pool;
		for (int i = 0; i < n; ++i)
			pool.emplace_back([&, i] { rc[size_t(i)] = dropest_shard_step(shards[i]); if (rc[size_t(i)] != DROPEST_OK) msg[size_t(i)] = dropest_last_error(); });
		for (auto &t : pool) t.join();
		for (int i = 0; i < n; ++i)
			if (rc[size_t(i)] != DROPEST_OK && msg[size_t(i)].find("another shard of the group failed") == std::string::npos) {
				if (rc[size_t(i)] == DROPEST_ERR_UNSUPPORTED) throw UnsupportedError("shard " + std::to_string(i) + ": " + msg[size_t(i)]);
				throw InvalidError("shard " + std::to_string(i) + ": " + msg[size_t(i)]);
			}
		for (int i = 0; i < n; ++i) if (rc[size_t(i)] != DROPEST_OK) throw InvalidError("shard " + std::to_string(i) + ": " + msg[size_t(i)]);
	});
}

dropest_status dropest_shard_matrix(dropest_shard *s, int filtered, uint64_t *ncols, uint64_t *nnz, const uint64_t **colptr,
                                    const uint32_t **rowidx, const uint32_t **values, const uint64_t **col_barcodes) {
	return guarded([&] {
		if (!s || !ncols || !nnz) throw InvalidError("null argument");
		dropest_shard::Mat &M = s->mat[filtered ? 0 : 1];
		*ncols = M.ncols; *nnz = M.nnz;
		if (colptr) *colptr = reinterpret_cast<const uint64_t *>(M.colptr_p ? M.colptr_p : M.colptr.data());
		if (M.bytes && (rowidx || values) && !M.widened) {    // the pass produced the byte form: decoded here, once, on host threads
			s->collect_lists(M);
			M.wide_rows.resize(M.nnz); M.wide_vals.resize(M.nnz);
			dropest_matrix_bytes mb{M.ncols, M.nnz, M.colptr32_p, M.delta8, M.vals8, M.rl_pos.size(), M.rl_pos.data(), M.rl_row.data(),
			                        M.vl_pos.size(), M.vl_pos.data(), M.vl_val.data()};
			if (M.nnz && dropest_matrix_bytes_widen(&mb, M.wide_rows.data(), M.wide_vals.data()) != DROPEST_OK) throw InvalidError(dropest_last_error());
			M.rows = M.wide_rows.data(); M.vals = M.wide_vals.data(); M.widened = true;
		}
		if (M.narrow && (rowidx || values) && !M.widened) {   // the pass produced the 16-bit form: the 32-bit view is made here, once
			M.wide_rows.resize(M.nnz); M.wide_vals.resize(M.nnz);
			dropest::parallel_ranges(M.nnz, [&](size_t b, size_t e, unsigned) { for (size_t i = b; i < e; ++i) { M.wide_rows[i] = M.rows16[i]; M.wide_vals[i] = M.vals16[i]; } }, 1 << 20, 16);
			for (size_t k = 0; k < M.ovf_pos.size(); ++k) M.wide_vals[M.ovf_pos[k]] = M.ovf_val[k];
			M.rows = M.wide_rows.data(); M.vals = M.wide_vals.data(); M.widened = true;
		}
		if (rowidx) *rowidx = M.rows;
		if (values) *values = M.vals;
		if (col_barcodes) *col_barcodes = reinterpret_cast<const uint64_t *>(M.barcode_p ? M.barcode_p : M.col_barcode.data());
	});
}

dropest_status dropest_shard_matrix_form(dropest_shard *s, int filtered, int32_t *form) {
	return guarded([&] {
		if (!s || !form) throw InvalidError("null argument");
		const dropest_shard::Mat &M = s->mat[filtered ? 0 : 1];
		*form = M.slots ? 3 : (M.bytes ? 2 : (M.narrow ? 1 : 0));
	});
}

dropest_status dropest_shard_matrix_narrow(dropest_shard *s, int filtered, uint64_t *ncols, uint64_t *nnz, const uint64_t **colptr,
                                           const uint16_t **rowidx, const uint16_t **values, const uint64_t **col_barcodes,
                                           uint64_t *n_overflow, const uint64_t **overflow_pos, const uint32_t **overflow_val) {
	return guarded([&] {
		if (!s || !ncols || !nnz || !rowidx || !values || !n_overflow || !overflow_pos || !overflow_val) throw InvalidError("null argument");
		const dropest_shard::Mat &M = s->mat[filtered ? 0 : 1];
		if (!M.narrow && M.nnz) throw UnsupportedError("the last step did not produce the 16-bit matrices (the option byte_matrix is on, gene ids beyond 65535, or the option narrow_matrix is off)");
		*ncols = M.ncols; *nnz = M.nnz;
		if (colptr) *colptr = reinterpret_cast<const uint64_t *>(M.colptr_p ? M.colptr_p : M.colptr.data());
		*rowidx = M.rows16; *values = M.vals16;
		if (col_barcodes) *col_barcodes = reinterpret_cast<const uint64_t *>(M.barcode_p ? M.barcode_p : M.col_barcode.data());
		*n_overflow = M.ovf_pos.size();
		*overflow_pos = reinterpret_cast<const uint64_t *>(M.ovf_pos.data()); *overflow_val = M.ovf_val.data();
	});
}

dropest_status dropest_shard_matrix_bytes(dropest_shard *s, int filtered, dropest_matrix_bytes *out, const uint64_t **col_barcodes) {
	return guarded([&] {
		if (!s || !out) throw InvalidError("null argument");
		dropest_shard::Mat &M = s->mat[filtered ? 0 : 1];
		if (!M.bytes && !M.slots && M.nnz) throw UnsupportedError("the last step did not produce the byte-form matrices (the shard option byte_matrix is off)");
		if (M.slots) s->encode_bytes(M); else s->collect_lists(M);
		static const uint32_t zero = 0;
		*out = dropest_matrix_bytes{M.ncols, M.nnz, (M.bytes || M.slots) ? M.colptr32_p : &zero, M.delta8, M.vals8, M.rl_pos.size(), M.rl_pos.data(), M.rl_row.data(),
		                            M.vl_pos.size(), M.vl_pos.data(), M.vl_val.data()};
		if (col_barcodes) *col_barcodes = reinterpret_cast<const uint64_t *>(M.barcode_p ? M.barcode_p : M.col_barcode.data());
	});
}

dropest_status dropest_shard_merged_barcodes(dropest_shard *s, uint64_t *n, uint64_t *source, uint64_t *target) {
	return guarded([&] {
		if (!s || !n) throw InvalidError("null argument");
		s->name_merged_pairs();
		*n = s->merged_barcodes.size();
		if (source && target) for (size_t i = 0; i < s->merged_barcodes.size(); ++i) { source[i] = s->merged_barcodes[i].first; target[i] = s->merged_barcodes[i].second; }
	});
}

// test hook (tests/test_host_mailbox.py: processes on the CPU): `rounds` gathers of `bytes` patterned bytes per rank, every one checked;
// returns 0, or the round (1-based) in which a slot did not hold what its rank wrote
int dropest_test_host_mailbox(uint64_t token, int32_t rank, int32_t world, uint32_t rounds, uint64_t bytes) {
	try {
		dropest::HostMailbox mb(token, rank, world);
		std::vector<unsigned char> mine(bytes), all(size_t(bytes) * size_t(world));
		for (uint32_t r = 0; r < rounds; ++r) {
			const size_t n = r % 3 == 2 ? size_t(bytes) / 2 : size_t(bytes);          // sizes change between rounds (the same on every rank)
			for (size_t i = 0; i < n; ++i) mine[i] = (unsigned char)(rank * 131 + r * 7 + i * 13);
			mb.gather(mine.data(), n, all.data());
			for (int p = 0; p < world; ++p)
				for (size_t i = 0; i < n; ++i) if (all[size_t(p) * n + i] != (unsigned char)(p * 131 + r * 7 + i * 13)) return int(r + 1);
		}
		mb.barrier();
		return 0;
	} catch (const std::exception &e) { g_last_error = e.what(); return -1; }
}

dropest_status dropest_shard_phase_stats(dropest_shard *s, uint32_t *n, dropest_kernel_stat *out) {
	return guarded([&] {
		if (!s || !n) throw InvalidError("null argument");
		uint32_t i = 0;
		for (auto &kv : s->phases) {
			if (out) { out[i].name = kv.first.c_str(); out[i].launches = kv.second.launches; out[i].ms = kv.second.ms; out[i].bytes = kv.second.bytes; }
			++i;
		}
		*n = i;
	});
}

// Column order of a global matrix from the all-gathered table of real cells -- the host path of dropest_shard::order_rows
// + the selection of assemble_matrix, without a device (the world-size-2 gloo tests on CPU drive it).
dropest_status dropest_plan_columns(uint64_t n, const uint64_t *barcode, const uint64_t *first_global, const uint32_t *n_genes,
                                    const uint32_t *req_genes, const uint32_t *req_umis, const int32_t *total_umis, int filtered,
                                    uint32_t min_genes_after_merge, int32_t max_cells, const char *const *side_strings, uint64_t n_side,
                                    uint64_t *n_cols, uint32_t *order) {
	return guarded([&] {
		if (!n_cols || (n && (!barcode || !first_global || !n_genes || !req_genes || !req_umis || !total_umis))) throw InvalidError("null argument");
		if (n >= 0xFFFFFFF0ull) throw UnsupportedError("too many cells");
		struct Row { dropest::u64 barcode, first_global; dropest::u32 req_genes, req_umis; int32_t total_umis; };
		std::vector<std::string> side;
		for (uint64_t i = 0; i < n_side; ++i) side.emplace_back(side_strings[i]);
		std::vector<Row> rows(n);
		std::vector<dropest::u32> sel;
		for (uint64_t i = 0; i < n; ++i) {
			rows[i] = Row{barcode[i], first_global[i], req_genes[i], req_umis[i], total_umis[i]};
			if (!filtered || req_genes[i] >= min_genes_after_merge) sel.push_back(dropest::u32(i));
		}
		if (filtered) {
			std::sort(sel.begin(), sel.end(), [&](dropest::u32 x, dropest::u32 y) { return compare_cells_rows(rows[x], rows[y], side); });
			if (max_cells > 0 && size_t(max_cells) < sel.size()) sel.erase(sel.begin(), sel.end() - max_cells);
		} else {
			std::sort(sel.begin(), sel.end(), [&](dropest::u32 x, dropest::u32 y) { return rows[x].first_global < rows[y].first_global; });
		}
		*n_cols = sel.size();
		if (order) std::copy(sel.begin(), sel.end(), order);
	});
}

dropest_status dropest_shard_set_option(dropest_shard *s, const char *key, int64_t value) {
	return guarded([&] {
		if (!s || !key) throw InvalidError("null argument");
		const std::string k(key);
		if (k == "trace") s->trace = value != 0;
		else if (k == "force_exchange") s->force_exchange = value != 0;
		else if (k == "reset_phase_stats") s->phases.clear();
		else if (k == "narrow_matrix") s->narrow_matrix = value != 0;
		else if (k == "byte_matrix") s->byte_matrix = value != 0;
		else if (k == "slots_matrix") s->slots_matrix = value != 0;
		else if (k == "raw_on_device") s->raw_on_device = value != 0;
		else if (k == "byte_list_cap") s->byte_list_cap = value > 0 ? uint64_t((value + 15) & ~15ll) : 0;
		else if (k == "packed_exchange") s->allow_packed = value != 0;
		else if (k == "exact_widths") s->exact_widths = value != 0;
		else if (k == "exchange_chunks") s->exchange_chunks = int(std::max<int64_t>(0, std::min<int64_t>(value, 16)));
		else throw InvalidError("unknown shard option: " + k);
	});
}

}  // extern "C"
