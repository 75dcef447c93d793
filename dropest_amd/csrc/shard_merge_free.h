// shard_merge_free.h -- the barcode merges WITHOUT a whitelist when the cells are sharded over several GPUs (included by shard_run.h):
// SimpleMergeStrategy (Estimation/Merge/SimpleMergeStrategy.cpp:16-108), PoissonSimpleMergeStrategy
// (PoissonSimpleMergeStrategy.cpp:15-43), MergeAllMergeStrategy (MergeAllMergeStrategy.h:16-50).
//
// One container in the reference; here the candidates of a cell are the cells that share UMI-genes with it, wherever they live:
//   index      the (UMI-gene -> cells) inverted index of SimpleMergeStrategy::init (:88-102) is sharded by hash(UMI-gene): every
//              shard sends the (UMI-gene, global cell) records of its filtered cells to the shard that owns the UMI-gene -- one
//              all-to-all(v) of 8-byte records
//   pairs      the index shard sorts its records; a run = the cells sharing one UMI-gene; ordered pairs (base, other at least as
//              large) of every run, sorted and run-length encoded: PARTIAL counts of common UMI-genes per pair
//   route      partial counts travel to the shard that owns the base (a second all-to-all(v)), where they are added per pair
//   decide     the owner of the base decides exactly as one GPU does (simple_merge.h: the same functions on the same numbers)
//   replay     bases whose answer depends on the iteration order of the reference's unordered containers are replayed with those
//              containers: the members of their UMI-genes come back from the index shards, cell ids are the GLOBAL first-seen ranks
//              (counted over all shards), UMI order is the global first occurrence
//   apply / finish   shared with the whitelist merge (shard_run.h: merge_apply, merge_gather_rows, merge_finish)
// merge_type = all needs no index: every shard compares ITS cells with the barcodes of all filtered cells (all-gathered).
#pragma once

void dropest_ctx::shard_merge_begin_free() {
	if (!initialized) throw InvalidError("You must initialize container");
	if (merged) throw InvalidError("merge_and_filter was already run");
	shard.reset(new ShardMerge());
}

namespace {

// records of one shard's index: destination shard in the top bits (the sort groups them by destination), then UMI-gene, then cell
struct FreeKeyArgs {
	const unsigned long long *mol_key; uint32_t n_mol;
	int umi_bits, gene_bits, cell_bits_g, dest_shift; unsigned long long gene_none; uint32_t world;
	const uint32_t *g_of;             // [n_cells] place of the local cell in the global list, 0xFFFFFFFF = takes no part
	unsigned long long *keys; uint32_t *n_valid;
};
__global__ __launch_bounds__(256) void free_index_keys_kernel(FreeKeyArgs a) {
	const uint32_t stride = gridDim.x * 256;
	uint32_t valid = 0;
	const int low_bits = a.umi_bits + a.gene_bits;
	const unsigned long long low_mask = (1ull << low_bits) - 1ull;
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < a.n_mol; i += stride) {
		const unsigned long long k = a.mol_key[i];
		const uint32_t g = a.g_of[uint32_t(k >> low_bits)];
		const bool ok = g != 0xFFFFFFFFu && ((k >> a.umi_bits) & a.gene_none) != a.gene_none;
		const unsigned long long low = k & low_mask;
		const unsigned long long dest = dropest::mix64(low) % a.world;
		a.keys[i] = ok ? ((dest << a.dest_shift) | (low << a.cell_bits_g) | g) : ~0ull;
		valid += ok;
	}
	valid = uint32_t(dropest::wave_reduce_add_u64(valid));
	if (dropest::lane_id() == 0 && valid) atomicAdd(a.n_valid, valid);
}
// first position of every bound in a sorted array (bounds ascending; n_bounds small)
__global__ void lower_bounds_kernel(const unsigned long long *__restrict__ keys, uint32_t n, const unsigned long long *__restrict__ bound, uint32_t n_bounds,
                                    uint32_t *__restrict__ out) {
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n_bounds) return;
	uint32_t l = 0, h = n;
	while (l < h) { const uint32_t m = (l + h) >> 1; if (keys[m] < bound[q]) l = m + 1; else h = m; }
	out[q] = l;
}
__global__ __launch_bounds__(256) void and_mask_kernel(unsigned long long *k, uint32_t n, unsigned long long mask) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) k[i] &= mask;
}
// runs of equal keys with the SUM of a weight (partial pair counts added where the base lives)
struct WeightedRuns {
	static constexpr int ITEMS = 8;
	static constexpr bool PACKED = false;
	static constexpr bool DIRECT = false;
	__device__ uint32_t direct_index(unsigned long long) const { return 0; }
	static constexpr int NV = 1;
	static constexpr unsigned OR_MASK = 0;
	const unsigned long long *keys; const uint32_t *weight;
	unsigned long long *run_key;
	uint32_t *out[NV];
	__device__ unsigned long long seg_key(uint32_t i) const { return keys[i]; }
	__device__ void load(uint32_t i, uint32_t (&v)[NV]) const { v[0] = weight[i]; }
	__device__ void write_head(uint32_t o, uint32_t, unsigned long long k) const { run_key[o] = k; }
};
// global ordinal of the first read of every local cell (ascending with the cell id)
__global__ __launch_bounds__(256) void cell_first_global_kernel(dropest::OrdinalMap m, const uint32_t *__restrict__ cell_first, uint32_t n, unsigned long long *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) out[i] = dropest::to_global_ordinal(m, cell_first[i]);
}
// number of entries of a sorted array below every query
__global__ __launch_bounds__(256) void count_below_kernel(const unsigned long long *__restrict__ sorted, uint32_t n, const unsigned long long *__restrict__ q, uint32_t nq,
                                                          uint32_t *__restrict__ out) {
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= nq) return;
	uint32_t l = 0, h = n;
	while (l < h) { const uint32_t m = (l + h) >> 1; if (sorted[m] < q[i]) l = m + 1; else h = m; }
	out[i] = l;
}

}  // namespace

void dropest_shard::cb_merge_free() {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	MergeWorld W;
	merge_gather_cells(W);
	const auto &Gm = W.Gm;
	const u32 nG = W.nG, lo = W.lo, hi = W.hi;
	c.shard_merge_begin_free();
	free_rows.reset();
	const std::vector<u32> order = merge_order(W);          // the reference's filtered order over ALL shards' cells
	std::vector<u32> pos_of(nG);
	for (u32 f = 0; f < nG; ++f) pos_of[order[f]] = f;
	std::vector<int64_t> my_tgt(hi - lo);
	for (u32 i = lo; i < hi; ++i) my_tgt[i - lo] = int64_t(i);   // no candidate: the cell itself

	if (c.cfg.merge_kind == DROPEST_MERGE_ALL) {
		Phase ph(this, "cbm:merge_all");
		std::vector<u64> code(nG); std::vector<int32_t> umis(nG);
		for (u32 f = 0; f < nG; ++f) { code[f] = Gm[order[f]].barcode; umis[f] = Gm[order[f]].total_umis; }
		std::vector<u32> bases(hi - lo), target_pos;
		for (u32 i = lo; i < hi; ++i) bases[i - lo] = pos_of[i];
		if (nG) c.merge_all_targets(code, umis, [&](u32 f) { return decode_code(Gm[order[f]].barcode, c.side); }, &bases, target_pos);
		for (u32 i = lo; i < hi && nG; ++i) if (target_pos[i - lo] != 0xFFFFFFFFu) my_tgt[i - lo] = int64_t(order[target_pos[i - lo]]);
	} else
		free_simple_targets(W, order, pos_of, my_tgt);

	MergeApplied A;
	merge_apply(W, my_tgt, order, A);
	if (free_rows) {   // PoissonSimple: the rows of every base with a neighbour were gathered for the estimator
		merge_finish(W, A, *free_rows);
		free_rows.reset();
		return;
	}
	// rows of my cells whose molecules end up on another shard
	{
		Phase ph(this, "cbm:export");
		dropest_ctx::ShardMerge &M = *c.shard;
		std::vector<u32> cells;
		M.listed_g.clear();
		for (u32 i = lo; i < hi; ++i) {
			const u32 t = A.final_t[i];
			if (t == i || (t >= lo && t < hi)) continue;
			cells.push_back(Gm[i].local_id); M.listed_g.push_back(i);
		}
		c.shard_export_rows(cells);
	}
	TravelRows T;
	merge_gather_rows(W, T);
	merge_finish(W, A, T);
}

// Simple / PoissonSimple: targets of my cells [lo, hi) (places in the global list)
void dropest_shard::free_simple_targets(const MergeWorld &W, const std::vector<u32> &order, const std::vector<u32> &pos_of, std::vector<int64_t> &my_tgt) {
	using namespace dropest;
	(void)order;
	dropest_ctx &c = *ctx;
	const auto &Gm = W.Gm;
	const u32 nG = W.nG, lo = W.lo, hi = W.hi;
	if (nG == 0) return;
	const bool poisson = c.cfg.merge_kind == DROPEST_MERGE_POISSON_SIMPLE;
	const int low_bits = c.layout.umi_bits + c.layout.gene_bits;
	const int gb = std::max(1, bit_length(uint64_t(nG - 1)));
	const int dest_bits = bit_length(uint64_t(world));       // `world` itself is a bound: the end of the last block
	const int dest_shift = low_bits + gb;
	if (dest_shift + dest_bits > 63) throw UnsupportedError("sharded simple merge: UMI-gene + cell + shard do not fit 63 bits");
	const size_t W_ = size_t(world);

	// the cells of all shards on the device: sizes (Cell::size()) and barcodes by place in the global list
	DevBuf<u64> d_gcb; DevBuf<u32> d_gng;
	d_gcb.alloc(nG); d_gng.alloc(nG);
	{
		std::vector<u64> bc(nG); std::vector<u32> ng(nG);
		for (u32 g = 0; g < nG; ++g) { bc[g] = Gm[g].barcode; ng[g] = Gm[g].n_genes; }
		HIP_CHECK(hipMemcpyAsync(d_gcb.p, bc.data(), size_t(nG) * 8, hipMemcpyHostToDevice, c.stream));
		HIP_CHECK(hipMemcpyAsync(d_gng.p, ng.data(), size_t(nG) * 4, hipMemcpyHostToDevice, c.stream));
		HIP_CHECK(stream_wait(c.stream));
	}
	auto bounds_of = [&](const u64 *d_sorted, u32 n, const std::vector<u64> &bound) {   // first position of every bound
		std::vector<u32> out(bound.size(), 0);
		if (!n) return out;
		DevBuf<u64> d_b; DevBuf<u32> d_o;
		d_b.alloc(bound.size()); d_o.alloc(bound.size());
		HIP_CHECK(hipMemcpyAsync(d_b.p, bound.data(), bound.size() * 8, hipMemcpyHostToDevice, c.stream));
		hipLaunchKernelGGL(lower_bounds_kernel, dim3(div_up(u32(bound.size()), 64)), dim3(64), 0, c.stream, reinterpret_cast<const unsigned long long *>(d_sorted), n,
		                   reinterpret_cast<const unsigned long long *>(d_b.p), u32(bound.size()), d_o.p);
		HIP_CHECK(hipGetLastError());
		c.fetch(out.data(), d_o.p, bound.size() * 4);
		return out;
	};
	auto counts_everywhere = [&](const std::vector<uint64_t> &send, std::vector<uint64_t> &recv) {   // COLLECTIVE: what everybody sends me
		std::vector<uint64_t> all(W_ * W_);
		tr->gather_host(send.data(), W_ * 8, all.data());
		recv.assign(W_, 0);
		for (int p = 0; p < world; ++p) recv[size_t(p)] = all[size_t(p) * W_ + size_t(rank)];
	};

	// A. index records of my filtered cells, grouped by the shard that owns their UMI-gene
	DevBuf<u64> idx_a, idx_b; DevBuf<u32> idv_a, idv_b;
	u64 *ik = nullptr; u32 n_idx = 0;
	{
		Phase ph(this, "cbm:index");
		std::vector<u32> g_of(std::max<u32>(c.n_cells, 1), 0xFFFFFFFFu);
		for (u32 i = lo; i < hi; ++i) g_of[Gm[i].local_id] = i;
		c.remap.ensure(std::max<u32>(c.n_cells, 1));
		HIP_CHECK(hipMemcpyAsync(c.remap.p, g_of.data(), size_t(std::max<u32>(c.n_cells, 1)) * 4, hipMemcpyHostToDevice, c.stream));
		const u32 n_mol = c.n_mol;
		c.keys_a.ensure(std::max<u32>(n_mol, 1)); c.keys_b.ensure(std::max<u32>(n_mol, 1)); c.vals_a.ensure(std::max<u32>(n_mol, 1)); c.vals_b.ensure(std::max<u32>(n_mol, 1));
		u64 *k = c.keys_a.p, *k_alt = c.keys_b.p;
		u32 *v = c.vals_a.p, *v_alt = c.vals_b.p;
		std::vector<uint64_t> send(W_, 0), recv;
		if (n_mol) {
			c.scalars.ensure(16);
			HIP_CHECK(hipMemsetAsync(c.scalars.p, 0, 16, c.stream));
			FreeKeyArgs ka{c.mol_key.p, n_mol, c.layout.umi_bits, c.layout.gene_bits, gb, dest_shift, c.layout.gene_none, u32(world), c.remap.p, k, c.scalars.p};
			hipLaunchKernelGGL(free_index_keys_kernel, dim3(std::min<u32>(div_up(n_mol, 256), 4096u)), dim3(256), 0, c.stream, ka);
			HIP_CHECK(hipGetLastError());
			c.radix_sort(k, v, k_alt, v_alt, n_mol, ((1ull << (dest_shift + dest_bits)) - 1ull) | (1ull << 63), 0);   // sentinels (all ones) sort last
			std::vector<u64> bound(W_ + 1);
			for (int p = 0; p <= world; ++p) bound[size_t(p)] = u64(p) << dest_shift;
			const std::vector<u32> off = bounds_of(k, n_mol, bound);
			for (int p = 0; p < world; ++p) send[size_t(p)] = off[size_t(p) + 1] - off[size_t(p)];
		}
		counts_everywhere(send, recv);
		uint64_t total = 0;
		for (uint64_t x : recv) total += x;
		if (total > 0xFFFFFFF0ull) throw UnsupportedError("more than 2^32 UMI-gene records on one index shard");
		n_idx = u32(total);
		idx_a.alloc(std::max<u32>(n_idx, 1)); idx_b.alloc(std::max<u32>(n_idx, 1)); idv_a.alloc(1); idv_b.alloc(1);
		{ const void *snd[1] = {k}; void *rcv[1] = {idx_a.p}; const size_t elem[1] = {8};
		  tr->exchange(1, snd, rcv, elem, send.data(), recv.data(), c.stream); }
		uint64_t out = 0;
		for (int p = 0; p < world; ++p) if (p != rank) out += send[size_t(p)];
		phases["cbm:index"].bytes += double(out) * 8;
		ik = idx_a.p;
		if (n_idx) {
			hipLaunchKernelGGL(and_mask_kernel, dim3(div_up(n_idx, 256)), dim3(256), 0, c.stream, reinterpret_cast<unsigned long long *>(ik), n_idx, (1ull << dest_shift) - 1ull);
			HIP_CHECK(hipGetLastError());
			u64 *ik_alt = idx_b.p; u32 *iv = idv_a.p, *iv_alt = idv_b.p;
			c.radix_sort(ik, iv, ik_alt, iv_alt, n_idx, (1ull << dest_shift) - 1ull, 0);
		}
	}

	// B. partial counts of common UMI-genes per ordered pair, from my part of the index; C. routed to the owners of the bases
	SimplePairs P;
	{
		Phase ph(this, "cbm:pairs");
		DevBuf<u64> run_key; DevBuf<u32> run_cnt;
		u32 runs = 0;
		c.simple_pair_table(ik, n_idx, gb, d_gng.p, d_gcb.p, P, &run_key, &run_cnt, &runs);
		if (!run_key.p) { run_key.alloc(1); run_cnt.alloc(1); }
		std::vector<u64> bound(W_ + 1);
		for (int p = 0; p <= world; ++p) bound[size_t(p)] = u64(W.goff[size_t(p)]) << 32;
		const std::vector<u32> off = bounds_of(run_key.p, runs, bound);
		std::vector<uint64_t> send(W_, 0), recv;
		for (int p = 0; p < world && runs; ++p) send[size_t(p)] = off[size_t(p) + 1] - off[size_t(p)];
		counts_everywhere(send, recv);
		uint64_t total = 0;
		for (uint64_t x : recv) total += x;
		if (total > 0xFFFFFFF0ull) throw UnsupportedError("more than 2^32 partial pair counts on one shard");
		const u32 n_part = u32(total);
		DevBuf<u64> pk_a, pk_b; DevBuf<u32> pc_a, pc_b;
		pk_a.alloc(std::max<u32>(n_part, 1)); pk_b.alloc(std::max<u32>(n_part, 1)); pc_a.alloc(std::max<u32>(n_part, 1)); pc_b.alloc(std::max<u32>(n_part, 1));
		{ const void *snd[2] = {run_key.p, run_cnt.p}; void *rcv[2] = {pk_a.p, pc_a.p}; const size_t elem[2] = {8, 4};
		  tr->exchange(2, snd, rcv, elem, send.data(), recv.data(), c.stream); }
		uint64_t out = 0;
		for (int p = 0; p < world; ++p) if (p != rank) out += send[size_t(p)];
		phases["cbm:pairs"].bytes += double(out) * 12;
		if (n_part) {
			u64 *pk = pk_a.p, *pk_alt = pk_b.p; u32 *pc = pc_a.p, *pc_alt = pc_b.p;
			const u64 cm = (1ull << gb) - 1ull;
			c.radix_sort(pk, pc, pk_alt, pc_alt, n_part, (cm << 32) | cm);
			WeightedRuns wr{};
			wr.keys = reinterpret_cast<const unsigned long long *>(pk); wr.weight = pc;
			DevBuf<u64> fin_key; DevBuf<u32> fin_cnt;
			const u32 n_pairs = run_segmented_reduce(c, "simple:pair_sums", wr, n_part, 12, [&](u32 t) {
				fin_key.alloc(t + 1); fin_cnt.alloc(t + 1);
				zero_async(c, fin_cnt.p, size_t(t + 1) * 4);
				wr.run_key = reinterpret_cast<unsigned long long *>(fin_key.p); wr.out[0] = fin_cnt.p;
			});
			c.simple_pairs_to_host(fin_key.p, fin_cnt.p, n_pairs, d_gcb.p, P);
		}
	}

	// D. decisions of my bases
	std::vector<u64> gid;                                   // global first-seen rank of every cell of the list (replays only)
	std::unordered_map<size_t, u32> of_gid;
	SimpleCells C;
	C.umis = [&](u32 g) { return size_t(Gm[g].total_umis); };
	C.genes = [&](u32 g) { return size_t(Gm[g].n_genes); };
	C.barcode = [&](u32 g) { return decode_code(Gm[g].barcode, c.side); };
	C.slot = [&](u32 g) { return g - lo; };
	C.filtered_pos = [&](u32 g) { return pos_of[g]; };
	C.container_id = [&](u32 g) { return size_t(gid[g]); };
	C.from_container_id = [&](size_t id) { return of_gid.at(id); };
	std::vector<u32> tgt(hi - lo), nb_count(hi - lo, 0), replay;
	for (u32 i = lo; i < hi; ++i) tgt[i - lo] = i;
	{
		Phase ph(this, "cbm:decide");
		simple_decide(c.cfg, poisson, P, C, [&](const std::vector<u32> &pb, const std::vector<u32> &pc) { return free_expected(W, pb, pc); }, tgt, nb_count, replay);
	}

	// E. replays (every shard takes part in the collectives, also with nothing to replay)
	uint64_t n_replay = replay.size(), any_replay = 0;
	{
		std::vector<uint64_t> every(W_);
		tr->gather_host(&n_replay, 8, every.data());
		for (uint64_t x : every) any_replay += x;
	}
	if (any_replay) {
		Phase ph(this, "cbm:replay");
		// the cells' indices in the reference's container = ranks of their first reads in the WHOLE stream, over all cells of all shards
		{
			const u32 nc = c.n_cells;
			DevBuf<u64> d_fg, d_x; DevBuf<u32> d_cnt, d_ids;
			d_fg.alloc(std::max<u32>(nc, 1)); d_x.alloc(nG); d_cnt.alloc(nG); d_ids.alloc(std::max<u32>(hi - lo, 1));
			if (nc) hipLaunchKernelGGL(cell_first_global_kernel, dim3(div_up(nc, 256)), dim3(256), 0, c.stream, ordinal_map(), c.cell_first.p, nc, reinterpret_cast<unsigned long long *>(d_fg.p));
			HIP_CHECK(hipGetLastError());
			std::vector<u64> fg(std::max<u32>(nc, 1));
			if (nc) c.fetch(fg.data(), d_fg.p, size_t(nc) * 8);
			std::vector<u64> mine(hi - lo), X;
			for (u32 i = lo; i < hi; ++i) mine[i - lo] = fg[Gm[i].local_id];
			std::vector<size_t> cnt;
			tr->gather_vec(mine, X, cnt);
			if (X.size() != nG) throw InvalidError("internal: first reads of the merge cells do not cover the cell list");
			HIP_CHECK(hipMemcpyAsync(d_x.p, X.data(), size_t(nG) * 8, hipMemcpyHostToDevice, c.stream));
			hipLaunchKernelGGL(count_below_kernel, dim3(div_up(nG, 256)), dim3(256), 0, c.stream, reinterpret_cast<const unsigned long long *>(d_fg.p), nc,
			                   reinterpret_cast<const unsigned long long *>(d_x.p), nG, d_cnt.p);
			HIP_CHECK(hipGetLastError());
			std::vector<u32> below(nG), every(size_t(nG) * W_);
			c.fetch(below.data(), d_cnt.p, size_t(nG) * 4);
			tr->gather_host(below.data(), size_t(nG) * 4, every.data());
			gid.assign(nG, 0);
			for (int p = 0; p < world; ++p) for (u32 g = 0; g < nG; ++g) gid[g] += every[size_t(p) * nG + g];
			of_gid.reserve(size_t(nG) * 2);
			for (u32 g = 0; g < nG; ++g) if (!of_gid.emplace(size_t(gid[g]), g).second) throw InvalidError("internal: two cells with one global index");
		}
		// my bases' molecules in the reference's walk order (the UMI order is made global inside: a collective)
		dropest_ctx::SimpleReplayInput R;
		{
			std::vector<u32> local(replay.size());
			for (size_t r = 0; r < replay.size(); ++r) local[r] = Gm[replay[r]].local_id;
			c.simple_replay_local(local, R, true);
		}
		// the cells of every queried UMI-gene, from the shard that indexes it
		struct QRow { u64 low; u32 src, tag; };
		struct ARow { u32 src, tag, cnt, pad; };
		std::vector<QRow> myq(R.query.size()), allq;
		for (size_t i = 0; i < R.query.size(); ++i) myq[i] = QRow{R.query[i], u32(rank), u32(i)};
		std::vector<size_t> qc;
		tr->gather_vec(myq, allq, qc);
		std::vector<ARow> my_ans, all_ans; std::vector<u32> my_mem, all_mem;
		{
			std::vector<u64> qlow;
			for (const QRow &q : allq) if (int(mix64(q.low) % u64(world)) == rank) { qlow.push_back(q.low); my_ans.push_back(ARow{q.src, q.tag, 0, 0}); }
			const u32 nq = u32(qlow.size());
			if (nq) {
				DevBuf<u64> d_q; DevBuf<u32> d_cnt, d_off, d_cells;
				d_q.alloc(nq); d_cnt.alloc(nq); d_off.alloc(nq);
				HIP_CHECK(hipMemcpyAsync(d_q.p, qlow.data(), size_t(nq) * 8, hipMemcpyHostToDevice, c.stream));
				hipLaunchKernelGGL(umig_members_kernel, dim3(div_up(nq, 256)), dim3(256), 0, c.stream, reinterpret_cast<const unsigned long long *>(ik), n_idx, gb,
				                   reinterpret_cast<const unsigned long long *>(d_q.p), nq, static_cast<const u32 *>(nullptr), d_cnt.p, static_cast<u32 *>(nullptr));
				HIP_CHECK(hipGetLastError());
				std::vector<u32> q_cnt(nq), q_off(nq);
				c.fetch(q_cnt.data(), d_cnt.p, size_t(nq) * 4);
				uint64_t total = 0;
				for (u32 i = 0; i < nq; ++i) { q_off[i] = u32(total); total += q_cnt[i]; my_ans[i].cnt = q_cnt[i]; }
				if (total > 0xFFFFFFF0ull) throw UnsupportedError("too many UMI-gene members in the tie replay");
				d_cells.alloc(std::max<uint64_t>(total, 1));
				HIP_CHECK(hipMemcpyAsync(d_off.p, q_off.data(), size_t(nq) * 4, hipMemcpyHostToDevice, c.stream));
				hipLaunchKernelGGL(umig_members_kernel, dim3(div_up(nq, 256)), dim3(256), 0, c.stream, reinterpret_cast<const unsigned long long *>(ik), n_idx, gb,
				                   reinterpret_cast<const unsigned long long *>(d_q.p), nq, d_off.p, d_cnt.p, d_cells.p);
				HIP_CHECK(hipGetLastError());
				my_mem.resize(total);
				if (total) c.fetch(my_mem.data(), d_cells.p, size_t(total) * 4);
			}
		}
		std::vector<size_t> ac, mc;
		tr->gather_vec(my_ans, all_ans, ac);
		tr->gather_vec(my_mem, all_mem, mc);
		// (the answers of a shard stand in the order of its member list)
		const u32 nq_mine = u32(R.query.size());
		std::vector<u32> q_off(nq_mine, 0), q_cnt(nq_mine, 0);
		std::vector<uint8_t> answered(nq_mine, 0);
		size_t at = 0;
		for (const ARow &a : all_ans) {
			if (int(a.src) == rank) { if (a.tag >= nq_mine) throw InvalidError("internal: replay answer out of range"); q_off[a.tag] = u32(at); q_cnt[a.tag] = a.cnt; answered[a.tag] = 1; }
			at += a.cnt;
		}
		if (at != all_mem.size()) throw InvalidError("internal: replay answers and member lists disagree");
		for (uint8_t x : answered) if (!x) throw InvalidError("internal: a replay query was not answered");
		for (size_t r = 0; r < replay.size(); ++r)
			tgt[replay[r] - lo] = simple_replay_base(c.cfg, poisson, replay[r], R.in_order[r], all_mem, q_off, q_cnt, C, P, nb_count[replay[r] - lo]);
	}
	for (u32 i = lo; i < hi; ++i) my_tgt[i - lo] = int64_t(tgt[i - lo]);
}

// PoissonSimple across shards: expected intersection sizes (PoissonTargetEstimator::estimate_intersection_size through
// PoissonSimpleMergeStrategy.cpp:15-43) of the pairs (base of mine, neighbour anywhere).  The estimator's tables come from the UMI
// distribution of ALL shards; a pair is evaluated where the neighbour lives, from the base's molecule rows -- which travel anyway if the
// base merges -- exactly as the whitelist -M does (shard_run.h: cb_merge).  COLLECTIVE: every shard calls it once.
std::vector<double> dropest_shard::free_expected(const MergeWorld &W, const std::vector<dropest::u32> &pb, const std::vector<dropest::u32> &pc) {
	using namespace dropest;
	dropest_ctx &c = *ctx;
	const u32 lo = W.lo, hi = W.hi;
	{   // rows of my bases with a neighbour
		Phase ph(this, "cbm:export");
		dropest_ctx::ShardMerge &M = *c.shard;
		std::vector<u32> cells;
		M.listed_g.clear();
		for (size_t p = 0; p < pb.size(); ++p)
			if (p == 0 || pb[p] != pb[p - 1]) { cells.push_back(W.Gm[pb[p]].local_id); M.listed_g.push_back(pb[p]); }   // (pairs stand sorted by base)
		c.shard_export_rows(cells);
	}
	free_rows.reset(new TravelRows());
	TravelRows &T = *free_rows;
	merge_gather_rows(W, T);
	merge_umi_distribution();
	struct Pair { u32 base, other; };
	std::vector<Pair> mine(pb.size()), allp;
	for (size_t p = 0; p < pb.size(); ++p) mine[p] = Pair{pb[p], pc[p]};
	std::vector<size_t> pcnt;
	{ Phase ph(this, "cbm:gather_pairs"); tr->gather_vec(mine, allp, pcnt); }
	struct Ans { uint64_t pair; double expected; };
	std::vector<Ans> my_ans, all_ans;
	{
		Phase ph(this, "cbm:expected");
		std::vector<u32> cand_local; std::vector<uint64_t> bb, be, which;
		for (size_t i = 0; i < allp.size(); ++i)
			if (allp[i].other >= lo && allp[i].other < hi) {
				which.push_back(i); cand_local.push_back(W.Gm[allp[i].other].local_id);
				if (T.beg[allp[i].base] == ~0ull) throw InvalidError("internal: a base's molecule rows were not exported");
				bb.push_back(T.beg[allp[i].base]); be.push_back(T.end[allp[i].base]);
			}
		std::vector<double> expected(which.size(), 0.0);
		c.shard_merge_expected(which.size(), cand_local.data(), bb.data(), be.data(), reinterpret_cast<const uint64_t *>(T.low_all.p), expected.data());
		my_ans.resize(which.size());
		for (size_t i = 0; i < which.size(); ++i) my_ans[i] = Ans{which[i], expected[i]};
		std::vector<size_t> acnt;
		tr->gather_vec(my_ans, all_ans, acnt);
	}
	size_t first = 0;
	for (int p = 0; p < rank; ++p) first += pcnt[size_t(p)];
	std::vector<double> out(pb.size(), 0.0);
	for (const Ans &a : all_ans) if (a.pair >= first && a.pair < first + pb.size()) out[size_t(a.pair - first)] = a.expected;
	return out;
}
