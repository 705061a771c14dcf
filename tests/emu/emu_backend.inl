// tests/emu/emu_backend.inl — the CPU "backend" for vx_host.inl (tests only).
namespace {

struct ExecParamsView; // ExecParams is defined by vx_host.inl; stages are templates so they see it late

struct Backend {
	std::string lastError;
	bool largeClass = true; // the emulation has no capacity classes; the host logic sets this for the HIP backend
	u32 upperItemsHint = 0; // (HIP backend: sizes the launch of the levels >= 1)
	u32* tailDone = nullptr;
	bool tailCleaned = false;
	void set_next_reset(u32*, u32, u32*, u32, int* const*, const u32*) {}
	u32 slowHint[2] = { ~0u, ~0u }; // (HIP backend: sizes the launches of the general passes behind the table-driven ones)

	static int device_count() { return 1; }
	bool init(int, std::string&) { return true; }
	bool wants_pyramid() const { return false; }
	bool wants_bricks() const { return false; } // the emulation reads the dense fields
	void run_rebrick(const GridView&, const int*, const int*, const MirrorState&, const int*, const u32*, u32) {}
	void make_current() {} // the emulated phases sample the grid directly
	void shutdown() {}
	void set_stream(void*) {}
	std::string error() const { return lastError; }
	void* alloc(size_t bytes)
	{
		void* p = malloc(bytes ? bytes : 1);
		if (p) memset(p, 0xA5, bytes); // poison: device memory is not zeroed either
		return p;
	}
	void free(void* p) { ::free(p); }
	bool ipc_handle(void*, unsigned char*) { lastError = "the emulation has no device allocations to export"; return false; }
	bool fill(void* p, int v, size_t bytes) { memset(p, v, bytes); return true; }
	bool h2d(void* d, const void* s, size_t bytes) { memcpy(d, s, bytes); return true; }
	bool d2h(void* d, const void* s, size_t bytes) { memcpy(d, s, bytes); return true; }
	void sync() {}
	bool sync_ok() { return true; }
	float idle_before_ms() { return -1.f; }
	bool d2h_async(void* d, const void* s, size_t bytes) { memcpy(d, s, bytes); return true; }
	bool d2h_side(void* d, const void* s, size_t bytes) { memcpy(d, s, bytes); return true; }
	bool h2d_async(void* d, const void* s, size_t bytes) { memcpy(d, s, bytes); return true; }
	bool h2d_2d(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height)
	{
		for (size_t r = 0; r < height; ++r) memcpy((u8*)d + r * dpitch, (const u8*)s + r * spitch, width);
		return true;
	}
	void* alloc_pinned(size_t bytes) { return malloc(bytes ? bytes : 1); }
	void free_pinned(void* p) { ::free(p); }
	static void release_pinned(void* p) { ::free(p); }
	bool d2h_bulk(void* const* dst, const void* const* src, const size_t* bytes, int count)
	{
		for (int i = 0; i < count; ++i) if (bytes[i]) memcpy(dst[i], src[i], bytes[i]);
		return true;
	}
	void begin_timing() {}
	float end_timing_ms() { return 0.f; }
	// Grid file format v1 expanded block by block (the HIP backend does this in k_decode_grid)
	void run_decode_grid(const u8* blob, const uint64_t* where, u32 n, i8* dist, u8* mat, u8* blend, u8* flags)
	{
		const u32 nb = n / 16;
		for (u32 id = 0; id < nb * nb * nb; ++id) {
			const u8* rec = blob + where[id * 2];
			const uint64_t sizes = where[id * 2 + 1];
			u32 fl; memcpy(&fl, rec, 4);
			flags[id] = (u8)(fl & 1u);
			const u8* src = rec + 4;
			u8* outs[3] = { (u8*)dist, mat, blend };
			const u32 rawBit[3] = { 2u, 4u, 8u };
			const u32 bx = id % nb, by = (id / nb) % nb, bz = id / (nb * nb);
			for (int s = 0; s < 3; ++s) {
				const u32 sz = (u32)((sizes >> (16 * s)) & 0xFFFFu);
				u8 tmp[4096];
				memset(tmp, 0, sizeof(tmp));
				if (fl & rawBit[s]) memcpy(tmp, src, sz < 4096 ? sz : 4096);
				else {
					u32 pos = 0;
					for (u32 i = 0; i + 1 < sz; i += 2) for (u32 k = 0; k < src[i] && pos < 4096; ++k) tmp[pos++] = src[i + 1];
				}
				for (u32 z = 0; z < 16; ++z) for (u32 y = 0; y < 16; ++y)
					memcpy(outs[s] + ((size_t)(bz * 16 + z) * n + by * 16 + y) * n + bx * 16, tmp + z * 256 + y * 16, 16);
				src += sz;
			}
		}
	}
	void run_box_ids(u32* out, const u32 first[3], const u32 count[3], u32 nb)
	{
		u32 i = 0;
		for (u32 z = 0; z < count[2]; ++z) for (u32 y = 0; y < count[1]; ++y) for (u32 x = 0; x < count[0]; ++x)
			out[i++] = ((first[2] + z) * nb + first[1] + y) * nb + first[0] + x;
	}
	void run_edit(const GridView& g, u8* flags, const u32* ids, u32 count, const EditParams& e)
	{
		const u32 nb = (u32)g.n / 16;
		for (u32 t = 0; t < count; ++t) {
			const u32 id = ids[t], bx = id % nb, by = (id / nb) % nb, bz = id / (nb * nb);
			const EditSection s = edit_section(e, bx, by, bz);
			for (int iz = 0; iz < s.count[2]; ++iz) for (int iy = 0; iy < s.count[1]; ++iy) for (int ix = 0; ix < s.count[0]; ++ix) edit_voxel(g, e, s, ix, iy, iz);
			if (e.kind == EDIT_BALL) flags[id] = edit_block_empty(g, bx, by, bz);
		}
	}
	// Grid file format v1 written block by block (the HIP backend does this in k_encode_grid)
	void run_encode_grid(const GridView& g, u32* meta, const uint64_t* where, u8* blob)
	{
		const u32 n = (u32)g.n, nb = n / 16;
		const u8* arrays[3] = { (const u8*)g.dist, g.mat, g.blend };
		for (u32 id = 0; id < nb * nb * nb; ++id) {
			const u32 bx = id % nb, by = (id / nb) % nb, bz = id / (nb * nb);
			u32 flags = edit_block_empty(g, bx, by, bz) ? 1u : 0u;
			u8* dst = blob ? blob + where[id] + 4 : nullptr;
			for (int s = 0; s < 3; ++s) {
				const u8* a = arrays[s];
				bool raw;
				const u32 sz = encode_stream_serial([&](u32 r) { return a + ((size_t)(bz * 16 + (r >> 4)) * n + by * 16 + (r & 15)) * n + bx * 16; }, dst, raw);
				if (raw) flags |= 2u << s;
				if (!blob) meta[id * 4 + s] = sz;
				if (dst) dst += sz;
			}
			if (!blob) meta[id * 4 + 3] = flags; else memcpy(blob + where[id], &flags, 4);
		}
	}
	void run_heightmap(const GridView& g, const i8* map, u8* flags)
	{
		const u32 n = (u32)g.n, nb = n / 16;
		i8* dist = const_cast<i8*>(g.dist);
		for (u32 z = 0; z < n; ++z) for (u32 y = 0; y < n; ++y) for (u32 x = 0; x < n; ++x)
			dist[((size_t)z * n + y) * n + x] = heightmap_distance((int)z, (int)map[(size_t)y * n + x]);
		for (u32 id = 0; id < nb * nb * nb; ++id) flags[id] = edit_block_empty(g, id % nb, (id / nb) % nb, id / (nb * nb));
	}
	bool d2d(void* d, const void* s, size_t bytes) { memcpy(d, s, bytes); return true; }
	void run_terrain(const GridView& g, u32 seed, float* height, const int dr[4], const int mr[4], u8* flags, const u32* ids, u32 count, u32 style)
	{
		const u32 n = (u32)g.n, nb = n / 16;
		for (u32 y = 0; y < n; ++y) for (u32 x = 0; x < n; ++x) height[(size_t)y * n + x] = vxt::height(n, x, y, seed);
		for (int z = dr[0]; z < dr[1]; ++z) for (int y = dr[2]; y < dr[3]; ++y) for (u32 x = 0; x < n; ++x) {
			i8 d; u8 m, b;
			vxt::voxel(x, (u32)y, (u32)z, height[(size_t)y * n + x], seed, d, m, b, style);
			const_cast<i8*>(g.dist)[dist_offset(g, (int)x, y, z)] = d;
			if (z >= mr[0] && z < mr[1] && y >= mr[2] && y < mr[3]) { const size_t o = mat_offset(g, (int)x, y, z); const_cast<u8*>(g.mat)[o] = m; const_cast<u8*>(g.blend)[o] = b; }
		}
		for (u32 t = 0; t < count; ++t) { const u32 id = ids[t]; flags[id] = edit_block_empty(g, id % nb, (id / nb) % nb, id / (nb * nb)); }
	}
	void run_copy_segments(const u32* seg, u32 count, const void* src, void* dst, u32 elemBytes)
	{
		for (u32 i = 0; i < count; ++i)
			memcpy((u8*)dst + (size_t)seg[i * 3 + 1] * elemBytes, (const u8*)src + (size_t)seg[i * 3] * elemBytes, (size_t)seg[i * 3 + 2] * elemBytes);
	}
	void run_scatter_blocks(const u32* ids, u32 count, u32 n, const u8* sd, const u8* sm, const u8* sb, u8* dist, u8* mat, u8* blend)
	{
		const u32 nb = n / 16;
		const u8* src[3] = { sd, sm, sb };
		u8* dst[3] = { dist, mat, blend };
		for (u32 i = 0; i < count; ++i) {
			const u32 id = ids[i], bx = id % nb, by = (id / nb) % nb, bz = id / (nb * nb);
			for (int k = 0; k < 3; ++k) {
				if (!src[k]) continue;
				for (u32 z = 0; z < 16; ++z) for (u32 y = 0; y < 16; ++y)
					memcpy(dst[k] + ((size_t)(bz * 16 + z) * n + by * 16 + y) * n + bx * 16, src[k] + (size_t)i * 4096 + z * 256 + y * 16, 16);
			}
		}
	}
	void end_timing_record() {}
	void end_overlapped() {}
	u32 head_partials() const { return 0; } // the emulation counts the block classes into the header itself
	float elapsed_ms() { return 0.f; }
	template <typename P>
	void run_reset(const P& p, u32 levels, u32* header, u32 headerWords, u32*, u32, bool = false)
	{
		memset(header, 0, (size_t)headerWords * 4);
		for (u32 l = 0; l < levels; ++l) memset(p.levels[l].slotOf, 0xFF, (size_t)p.levels[l].cnt * p.levels[l].cnt * p.levels[l].cnt * 4);
	}
	bool stage_timing_on() const { return true; } // the emulation always runs the serial order
	template <typename P> void run_overlapped_tail(const P&, u32) {}
	template <typename P> bool single_stream(const P&, u32) const { return false; } // (HIP backend: one launch behind the classification)
	template <typename P> void run_main_staged(const P&, u32) {}
	u32 emitFrom = 0; // (HIP backend: partial runs; the emulation always meshes every level)
	template <typename P> bool partial_applies(const P&, u32) const { return false; }
	// (HIP backend: incremental runs as three launches; the emulation walks the chain with work lists)
	template <typename P> bool dirty_fused_applies(const P&, u32, bool) const { return false; }
	struct DirtyLaunch {
		u32 lo[MAX_LEVELS][3], hi[MAX_LEVELS][3], start[MAX_LEVELS + 1];
		u32* work; u32* info; u32* ticket; u32 ticketTarget;
		u32* header; u32 resetFrom, resetTo, poolVerts, poolIdx;
		u32* roleTicket; u32* slowDone;
		BlockRecord* hostRecs; u32* hostHeader; u32 headerWords, publishedWord;
	};
	template <typename P> void run_dirty_fused(const P&, u32, const DirtyLaunch&, bool, bool) {}
	void stage_enable(bool) {}
	void stage_mark(int) {}
	// halo messages: the same piece descriptors, moved with memcpy; no communicator (multi-process CPU runs exchange
	// through torch.distributed in voxels_amd/slab.py)
	void run_halo_moves(const HaloMove* lo, const HaloMove* hi, const GridView&, const MirrorState&, bool)
	{
		for (const HaloMove* m : { lo, hi }) {
			if (!m) continue;
			const HaloMove& mv = *m;
			for (u32 i = 0; i < mv.count; ++i) {
				const HaloPiece& p = mv.piece[i];
				for (int l = 0; l < p.layers; ++l) for (u32 a = 0; a < p.rows; ++a) {
					u8* f = p.field + halo_field_offset(p, p.firstLayer + l, a);
					u8* s = mv.staging + p.stagingOffset + ((size_t)l * p.rows + a) * p.rowBytes;
					if (mv.unpack) memcpy(f, s, p.rowBytes); else memcpy(s, f, p.rowBytes);
				}
			}
		}
	}
	static bool comm_unique_id(void*) { return false; }
	bool comm_init(int, int, const void*) { lastError = "the emulation has no communicator"; return false; }
	void comm_destroy() {}
	bool comm_exchange(int, const void*, size_t, void*, size_t, int, const void*, size_t, void*, size_t) { lastError = "the emulation has no communicator"; return false; }
	bool copy_from_peer(void* dst, Backend&, const void* src, size_t bytes) { memcpy(dst, src, bytes); return true; }
	bool lists_publish_header(u32*, const u32*, u32, u32*) { return false; }
	template <typename P>
	bool run_block_lists(const P& p, const ListPlan& plan, u32 levels)
	{
		for (u32 l = 0; l < levels; ++l) {
			const LevelDesc& L = p.levels[l];
			u32 n = 0;
			for (u32 id = 0; id < L.cnt * L.cnt * L.cnt; ++id) {
				const int slot = listed_block_slot(L, id);
				if (slot >= 0) listed_block_fill(L.listed[n++], L, id, (u32)slot, plan.idBase[l]);
			}
			plan.totals[l] = n;
		}
		return false;
	}
	bool stage_ms(float*) { return false; }
	bool run_selftest(u32* out) { memset(out, 0, 16 * 4); return true; } // the device forms do not exist here

	// classification of one level-0 block (portable form of k_classify)
	template <typename P>
	static void classify_block(const P& p, u32 bx, u32 by, u32 bz, bool accumulate, std::vector<i8>& samp)
	{
		stage_samples(p.G.grid, bx, by, bz, 1, samp.data(), 0, 1);
		u32 bits[128];
		memset(bits, 0, sizeof(bits));
		u32 cnt = 0;
		for (int c = 0; c < BLOCK_CELLS; ++c) {
			i8 V[8];
			cell_values(samp.data(), c & 15, (c >> 4) & 15, c >> 8, V);
			const u32 code = reg_case_code(V);
			if (code != 0 && code != 255) { bits[c >> 5] |= 1u << (c & 31); ++cnt; }
		}
		publish_level0_block(p.G, p.levels[0], bx, by, bz, bits, cnt, accumulate);
	}

	template <typename P> bool classify_activates_ancestors(const P&) const { return false; }
	template <typename P> bool ancestors_with_classification(const P&, u32) const { return false; }
	template <typename P>
	void run_classify(const P& p, bool)
	{
		const LevelDesc& L = p.levels[0];
		std::vector<i8> samp(SAMPLES + 7);
		for (u32 bz = L.zb0; bz < L.zb1; ++bz)
		for (u32 by = L.yb0; by < L.yb1; ++by)
		for (u32 bx = 0; bx < L.cnt; ++bx) classify_block(p, bx, by, bz, false, samp);
	}

	template <typename P>
	void run_classify_blocks(const P& p, const u32* coords, u32 count)
	{
		const LevelDesc& L = p.levels[0];
		std::vector<i8> samp(SAMPLES + 7);
		for (u32 k = 0; k < count; ++k) {
			u32 bx, by, bz;
			block_coords(coords[k], L.cnt, bx, by, bz);
			classify_block(p, bx, by, bz, true, samp);
		}
	}

	template <typename P>
	void run_build_worklist(const P& p, const u32* coords, const u32* start, const u32* cnt, u32 levels, u32* work)
	{
		for (u32 l = 0; l < levels; ++l) {
			const LevelDesc& L = p.levels[l];
			for (u32 k = 0; k < cnt[l]; ++k) {
				const int slot = L.slotOf[coords[start[l] + k]];
				if (slot >= 0) work[start[l] + p.G.workCount[l]++] = (u32)slot;
			}
		}
	}

	template <typename P>
	void run_gather_records(const P& p, u32 levels, const u32* start, BlockRecord* out)
	{
		for (u32 l = 0; l < levels; ++l)
			for (u32 i = 0; i < p.G.workCount[l]; ++i) out[start[l] + i] = p.levels[l].records[p.G.workItems[l][i]];
	}

	// slots to process on a level: all of them (full run) or the work list (incremental run)
	template <typename P>
	static u32 item_count(const P& p, u32 level) { return p.G.dirty ? p.G.workCount[level] : *p.levels[level].nActive; }
	template <typename P>
	static u32 item_slot(const P& p, u32 level, u32 i) { return p.G.dirty ? p.G.workItems[level][i] : i; }

	template <typename P>
	void run_hierarchy(const P& p, u32 levels, bool = false)
	{
		const LevelDesc& L0 = p.levels[0];
		for (u32 s = 0; s < *L0.nActive; ++s) {
			u32 bx, by, bz;
			block_coords(L0.slotCoord[s], L0.cnt, bx, by, bz);
			for (u32 l = 1; l < levels; ++l) {
				const LevelDesc& L = p.levels[l];
				const u32 px = bx >> l, py = by >> l, pz = bz >> l;
				if (px >= L.cnt || py >= L.cnt || pz >= L.cnt) break;
				const u32 id = block_coord_id(px, py, pz, L.cnt);
				if (L.slotOf[id] >= 0) break;
				const u32 slot = (*L.nActive)++;
				L.slotOf[id] = (int)slot;
				L.slotCoord[slot] = id;
			}
		}
	}

	template <typename P>
	void run_material(const P& p, u32 level)
	{
		const LevelDesc& L = p.levels[level];
		MatState* st = new MatState;
		for (u32 it = 0; it < item_count(p, level); ++it) {
			const u32 slot = item_slot(p, level, it);
			u32 bx, by, bz;
			block_coords(L.slotCoord[slot], L.cnt, bx, by, bz);
			stage_samples(p.G.grid, bx, by, bz, L.mult, st->samp, 0, 1);
			memset(st->ntBits, 0, sizeof(st->ntBits));
			mat_phase_classify(*st, 0, 1);
			mat_phase_children(*st, p.levels, level, bx, by, bz, 0, 1);
			st->voteCount = 0;
			mat_phase_select(*st, p.G, p.levels, level, slot, bx, by, bz, 0, 1);
			mat_phase_vote(*st, p.G, p.levels, level, slot, bx, by, bz, 0, 1);
		}
		delete st;
	}

	// one capacity class of the regular pass: blocks whose non-trivial cell count is in (lo, CAP]
	template <int CAP, typename P>
	void regular_pass(const P& p, u32 levels, u32 lo)
	{
		const Tables T = tables_from_image(p.tables);
		typedef RegStateT<CAP> ST;
		ST* st = new ST;
		// level 0 of a full run: blocks without a zero sample take the table-driven pass (tv_fast0.h), like k_regular0_fast
		std::vector<unsigned long long> vrow(256, 0);
		for (u32 code = 0; code < 256; ++code) memcpy(&vrow[code], p.tables + TAB_REG_VERT + code * 6, 6);
		const F0Tables FT = f0_tables_from_image(p.tables, vrow.data());
		Fast0State<640>* fst = new Fast0State<640>;
		Fast1State<640>* fst1 = new Fast1State<640>;
		u32 fastBlocks = 0, generalBlocks = 0, fastCoarse = 0, generalCoarse = 0;
		for (u32 level = 0; level < levels; ++level) {
			const LevelDesc& L = p.levels[level];
			for (u32 it = 0; it < item_count(p, level); ++it) {
				const u32 slot = item_slot(p, level, it);
				const u32 ntc = L.ntCount[slot];
				if ((lo && ntc <= lo) || ntc > (u32)CAP) continue; // the first class (lo == 0) also owns empty blocks
				RegBlockCtx b;
				b.level = level; b.slot = slot; b.mult = L.mult;
				block_coords(L.slotCoord[slot], L.cnt, b.bx, b.by, b.bz);
				if (ntc == 0 || (level == 0 && L.skip[slot])) { reg_write_empty_record(L, slot); continue; }
				if (level == 0 && !p.G.dirty && CAP == 640 && !getenv("VX_EMU_NO_FAST0") && f0_block_serial(*fst, FT, p.G, L, p.P, slot, b.bx, b.by, b.bz, p.G.stats)) { ++fastBlocks; continue; }
				if (level >= 1 && level < PYRAMID_LEVELS && !p.G.dirty && CAP == 640 && !getenv("VX_EMU_NO_FAST1") && f1_block_serial(*fst1, FT, p.G, L, p.P, level, slot, b.bx, b.by, b.bz, p.G.stats)) { ++fastCoarse; continue; }
				if (level == 0) ++generalBlocks; else ++generalCoarse;
				reg_phase_begin(*st, L, slot, 0, 1);
				reg_phase_stage(*st, p.G, L, b, 0, 1);
				for (int w = 0; w < 128; ++w) st->wordPrefix[w] = (u16)TV_POPC(st->ntBits[w]);
				st->wordPrefix[128] = (u16)exclusive_scan(st->wordPrefix, 128);
				reg_phase_list(*st, L, b, 0, 1);
				reg_phase_cells(*st, T, p.G, L, b, 0, 1);
				reg_phase_count(*st, T, b, 0, 1);
				st->vTotal = exclusive_scan(st->vbase, st->wordPrefix[128]);
				st->vOff = TV_ATOMIC_ADD(&p.P.cursors[CUR_V], st->vTotal);
				for (u32 chunk = 0; chunk == 0 || chunk < st->vTotal; chunk += VDESC_CAP) {
					reg_phase_describe(*st, chunk, 0, 1);
					reg_phase_emit_vertices(*st, T, p.G, p.P, b, chunk, 0, 1);
				}
				reg_phase_keep(*st, T, p.G, b, 0, 1);
				st->iTotal = exclusive_scan(st->ibase, st->wordPrefix[128]);
				st->iOff = TV_ATOMIC_ADD(&p.P.cursors[CUR_I], st->iTotal);
				for (u32 chunk = 0; chunk < st->iTotal; chunk += VDESC_CAP) {
					reg_phase_stage_indices(*st, T, chunk, 0, 1);
					reg_phase_flush_indices(*st, T, p.P, chunk, 0, 1);
				}
				reg_phase_record(*st, p.G.stats, L, b, p.P, 0);
			}
		}
		delete st;
		delete fst;
		delete fst1;
		if (getenv("VX_EMU_TRACE") && (fastBlocks || generalBlocks)) fprintf(stderr, "[emu] level-0 blocks: %u through the fast pass, %u through the general pass (class <= %d cells)\n", fastBlocks, generalBlocks, CAP);
		if (getenv("VX_EMU_TRACE") && (fastCoarse || generalCoarse)) fprintf(stderr, "[emu] level >= 1 blocks: %u through the fast pass, %u through the general pass (class <= %d cells)\n", fastCoarse, generalCoarse, CAP);
	}

	template <typename P>
	void run_regular(const P& p, u32 levels)
	{
		regular_pass<640>(p, levels, 0);      // same two capacity classes as the HIP kernels
		regular_pass<4096>(p, levels, 640);
	}

	template <typename P>
	void run_transition(const P& p, u32 levels)
	{
		const Tables T = tables_from_image(p.tables);
		TrState* st = new TrState;
		for (u32 level = 1; level < levels; ++level) {
			const LevelDesc& L = p.levels[level];
			if (!L.hasTransitions) continue;
			for (u32 it = 0; it < item_count(p, level); ++it) {
				const u32 slot = item_slot(p, level, it);
				RegBlockCtx b;
				b.level = level; b.slot = slot; b.mult = L.mult;
				block_coords(L.slotCoord[slot], L.cnt, b.bx, b.by, b.bz);
				tr_phase_load(*st, p.G, L, b, 0, 1);
				tr_phase_classify(*st, 0, 1);
				for (int f0 = 0; f0 < 6;) {
					const int f1 = tr_batch_end(*st, f0);
					tr_phase_batch_bits(*st, f0, f1, 0, 1);
					st->wordPrefix[48] = (u16)exclusive_scan(st->wordPrefix, 48);
					st->vTotal = st->iTotal = st->vOff = st->iOff = 0;
					if (st->wordPrefix[48]) {
						tr_phase_cells_of(*st, 0, 1);
						tr_phase_list(*st, T, L, b, 0, 1);
						tr_phase_count(*st, T, 0, 1);
						st->vTotal = exclusive_scan(st->vbase, st->wordPrefix[48]);
						st->iTotal = exclusive_scan(st->ibase, st->wordPrefix[48]);
						st->vOff = TV_ATOMIC_ADD(&p.P.cursors[CUR_V], st->vTotal);
						st->iOff = TV_ATOMIC_ADD(&p.P.cursors[CUR_I], st->iTotal);
						for (u32 chunk = 0; chunk == 0 || chunk < st->vTotal; chunk += VDESC_CAP) {
							tr_phase_describe(*st, chunk, 0, 1);
							tr_phase_emit_vertices(*st, T, p.G, F1HostSampler{ &p.G.grid }, p.P, b, chunk, 0, 1);
						}
						for (u32 chunk = 0; chunk < st->iTotal; chunk += TR_INDEX_CHUNK) {
							tr_phase_stage_indices(*st, T, chunk, 0, 1);
							tr_phase_flush_indices(*st, T, p.P, chunk, 0, 1);
						}
					}
					tr_phase_record(*st, L, b, p.P, f0, f1, 0);
					f0 = f1;
				}
			}
		}
		delete st;
	}
};

} // namespace
