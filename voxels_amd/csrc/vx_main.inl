// vx_main.inl — k_main: everything a full run does between k_run_head (slots) and k_tail (lists) as ONE launch (gfx950).
// There is no classification pass in front of it: level-0 blocks form their own bitmaps (f0_walk<.., SELF>), level-1
// material blocks their children's (mat_block, selfChild).  Included by vx_hip.hip
// behind the passes whose per-block bodies it calls: f0_walk (table-driven regular cells of level-0 blocks, vx_fast0.inl),
// mat_block (the material vote of one block of a level >= 1, vx_hip.hip), f1_block (the table-driven regular cells of one
// block of a level 1..3, vx_fast1.inl), tr_block (the transition cells of one block, vx_hip.hip).
//
// The reference walks the levels one after the other (TransVoxelRun::Execute, src/TransVoxelImpl.cpp:492-531): the material
// cache of a level-L cell is a vote over its eight children on level L-1 (:753-838), a block's regular and transition
// cells read its own cache, and level 0 needs none of that.  As launches that is a chain - material L1, L2, L3, then the
// passes over the levels >= 1, the level-0 pass beside them on a second stream, stream events at the fork and at the join -
// whose fixed cost is the whole of a small run (a 128^3 grid, a rank's slab of an 8-GPU job).  Here the same work is two
// queues handed out by atomic counters to persistent workgroups:
//   * the upper queue, in dependency order:
//         [ material blocks of level 1 | of level 2 | ... | regular blocks of levels 1..3 | transition blocks ]
//     An item waits for exactly what it reads from other workgroups, as late as it can: a material block for its children's
//     caches right in front of its vote (sampling, classification and candidate selection need none of them), a regular block
//     for its own material behind the request of its lattice samples, a transition block for it behind planes,
//     classification and scans.  The flags are LevelDesc::matDone (publish_done_through / wait_done: write-through stores and
//     loads, placement-independent, every wait bounded).
//   * the level-0 queue: the active level-0 slots, taken in batches of consecutive slots (x-neighbour blocks: a batch is
//     walked with the register prefetch of f0_walk, and its blocks share halo lines in one XCD's L2).
// A workgroup prefers one of the queues (a share of the workgroups the upper one, so that the latency-bound dependency chain
// of the levels >= 1 starts at once and runs beside the bandwidth-hungry level-0 blocks on every CU) and moves on to the
// other when its own is empty: nobody idles while work is left, and no stream event or second stream is involved.
//
// Progress: an item only waits for items BEFORE it in the upper queue; those were dequeued earlier, by workgroups that are
// running (a workgroup dequeues while it runs, never before), and the first unfinished item of the queue waits for
// nothing.  So no co-residency of the whole grid is needed, and no order of dispatch is assumed.
namespace {

constexpr u32 UP_TAB_LDS = TR_TAB_LDS > F0_TAB_LDS ? TR_TAB_LDS : F0_TAB_LDS;      // one table image at a time: regular (F0) or transition
constexpr u32 UP_STATE_LDS = sizeof(TrState) > sizeof(Fast1State<REG_CAP_SMALL>)
	? (sizeof(TrState) > sizeof(MatLds) ? sizeof(TrState) : sizeof(MatLds))
	: (sizeof(Fast1State<REG_CAP_SMALL>) > sizeof(MatLds) ? sizeof(Fast1State<REG_CAP_SMALL>) : sizeof(MatLds));
constexpr u32 MAIN_STATE_LDS = sizeof(Fast0State<REG_CAP_SMALL>) > UP_STATE_LDS ? sizeof(Fast0State<REG_CAP_SMALL>) : UP_STATE_LDS;
static_assert((UP_TAB_LDS & 15u) == 0, "the state behind the tables stays 16-byte aligned");

struct MainPlan {
	u32 levels;     // levels of the run: material items for 1 .. levels - 1
	u32 fastEnd;    // regular items for the levels 1 .. fastEnd - 1 (the levels with a lattice copy)
	u32 level0;     // 1: the level-0 queue is part of the launch (its LDS then holds a Fast0State), and no classification pass ran
	                // (k_run_head<allocate>): level-0 blocks and level-1 material blocks form the bitmaps they need
	u32 batch;      // level-0 slots per dequeue (one head for the chip: a head per XCD over spatial granules - round 6,
	                // profiles/r06_xcd_heads.txt - fetched 9 % fewer lines and was no faster)
	u32 upperNum, upperDen; // workgroups with blockIdx % upperDen < upperNum prefer the upper queue
	// incremental runs (k_main<true>, vx_polygonize_dirty): the queues hand out the entries of the levels' work lists
	// (Globals::workItems / workCount, written by k_dirty_head); a material block only waits for the children that are part of
	// this run - those inside the dirty box of their level ([boxLo, boxHi) in block coordinates x, y, z)
	u32 boxLo[MAX_LEVELS][3], boxHi[MAX_LEVELS][3];
	// partial runs (k_main<false, true>, vx_polygonize_from): the levels below emitFrom keep their caches and bitmaps up to date
	// but produce no meshes (another device produces those, libVoxels.so with VOXELS_DEVICES): no level-0 queue, regular and
	// transition items only for the levels >= emitFrom
	u32 emitFrom;
};

// DIRTY: the incremental run's form (TransVoxelRun::Execute with a Modification, src/TransVoxelImpl.cpp:429-465): the same two
// queues over the work lists k_dirty_head wrote - level-0 blocks with the bitmaps the head formed (no SELF), material blocks
// that keep the old cache contents where the reference does (mat_block: defineAll), and the levels beyond the lattice copies
// handed to the general pass of the run's last kernel through Globals::slowItems[1].
// PARTIAL: the levels below plan.emitFrom are somebody else's to mesh (the helper devices of a multi-device Execute): their
// material blocks run as always - the caches a later Modification continues from - and the level-1 material blocks also
// write the bitmaps of their level-0 children, which no level-0 walk forms in such a run.
template <bool DIRTY, bool PARTIAL = false>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(4))) void k_main(ExecParamsDev pArg, MainPlan plan)
{
#define MAIN_PARAMS() (void)pArg; const ExecParamsDev& p = kernarg_params() // (vx_hip.hip: read afresh, per item)
	MAIN_PARAMS();
	u8* tab = smem;
	u8* state = smem + UP_TAB_LDS;
	// (one 16-byte aligned block of statics in front of the dynamic region: its base stays aligned for the 16-byte LDS accesses)
	__shared__ __attribute__((aligned(16))) struct { u32 wgStats[20]; u32 scanScratch[8]; u32 zeroFlag[2]; u32 quietFaces[2]; u32 zeroFlag0[2]; u32 nextItem[2]; } sh;
	static_assert(sizeof(sh) % 16 == 0, "static LDS in front of the dynamic region");
	u32* const wgStats = sh.wgStats; u32* const scanScratch = sh.scanScratch; u32* const zeroFlag = sh.zeroFlag; u32* const quietFaces = sh.quietFaces;

	const int tid0 = (int)threadIdx.x;
	if (tid0 < 20) wgStats[tid0] = 0;
	if (tid0 < 2) { zeroFlag[tid0] = 0; quietFaces[tid0] = 0; sh.zeroFlag0[tid0] = 0; }
	// the upper queue's segments (the slot counts of all levels are final since the classification): matEnd[l] = end of level
	// l's material items; the regular and the transition items are prefixes of the same order
	u32 matEnd[MAX_LEVELS];
	u32 run = 0;
#pragma unroll
	for (u32 l = 0; l < MAX_LEVELS; ++l) {
		if (l >= 1 && l < plan.levels) run += DIRTY ? p.G.workCount[l] : p.G.slotCounts[l];
		matEnd[l] = r0_uniform(run);
	}
	const u32 matTotal = matEnd[MAX_LEVELS - 1];
	u32 regTotal = 0, trTotal = 0, skipItems = 0; // (PARTIAL: the blocks of the levels 1 .. emitFrom - 1 lead both segments and are left out)
#pragma unroll
	for (u32 l = 1; l < MAX_LEVELS; ++l) {
		if (l < plan.fastEnd) regTotal = matEnd[l];
		if (l < plan.levels && p.levels[l].hasTransitions) trTotal = matEnd[l];
		if (PARTIAL && l + 1u == plan.emitFrom) skipItems = matEnd[l];
	}
	if (PARTIAL) { regTotal = max(regTotal, skipItems) - skipItems; trTotal = max(trTotal, skipItems) - skipItems; }
	if (VX_ABL & 4) trTotal = 0;
	if (VX_ABL & 8) regTotal = 0;
	const u32 upperTotal = matTotal + regTotal + trTotal;
	const u32 total0 = (plan.level0 && !(VX_ABL & 128)) ? r0_uniform(DIRTY ? p.G.workCount[0] : p.G.slotCounts[0]) : 0u;

	u32 tabKind = 0;          // which table image the LDS holds: 0 none, 1 regular (F0), 2 transition
	F0Tables FT = {};
	Tables TT = {};
	u32 parity = 0, quietParity = 0, parity0 = 0;
	bool upperLeft = upperTotal != 0, level0Left = total0 != 0; // (this workgroup's knowledge: a queue is empty once a dequeue came back beyond its end)
	const bool preferUpper = (blockIdx.x % plan.upperDen) < plan.upperNum;
	// One barrier per item says both "the previous item is done with the LDS state" and "the ticket is there": the ticket travels
	// through sh.nextItem[item parity].  (Round 6: drawing the next ticket one item ahead - the returning atomic's 2 us round trip
	// behind the item instead of in front of it - was 2.5 % SLOWER for either queue: the answer comes back in order with the
	// item's first loads and is waited for with them, and a held upper item delays whoever depends on it.)
	u32 turn = 0;

#if defined(VX_MAIN_PROFILE)
	// tools builds: where the workgroups' time goes, by role (cycles as thread 0 sees them; header words behind the large-block counter)
	unsigned long long profTick = __builtin_readcyclecounter();
	u32 prof[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#define MAIN_TICK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); prof[i] += (u32)((now_ - profTick) >> 6); profTick = now_; } while (0)
#else
#define MAIN_TICK(i) do { } while (0)
#endif
#if defined(VX_MAIN_TRACE)
	unsigned long long trT0 = 0, trT1 = 0; u32 trWhat = 0;
#define MAIN_TRACE_END() do { if (tid0 == 0) { const u32 at_ = atomicAdd(&g_mainTraceN, 1u); if (at_ < 8192u) { unsigned long long* e_ = g_mainTrace + 18u * at_; for (u32 m_ = 0; m_ < 12u; ++m_) { e_[6u + m_] = g_marks[(blockIdx.x & 4095u) * 12u + m_]; g_marks[(blockIdx.x & 4095u) * 12u + m_] = 0; } e_[0] = trWhat; e_[1] = blockIdx.x; e_[2] = trT0; e_[3] = trT1; e_[4] = g_waitEnd[blockIdx.x & 4095u]; e_[5] = __builtin_amdgcn_s_memrealtime(); } } } while (0)
#else
#define MAIN_TRACE_END() do { } while (0)
#endif
	while (upperLeft || level0Left) {
		MAIN_PARAMS();
#if defined(VX_MAIN_TRACE)
		trT0 = __builtin_amdgcn_s_memrealtime();
#endif
		int tid = tid0;
		asm volatile("" : "+v"(tid)); // (per item: what a lane derives from its index alone is not hoisted out of the loop and kept in registers)
		const bool takeUpper = upperLeft && (preferUpper || !level0Left);
		turn ^= 1u;
		if (tid0 == 0) sh.nextItem[turn] = takeUpper ? atomicAdd(p.G.upperHead, 1u) : atomicAdd(p.G.level0Head, plan.batch);
		__syncthreads(); // the previous item is done (with the LDS state), and the ticket is there
		MAIN_TICK(0);
		if (!takeUpper) {
			// ---- a batch of consecutive level-0 slots -----------------------------------------------------------------
			const u32 first = r0_uniform(sh.nextItem[turn]);
			if (first >= total0) { level0Left = false; continue; }
			if (tabKind != 1u) { FT = f0_stage_tables(tab, p.tables, (u32)tid); tabKind = 1u; } // (visible after the walk's first barrier)
			MAIN_TICK(1);
#if defined(VX_MAIN_TRACE)
			trT1 = __builtin_amdgcn_s_memrealtime(); trWhat = first; if (tid0 == 0) g_waitEnd[blockIdx.x & 4095u] = 0;
#endif
			f0_walk<REG_CAP_SMALL, false, !DIRTY>(p, FT, *(Fast0State<REG_CAP_SMALL>*)state, wgStats, sh.zeroFlag0, parity0, total0, 0u, first, 1u, min(first + plan.batch, total0), tid);
			MAIN_TICK(2);
			MAIN_TRACE_END();
			continue;
		}
		// ---- one item of the upper queue (a few thousand items per run) ---------------------------------------------------
		const u32 item = r0_uniform(sh.nextItem[turn]);
		MAIN_TICK(1);
		if (item >= upperTotal) { upperLeft = false; continue; }
		// item -> (kind, level, slot): the position inside its segment, looked up in the level boundaries
		const bool isMat = item < matTotal, isReg = !isMat && item < matTotal + regTotal;
		const u32 f = isMat ? item : ((isReg ? item - matTotal : item - matTotal - regTotal) + (PARTIAL ? skipItems : 0u));
		u32 level = 1, base = 0;
#pragma unroll
		for (u32 l = 1; l + 1 < MAX_LEVELS; ++l) if (f >= matEnd[l]) { level = l + 1; base = matEnd[l]; }
		u32 slot = f - base;
		if (DIRTY) slot = r0_uniform(p.G.workItems[level][slot]);
#if defined(VX_MAIN_TRACE)
		trT1 = __builtin_amdgcn_s_memrealtime(); trWhat = ((isMat ? 1u : (isReg ? 2u : 3u)) << 28) | (level << 24) | slot; if (tid0 == 0) g_waitEnd[blockIdx.x & 4095u] = 0;
#endif
		if (isMat) {
			if (DIRTY) {
				mat_block<true>(p, level, slot, *(MatLds*)state, tid, false, plan.boxLo[level - 1u], plan.boxHi[level - 1u]);
				// a level without a lattice copy has no table-driven regular pass: the general pass of the run's last kernel takes the block
				if (level >= plan.fastEnd && tid0 == 0) p.G.slowItems[1][atomicAdd(&p.G.slowCount[1], 1u)] = (level << 24) | slot;
			} else if (PARTIAL)
				mat_block<true, true>(p, level, slot, *(MatLds*)state, tid, true, nullptr, nullptr, plan.emitFrom);
			else
				mat_block<true>(p, level, slot, *(MatLds*)state, tid, (VX_ABL & 256) ? false : plan.level0 != 0u);
			MAIN_TICK(3);
			MAIN_TRACE_END();
			continue;
		}
		const LevelDesc& L = p.levels[level];
		const u32 coord = r0_uniform(L.slotCoord[slot]);
		const GridView& g = p.G.grid;
		const F1BrickSampler smp = { g.bDist, g.bMat, g.bBlend, g.n - 1, (u32)g.n >> 4, (u32)g.bRowsY, g.bYb0, g.bZb0 };
		if (isReg) {
			if (tabKind != 1u) { FT = f0_stage_tables(tab, p.tables, (u32)tid); tabKind = 1u; } // (behind the barriers of the dequeue; visible after the block's first barrier)
			f1_block<REG_CAP_SMALL, true>(p, FT, smp, *(Fast1State<REG_CAP_SMALL>*)state, wgStats, zeroFlag, parity, level, slot, coord, 0u, 0u, tid);
			MAIN_TICK(4);
		} else {
			if (tabKind != 2u) { TT = stage_transition_tables(tab, p.tables, (u32)tid); tabKind = 2u; }
			RegBlockCtx b;
			b.level = level; b.slot = slot;
			tr_block<false, true>(p, b, coord, *(TrState*)state, TT, scanScratch, quietFaces, quietParity, smp, tid, false);
			MAIN_TICK(5);
		}
		MAIN_TRACE_END();
	}
	__syncthreads();
	{
		int tid = tid0;
		asm volatile("" : "+v"(tid)); // (the address is formed here, not carried through the kernel)
		if (tid < 20 && wgStats[tid]) atomicAdd(&p.G.stats[tid], wgStats[tid]);
	}
#if defined(VX_MAIN_PROFILE)
	MAIN_TICK(6);
	if (tid0 == 0) for (int i = 0; i < 8; ++i) if (prof[i]) atomicAdd(&p.G.largeBlocks[4 + i], prof[i]);
#endif
}

} // namespace
