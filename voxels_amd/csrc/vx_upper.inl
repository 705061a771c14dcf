// vx_upper.inl — k_upper: everything a full run does on the LOD levels >= 1 as ONE launch (gfx950).  Included by vx_hip.hip
// behind the passes whose per-block bodies it calls: mat_block (the material vote of one block, vx_hip.hip), f1_block (the
// table-driven regular cells of one block, vx_fast1.inl), tr_block (the transition cells of one block, vx_hip.hip).
//
// The reference walks the levels one after the other (TransVoxelRun::Execute, src/TransVoxelImpl.cpp:492-531): the material
// cache of a level-L cell is a vote over its eight children on level L-1 (:753-838), and a block's regular and transition
// cells read its own cache.  As launches that is a chain - material L1, L2, L3, then the two passes over all levels, each
// behind a stream event - whose fixed cost is the whole of a small run (a 128^3 grid, a rank's slab of an 8-GPU job).
// Here the same work is a queue of items in dependency order,
//     [ material blocks of level 1 | of level 2 | ... | regular blocks of levels 1..3 | transition blocks ]
// handed out by one atomic counter to persistent workgroups.  An item waits for exactly what it reads from other
// workgroups, as late as it can: a material block for its children's caches right in front of its vote (sampling,
// classification and candidate selection need none of them), a regular block for its own material behind the request of
// its lattice samples, a transition block for it behind planes, classification and scans.  The flags are
// LevelDesc::matDone (publish_done / wait_done above: agent-scope release / acquire, placement-independent).
//
// Progress: an item only waits for items BEFORE it in the queue; those were dequeued earlier, by workgroups that are
// running (a workgroup dequeues while it runs, never before), and the first unfinished item of the queue waits for
// nothing.  So no co-residency of the whole grid is needed, and no order of dispatch is assumed.
namespace {

constexpr u32 UP_TAB_LDS = TR_TAB_LDS > F0_TAB_LDS ? TR_TAB_LDS : F0_TAB_LDS;      // one table image at a time: regular (F0) or transition
constexpr u32 UP_STATE_LDS = sizeof(TrState) > sizeof(Fast1State<REG_CAP_SMALL>)
	? (sizeof(TrState) > sizeof(MatLds) ? sizeof(TrState) : sizeof(MatLds))
	: (sizeof(Fast1State<REG_CAP_SMALL>) > sizeof(MatLds) ? sizeof(Fast1State<REG_CAP_SMALL>) : sizeof(MatLds));
static_assert((UP_TAB_LDS & 15u) == 0, "the state behind the tables stays 16-byte aligned");

struct UpperPlan {
	u32 levels;    // levels of the run: material items for 1 .. levels - 1
	u32 fastEnd;   // regular items for the levels 1 .. fastEnd - 1 (the levels with a lattice copy)
	u32 persistent; // 1: workgroups stay until the queue is empty; 0: a workgroup takes its share (items / workgroups) and leaves
};

#if !defined(VX_UP_WAVES)
#define VX_UP_WAVES 5
#endif

__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(VX_UP_WAVES))) void k_upper(ExecParamsDev p, UpperPlan plan)
{
	u8* tab = smem;
	u8* state = smem + UP_TAB_LDS;
	// (one 16-byte aligned block of statics in front of the dynamic region: its base stays aligned for the 16-byte LDS accesses)
	__shared__ __attribute__((aligned(16))) struct { u32 wgStats[20]; u32 scanScratch[8]; u32 zeroFlag[2]; u32 quietFaces[2]; u32 nextItem; u32 pad[3]; } sh;
	static_assert(sizeof(sh) % 16 == 0, "static LDS in front of the dynamic region");
	u32* const wgStats = sh.wgStats; u32* const scanScratch = sh.scanScratch; u32* const zeroFlag = sh.zeroFlag; u32* const quietFaces = sh.quietFaces;

	const int tid0 = (int)threadIdx.x;
	if (tid0 < 20) wgStats[tid0] = 0;
	if (tid0 < 2) { zeroFlag[tid0] = 0; quietFaces[tid0] = 0; }
	// the queue's segments (the slot counts of all levels are final since the classification): matEnd[l] = end of level l's
	// material items; the regular and the transition items are prefixes of the same order
	u32 matEnd[MAX_LEVELS];
	u32 run = 0;
#pragma unroll
	for (u32 l = 0; l < MAX_LEVELS; ++l) {
		if (l >= 1 && l < plan.levels) run += p.G.slotCounts[l];
		matEnd[l] = r0_uniform(run);
	}
	const u32 matTotal = matEnd[MAX_LEVELS - 1];
	u32 regTotal = 0, trTotal = 0;
#pragma unroll
	for (u32 l = 1; l < MAX_LEVELS; ++l) {
		if (l < plan.fastEnd) regTotal = matEnd[l];
		if (l < plan.levels && p.levels[l].hasTransitions) trTotal = matEnd[l];
	}
	const u32 total = matTotal + regTotal + trTotal;
	// A workgroup takes its share of the queue and leaves: workgroups that stay for the whole launch keep the LDS they
	// got at its start - before the level-0 pass on the other stream has any - to the end, and the two then run one after
	// the other instead of side by side (1024^3: 0.54 instead of 0.44 ms per step).  Short-lived ones hand their place back.
	u32 quota = plan.persistent ? 0xFFFFFFFFu : (total + gridDim.x - 1u) / gridDim.x;

	const GridView& g = p.G.grid;
	const F1BrickSampler smp = { g.bDist, g.bMat, g.bBlend, g.n - 1, (u32)g.n >> 4, (u32)g.bRowsY, g.bYb0, g.bZb0 };
	u32 tabKind = 0;          // which table image the LDS holds: 0 none, 1 regular (F0), 2 transition
	F0Tables FT = {};
	Tables TT = {};
	u32 parity = 0, quietParity = 0;

	for (; quota; --quota) {
		// ---- dequeue (one returning atomic per item; a few thousand items per run) -----------------------------------------
		__syncthreads(); // the previous item is done (with the LDS state, and with `nextItem`)
		if (tid0 == 0) sh.nextItem = atomicAdd(p.G.upperHead, 1u);
		__syncthreads();
		const u32 item = r0_uniform(sh.nextItem);
		if (item >= total) break;
		int tid = tid0;
		asm volatile("" : "+v"(tid)); // (per item: what a lane derives from its index alone is not hoisted out of the loop and kept in registers)

		// item -> (kind, level, slot): the position inside its segment, looked up in the level boundaries
		const bool isMat = item < matTotal, isReg = !isMat && item < matTotal + regTotal;
		const u32 f = isMat ? item : (isReg ? item - matTotal : item - matTotal - regTotal);
		u32 level = 1, base = 0;
#pragma unroll
		for (u32 l = 1; l + 1 < MAX_LEVELS; ++l) if (f >= matEnd[l]) { level = l + 1; base = matEnd[l]; }
		const u32 slot = f - base;
		if (isMat) {
			mat_block<true>(p, level, slot, *(MatLds*)state, tid);
			continue;
		}
		const LevelDesc& L = p.levels[level];
		const u32 coord = r0_uniform(L.slotCoord[slot]);
		if (isReg) {
			if (tabKind != 1u) { FT = f0_stage_tables(tab, p.tables, (u32)tid); tabKind = 1u; } // (behind the barriers of the dequeue; visible after the block's first barrier)
			f1_block<REG_CAP_SMALL, true>(p, FT, smp, *(Fast1State<REG_CAP_SMALL>*)state, wgStats, zeroFlag, parity, level, slot, coord, 0u, 0u, tid);
		} else {
			if (tabKind != 2u) { TT = stage_transition_tables(tab, p.tables, (u32)tid); tabKind = 2u; }
			RegBlockCtx b;
			b.level = level; b.slot = slot;
			tr_block<false, true>(p, b, coord, *(TrState*)state, TT, scanScratch, quietFaces, quietParity, smp, tid);
		}
	}
	__syncthreads();
	if (tid0 < 20 && wgStats[tid0]) atomicAdd(&p.G.stats[tid0], wgStats[tid0]);
}

} // namespace
