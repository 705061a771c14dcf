#!/usr/bin/env python3
"""bench.py — Mvoxels/s of one full polygonization (Polygonizer::Execute equivalent) of the 1024^3 procedural
terrain, LOD levels 0..3 with transition cells and materials, on N MI355X GPUs of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step = one vx_polygonize over the resident grid (classify -> hierarchy -> material -> regular -> transition
kernels, the device-built block lists, and the small header read-back that tells the host the counts).  With N > 1 the grid is sharded in slabs
(along y by default: a terrain's surface lives in a few z-layers; strong scaling: the 1024^3 grid is fixed); the slab
halo (1 distance layer down, 2 distance + 1 material + 1 blend layer up) is exchanged over RCCL through the C ABI
(vx_halo_exchange) once before the steps — a polygonization of an unchanged grid needs no exchange — or, with
--halo-every-step, inside every step, as the path does after an edit.  Inputs are generated on the device
(vx_grid_create_terrain / vx_grid_fill_terrain: the bytes of the host generator voxels_synth) and are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=int(os.environ.get("VOXELS_BENCH_N", "1024")))
    ap.add_argument("--levels", type=int, default=int(os.environ.get("VOXELS_BENCH_LEVELS", "4")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--slab-axis", choices=["y", "z"], default="y", help="axis along which the grid is cut into one slab per GPU (y balances height-field terrains, whose surface sits in a few z-layers)")
    ap.add_argument("--serialize", action="store_true", help="run EVERY launch of this process with the library's streams serialised (one kernel at a time), for per-kernel profiling: the end-to-end figures are skipped")
    ap.add_argument("--no-extra", action="store_true", help="skip the second ('caves') workload reported under config.extra")
    ap.add_argument("--no-isolated", action="store_true", help="skip the second context that times the parts of k_main as launches of their own (profiling runs: only the product path's kernels in the trace)")
    ap.add_argument("--require-c-abi-transport", action="store_true", help="N > 1 only: fail if the C-ABI RCCL communicator (vx_comm_init) cannot be brought up.  Default: the halo then moves through torch.distributed (RCCL as well) and the line says so in config.halo_transport / config.c_abi_rccl_error - such a line is no evidence for vx_halo_exchange, but the timed steps do not contain the exchange unless --halo-every-step")
    ap.add_argument("--allow-torch-transport", action="store_true", help=argparse.SUPPRESS)  # (the default since round 5; kept so that old command lines still parse)
    ap.add_argument("--dry-run-cpu", action="store_true", help=argparse.SUPPRESS)  # tests/test_bench_dry_run.py: the script's own control flow (rank bookkeeping, collectives, transports, self-check, the JSON line) on CPU - gloo instead of RCCL, whatever library VOXELS_HIP_LIBRARY names instead of the HIP one.  Its numbers mean nothing and the line says so.
    ap.add_argument("--halo-every-step", action="store_true", help="N > 1 only: exchange the slab halo inside every timed step (as after an edit) instead of once before the steps")
    return ap.parse_args()


def mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / (1 << 20)
    except OSError:
        pass
    return 0.0


def cpu_quota_cores():
    """CPUs' worth of time per scheduler period the container may use (cgroup v2 cpu.max / v1 cfs_quota_us), None without a quota."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline(n, seed, levels=4):
    """The unmodified reference (oracle/_ref) — or the port when _ref is absent — on the host cores: the SAME grid as the
    GPU run when the host has the memory for it (else a sub-world), one thread per CPU the process may use (affinity mask,
    container quota), plus a one-thread figure on a bounded
    sub-world.  The reference cannot limit LOD levels: it always produces log2(n/16)+1 of them (stated in `sample`).
    Reported baseline only; never part of the product path.  When the grid is the GPU run's, the first `levels` levels of
    what the reference produced are digested (voxels_amd/digest.py: every byte of every mesh, ids, corners, counts) so that
    the caller can compare them with the GPU run's: `_digest` in the returned dict (popped by the caller)."""
    import vxo
    from voxels_amd import digest, synth
    oracle = vxo.load_ref() or vxo.load_port()
    if oracle is None:
        return None
    host_cpus = os.cpu_count() or 1
    quota = cpu_quota_cores()
    # threads: the CPUs this process may actually use - its affinity mask, and the container's CPU bandwidth quota when there is
    # one (more threads than the quota pays for only get the whole process put to sleep until the scheduler's period ends)
    cores = max(1, min(host_cpus, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cpus, int(np.ceil(quota)) if quota else host_cpus))
    big = n if (mem_available_gb() >= 40 and cores >= 16) else (512 if cores >= 16 else 256)

    counts = {}

    def run(size, threads, reps, count=False):
        d, m, b = synth.terrain(size, 0, size, seed)
        g = oracle.grid_from_dense(d, m, b)
        del d, m, b
        best = None
        for r in range(reps):
            t = time.perf_counter()
            s = oracle.execute(g, threads=threads)
            dt = time.perf_counter() - t
            if count and r == reps - 1:  # what the reference produced (all its levels): checked against the drop-in run of the same grid
                lv = s.all_levels()
                counts["levels"] = len(lv)
                counts["verts"] = int(sum(len(l.verts) for l in lv))
                counts["indices"] = int(sum(len(l.idx) for l in lv))
                counts["_digest"] = digest.surface_digest(lv[:levels])
            s.destroy()
            best = dt if best is None else min(best, dt)
        return best

    t_all = run(big, cores, 2, count=(big == n))
    t_one = run(256, 1, 1)
    ref_digest = counts.pop("_digest", None)
    return {"_digest": ref_digest, "value": round(big ** 3 / t_all / 1e6, 3), "unit": "Mvoxels/s", "cores": cores, "host_cpus": host_cpus,
            "cpu_quota_cores": quota, "kind": oracle.kind, "counts": counts or None,
            "one_thread": {"value": round(256 ** 3 / t_one / 1e6, 3), "unit": "Mvoxels/s", "cores": 1,
                           "sample": "256^3 sub-world of the same seeded terrain, all 5 reference LOD levels, 1 run of %.1f s" % t_one},
            "sample": "%s of the same seeded terrain, all %d reference LOD levels (the reference cannot limit levels; the GPU "
                      "run produces the %s finest), Polygonizer::Execute only, best of 2, %.2f s per run"
                      % ("the same %d^3 grid" % big if big == n else "%d^3 sub-world" % big, int(np.log2(big // 16)) + 1, "4", t_all)}


def dropin_e2e(poly):
    """Polygonizer::Execute through the drop-in C++ API (libVoxels.so): the resident grid is written out as the
    reference's grid file, a separate process loads it with Grid::Load and times Execute (upload of the packed grid,
    kernels, lists, download of every level into PolygonBlock vectors)."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "tools", "dropin_bench")
    if not os.path.exists(exe):
        return None
    blob = poly.pack()
    with tempfile.NamedTemporaryFile(suffix=".vxgrid", delete=False) as f:
        f.write(blob.tobytes())
        path = f.name
    try:
        # (VOXELS_PREWARM_MB: InitializeVoxels page-locks one mesh arena of that size ahead of the first Execute - reported as
        # initialize_ms, next to execute_ms_first)
        r = subprocess.run([exe, path, "3"], capture_output=True, text=True, timeout=600, env=dict(os.environ, VOXELS_TRACE="1", VOXELS_PREWARM_MB=os.environ.get("VOXELS_PREWARM_MB", "640")))
        if r.returncode != 0 or not r.stdout.strip():
            return {"error": (r.stderr or "rc %d" % r.returncode)[-200:]}
        out = json.loads(r.stdout.strip().splitlines()[-1])
        # VOXELS_TRACE: where the last Execute spent its time (grid to device | vx_polygonize | meshes to host)
        laps = [l.split("]", 1)[1].rsplit(None, 2) for l in r.stderr.splitlines() if l.startswith("[Voxels]")]
        out["last_run_breakdown_ms"] = {what.strip(): float(ms) for what, ms, _ in laps[-3:]}
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[-200:]}
    finally:
        os.unlink(path)


def main():
    args = parse()
    import torch
    from voxels_amd import Polygonizer, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    dry = args.dry_run_cpu
    if not dry:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    dist_pkg = None
    if world > 1:
        import torch.distributed as dist_pkg
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if dry:
            dist_pkg.init_process_group("gloo")
        else:
            dist_pkg.init_process_group("nccl", device_id=dev)

    def device_sync():
        if not dry:
            torch.cuda.synchronize()

    n, levels, seed = args.n, args.levels, 1337
    coarse = 16 << (levels - 1)
    assert n % (coarse * world) == 0, "slabs must hold whole blocks of the coarsest level"
    planes = n // world
    z0, z1 = rank * planes, (rank + 1) * planes

    # ---- the grid is generated where it lives (vx_grid_create_terrain / vx_grid_fill_terrain: the host generator's bytes,
    #      checked in tests/) — nothing crosses PCIe.  One rank owns the whole grid; N ranks attach one slab each (torch
    #      tensors with the halo layers the path reads) and exchange halos through the C ABI over RCCL. -------------------
    axis = args.slab_axis if world > 1 else "z"
    poly = Polygonizer(device=local_rank)
    assert dry or poly.backend == "hip:gfx950"
    if not dry:
        poly.set_stream(torch.cuda.current_stream().cuda_stream)
    poly.set_materials(synth.default_lut())
    t_gen = time.perf_counter()
    slab = None
    halo_transport = None
    comm_error = None
    if world == 1:
        poly.create_terrain(n, seed)
    else:
        from voxels_amd.slab import SlabBuffers
        slab = SlabBuffers(torch, n, rank, world, dev, axis=axis)
        slab.attach(poly)
        if dry:
            # (no generator on the device: the host generator's bytes, which vx_grid_fill_terrain reproduces - tests/)
            d, m, b = synth.terrain(n, seed=seed)
            sl = (slice(slab.z0, slab.z1),) if axis == "z" else (slice(None), slice(slab.z0, slab.z1))
            slab.fill_own(np.ascontiguousarray(d[sl]), np.ascontiguousarray(m[sl]), np.ascontiguousarray(b[sl]),
                          synth.block_empty_flags(d[sl]) if axis == "z" else synth.block_empty_flags(d))
            slab.attach(poly)
        else:
            poly.fill_terrain(seed)
        # The exchange runs through the C ABI (vx_comm_init / vx_halo_exchange: RCCL bound by the library itself).  Every rank
        # walks through the SAME sequence of collectives whatever fails where: rank 0 broadcasts a status byte + the id
        # (zeros when it could not make one), all ranks agree (MIN) before anybody enters ncclCommInitRank, and agree again
        # on its outcome.  A failure is reported in the line (config.halo_transport, config.c_abi_rccl_error) and on stderr;
        # --require-c-abi-transport makes it end the run.
        msg = np.zeros(129, np.uint8)
        why = ""
        if rank == 0:
            try:
                msg[1:] = poly.comm_unique_id()
                msg[0] = 1
            except Exception as e:  # noqa: BLE001
                why = str(e)
        uid = torch.from_numpy(msg).to(dev)
        dist_pkg.broadcast(uid, 0)
        msg = uid.cpu().numpy()
        ok = torch.tensor([int(msg[0])], dtype=torch.int32, device=dev)
        dist_pkg.all_reduce(ok, op=dist_pkg.ReduceOp.MIN)
        if int(ok.item()) == 1:
            try:
                poly.comm_init(world, rank, np.ascontiguousarray(msg[1:]))
            except Exception as e:  # noqa: BLE001
                why = str(e)
                ok = torch.tensor([0], dtype=torch.int32, device=dev)
            dist_pkg.all_reduce(ok, op=dist_pkg.ReduceOp.MIN)
        if int(ok.item()) == 1:
            halo_transport = "c-abi-rccl"
        elif args.require_c_abi_transport:
            raise SystemExit("rank %d: the C-ABI RCCL communicator could not be brought up (%s)" % (rank, why or "another rank failed"))
        else:
            # (loud, and in the line: a SCALE record made this way measures the sharded polygonization, not vx_halo_exchange)
            comm_error = why or "another rank failed"
            sys.stderr.write("rank %d: C-ABI RCCL communicator unavailable (%s): torch.distributed moves the halo, the line says so\n" % (rank, comm_error))
            halo_transport = "torch-distributed"
    device_sync()
    t_gen = time.perf_counter() - t_gen

    def halo_exchange():
        """1 distance layer from the slab below; 2 distance layers + 1 material + 1 blend layer and the flags of the
        boundary block layers from the slab above — vx_halo_exchange: pack kernel, one grouped ncclSend/ncclRecv batch on the
        library's stream, unpack kernel; no host wait."""
        if world > 1:
            if halo_transport == "c-abi-rccl":
                poly.halo_exchange()
            else:
                slab.gather_flags(dist_pkg)
                slab.halo_exchange(dist_pkg)
                slab.attach(poly)

    halo_exchange()
    device_sync()

    if args.serialize:
        poly.set_stage_timing(True)

    def step():
        # a polygonization of an unchanged grid needs no exchange (the halo is resident since the one above); with
        # --halo-every-step every step pays it, as a run after an edit would
        if args.halo_every_step:
            halo_exchange()
        return poly.execute(levels)

    def barrier():
        if world > 1:
            dist_pkg.barrier()
        device_sync()

    for _ in range(args.warmup):
        info = step()
    barrier()
    t0 = time.perf_counter()
    run_dev_ms = 0.0
    for _ in range(args.steps):
        info = step()
        run_dev_ms += info.device_ms
    barrier()
    run_dev_ms /= max(args.steps, 1)
    if dry:
        run_dev_ms = run_dev_ms or 1.0  # (the emulation reports no device time)
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist_pkg.all_reduce(te, op=dist_pkg.ReduceOp.MAX)
        elapsed = float(te.item())
    ms_per_step = elapsed / args.steps * 1e3

    # N > 1: the other variant as well - a step that includes the halo exchange (what a run after an edit pays) when the
    # timed steps above ran without it, and the other way round.  Every rank runs the same sequence (no collective is
    # conditional on a rank's own state).
    ms_other = None
    if world > 1:
        k = max(1, min(args.steps, 10))
        barrier()
        t1 = time.perf_counter()
        for _ in range(k):
            if not args.halo_every_step:
                halo_exchange()
            poly.execute(levels)
        barrier()
        to = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist_pkg.all_reduce(to, op=dist_pkg.ReduceOp.MAX)
        ms_other = float(to.item()) / k * 1e3

    # ---- N > 1: the correctness bit of a sharded run.  Every rank digests the blocks it produced
    #      (voxels_amd/digest.py: per-level totals + an order-independent 64-bit hash over every byte of every mesh, the block
    #      ids, corners and counts), the digests are summed over the ranks, and rank 0 polygonizes the WHOLE grid in a second
    #      context on its own GPU (6.6 GB at 1024^3) and compares.  A difference ends the run with a non-zero exit code on
    #      every rank: a SCALE record either carries "multi_gpu_check": "equal" or does not exist. ---------------------------
    multi_gpu_check = None
    if world > 1:
        from voxels_amd import digest
        poly.execute(levels)
        mine = digest.surface_digest(poly.all_levels())
        tsum = torch.from_numpy(digest.pack(mine)).to(dev)
        dist_pkg.all_reduce(tsum, op=dist_pkg.ReduceOp.SUM)
        summed = digest.unpack(tsum.cpu().numpy(), mine[0].shape[0])  # (the levels the run produced: fewer than requested on small grids)
        # The levels whose block is larger than a slab (1024^3 on 8 ranks: 4, 5, 6) - the reference always produces
        # log2(n / 16) + 1 - come from rank 0: the ranks' own layers gathered once (CoarseLevels, voxels_amd/slab.py: every rank
        # walks through the same collectives), a partial run over them.  Only when the slabs ran every level they can.
        from voxels_amd.slab import CoarseLevels, sharded_levels
        coarse_info = None
        all_levels_run = mine[0].shape[0] == sharded_levels(n, world)

        def make_coarse_poly():
            q = Polygonizer(device=local_rank)
            q.set_materials(synth.default_lut())
            return q
        t_coarse = time.perf_counter()
        coarse = CoarseLevels(slab, dist_pkg, make_coarse_poly) if all_levels_run else None
        verdict = torch.zeros(1, dtype=torch.int32, device=dev)
        if rank == 0:
            # (whatever happens here, rank 0 reaches the broadcast below: the other ranks are waiting in it)
            try:
                upper = coarse.execute() if coarse is not None else []
                device_sync()
                t_coarse = time.perf_counter() - t_coarse
                if upper:
                    summed = digest.join(summed, digest.surface_digest(upper, first_level=coarse.first))
                    coarse_info = {"levels": [coarse.first, coarse.levels], "gather_and_run_ms": round(t_coarse * 1e3, 2), "device_ms": round(float(coarse.info.device_ms), 4),
                                   "first_meshed_level": int(coarse.info.first_meshed_level),
                                   "note": "levels whose block is larger than a slab, on rank 0 from the gathered fields (vx_polygonize_from); outside the timed steps"}
                whole_poly = Polygonizer(device=local_rank)
                whole_poly.set_materials(synth.default_lut())
                if dry:
                    d, m, b = synth.terrain(n, seed=seed)
                    whole_poly.upload(d, m, b, synth.block_empty_flags(d))
                else:
                    whole_poly.create_terrain(n, seed)
                whole_poly.execute(0 if upper else levels)
                ref = digest.surface_digest(whole_poly.all_levels())
                whole_poly.close()
                equal = digest.digests_equal(summed, ref)
                verdict[0] = 1 if equal else 0
                multi_gpu_check = {"result": "equal" if equal else "DIFFERENT",
                                   "kind": "self-consistency of the sharded run (the whole-grid run of the SAME library is the reference here; the "
                                           "whole-grid run is what tests/ and parity_vs_reference compare with the oracle)",
                                   "totals_per_level_blocks_verts_indices_tverts_tindices": summed[0].tolist(),
                                   "hash": "%016x" % int(summed[1]), "reference": "whole-grid run of the same library on rank 0's GPU",
                                   "reference_totals": ref[0].tolist(), "reference_hash": "%016x" % int(ref[1]), "coarse_levels": coarse_info}
            except Exception as e:  # noqa: BLE001
                verdict[0] = 0
                multi_gpu_check = {"result": "ERROR", "error": str(e)[-300:]}
        if coarse is not None:
            coarse.close()
        dist_pkg.broadcast(verdict, 0)
        if int(verdict.item()) != 1:
            if rank == 0:
                sys.stderr.write("multi-GPU self-check FAILED: %s\n" % json.dumps(multi_gpu_check))
            dist_pkg.destroy_process_group()
            raise SystemExit(3)

    # ---- per-kernel device timing (HIP events on the stream the kernels run on).  With stage timing enabled the
    #      library serialises its streams so that every kernel's duration is its own; the timed steps above ran the
    #      normal, overlapped pipeline (level-0 regular pass and transition pass beside the material chain). ----
    if not dry:
        poly.set_stage_timing(True)
    stage = np.zeros(8, np.float64)
    reps = 5
    dev_ms = 0.0
    for _ in range(reps):
        info = poly.execute(levels)
        if not dry:
            stage += poly.stage_times()
        dev_ms += info.device_ms
    stage /= reps
    dev_ms /= reps
    if dry:
        stage[:] = 1.0  # (no stage timing on CPU: placeholders, the line is marked dry_run)
        dev_ms = dev_ms or 1.0
    if not args.serialize:
        poly.set_stage_timing(False)
    per_level = []
    for l in range(info.levels):
        lv = poly.level(l, with_data=False)
        per_level.append([int(lv.infos["n_verts"].sum()), int(lv.infos["n_idx"].sum()), int(lv.infos["n_tverts"].sum()), int(lv.infos["n_tidx"].sum())])
    totals = np.array(per_level, np.uint64).sum(axis=0)
    v0, i0 = per_level[0][0], per_level[0][1]
    slab_voxels = n * n * planes
    surface_blocks = int(info.active_blocks[0])
    blocks_read = int(info.blocks_read)
    # algorithmic bytes per kernel (SURVEY.md §8(d) shares): the distance samples of the blocks the flags and sign summaries
    # do not already prove surface-free (the full n^3 figure is kept beside it, see whole_execute), material + blend of the
    # surface blocks of level 0 and every mesh - all of it k_main's in a single-stream run; k_classify's share in the chain
    # of launches (VX_UPPER=0 / dense surfaces), where the regular pass is not charged the distances again.
    out_bytes_all = 48 * (int(totals[0]) + int(totals[2])) + 4 * (int(totals[1]) + int(totals[3]))
    # (single-stream runs have no classification pass: k_run_head hands out the slots from the flags and the sign summaries,
    # and k_main is the one reader of the distance samples - of the blocks_read blocks that are not proven surface-free)
    alg = {"k_main": float(4096 * blocks_read + 2 * 4096 * surface_blocks + out_bytes_all)}
    stage_names = ["k_reset+k_run_head", "-", "--", "k_main", "k_after_level0", "k_after_upper", "---", "k_lists"]
    stage_ms = {k: round(float(v), 4) for k, v in zip(stage_names, stage) if not k.startswith("-")}
    single_stream = poly.stage_layout() == 1
    if not single_stream:  # (a configuration without k_main: the chain of launches, VX_UPPER=0)
        stage_names = ["reset", "k_classify", "k_hierarchy", "k_material", "k_regular0", "k_regular", "k_transition", "k_lists"]
        stage_ms = {k: round(float(v), 4) for k, v in zip(stage_names, stage)}
        alg = {"k_classify": float(4096 * blocks_read),
               "k_regular0": float(2 * 4096 * surface_blocks + 48 * v0 + 4 * i0),
               "k_regular": float(48 * (int(totals[0]) - v0) + 4 * (int(totals[1]) - i0)),
               "k_transition": float(48 * int(totals[2]) + 4 * int(totals[3]))}
    dominant = max(alg.keys(), key=lambda k: stage_ms[k])
    achieved = alg[dominant] / (stage_ms[dominant] * 1e-3) / 1e9
    # The per-block bodies of k_main also exist as launches of their own (incremental runs, capacity classes): timed one at a
    # time in a second context (VX_UPPER=0: the chain of launches k_main replaces) they show what each part of k_main costs
    # alone - the figures earlier rounds reported as k_regular0 / k_regular / k_transition
    isolated = None
    if single_stream and world == 1 and not args.serialize and not args.no_isolated:
        try:
            os.environ["VX_UPPER"] = "0"
            iso = Polygonizer(device=local_rank)
            del os.environ["VX_UPPER"]
            iso.set_materials(synth.default_lut())
            iso.create_terrain(n, seed)
            iso.set_stage_timing(True)
            ist = np.zeros(8, np.float64)
            for _ in range(2):
                iso.execute(levels)
            for _ in range(reps):
                iso.execute(levels)
                ist += iso.stage_times()
            ist /= reps
            iso.close()
            inames = ["reset", "k_classify", "k_hierarchy", "k_material", "k_regular0", "k_regular", "k_transition", "k_lists"]
            ialg = {"k_regular0": float(2 * 4096 * surface_blocks + 48 * v0 + 4 * i0),
                    "k_regular": float(48 * (int(totals[0]) - v0) + 4 * (int(totals[1]) - i0)),
                    "k_transition": float(48 * int(totals[2]) + 4 * int(totals[3])), "k_material": 0.0}
            isolated = {k: {"ms": round(float(v), 4), "algorithmic_bytes": ialg.get(k), "frac": (round(ialg[k] / (float(v) * 1e-3) / 8e12, 5) if ialg.get(k) and v > 0 else None)}
                        for k, v in zip(inames, ist) if k in ialg}
        except Exception as e:  # noqa: BLE001
            isolated = {"error": str(e)[-200:]}
        finally:
            os.environ.pop("VX_UPPER", None)
    # HBM bytes per launch from rocprofv3 PMC passes of this same command, when a summary was committed
    traffic = None
    traffic_all = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            pm = json.load(open(pmc_path))
            if pm.get("n") == n and pm.get("levels") == levels and pm.get("gpus") == world:
                traffic = pm.get("hbm_bytes_per_launch", {}).get(dominant)
                traffic_all = pm.get("hbm_bytes_per_launch")
                if single_stream and traffic_all:
                    # (the profiled command also runs the serialised stages and a cold run, whose launches of the per-block
                    # kernels show up in the summary with a few MB per execute: a product run is these three launches)
                    traffic_all = {k: v for k, v in traffic_all.items() if k in ("k_run_head", "k_main", "k_tail")}
        except Exception:
            traffic = None
    roofline = {"kernel": dominant, "bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": alg[dominant], "avg_launch_ms": stage_ms[dominant],
                "per_kernel": {k: {"ms": stage_ms[k], "algorithmic_bytes": alg[k], "frac": round(alg[k] / (stage_ms[k] * 1e-3) / 8e12, 5) if stage_ms[k] > 0 else None} for k in alg},
                "parts_of_k_main_as_launches_of_their_own": isolated}
    outputs = 48 * (int(totals[0]) + int(totals[2])) + 4 * (int(totals[1]) + int(totals[3]))
    bytes_nominal = int(info.algorithmic_bytes)                               # §8(d): n^3 + 2*4096*surface blocks + outputs
    bytes_needed = 4096 * blocks_read + 2 * 4096 * surface_blocks + outputs  # with only the blocks that have to be read
    whole = {"algorithmic_bytes": bytes_nominal, "bytes_actually_read": bytes_needed,
             "blocks_read": blocks_read, "blocks_total": (n // 16) ** 2 * (planes // 16),
             "device_ms": round(run_dev_ms, 4), "device_ms_serialized": round(dev_ms, 4),
             "achieved_GBps_nominal": round(bytes_nominal / (run_dev_ms * 1e-3) / 1e9, 2),
             "frac_of_8TBps_nominal": round(bytes_nominal / (run_dev_ms * 1e-3) / 8e12, 5),
             "achieved_GBps": round(bytes_needed / (run_dev_ms * 1e-3) / 1e9, 2),
             "frac_of_8TBps": round(bytes_needed / (run_dev_ms * 1e-3) / 8e12, 5),
             "line_traffic_bytes_per_kernel": traffic_all,
             "line_traffic_bytes": int(sum(traffic_all.values())) if traffic_all else None,
             "note": "frac_of_8TBps counts the bytes a run has to touch (distance samples of the blocks that are not proven "
                     "surface-free by their flags, material+blend of surface blocks, outputs); the *_nominal figures use "
                     "SURVEY.md §8(d)'s n^3 term although most of it is never read"}

    # ---- end to end: what a caller waits for --------------------------------------------------------------
    def timed(fn, reps=3):
        best = None
        for _ in range(reps):
            device_sync()
            t = time.perf_counter()
            fn()
            device_sync()
            dt = (time.perf_counter() - t) * 1e3
            best = dt if best is None else min(best, dt)
        return round(best, 3)

    def run_and_lists():
        poly.execute(levels)
        for l in range(levels):
            poly.level(l, with_data=False)
            poly.level_ranges(l)

    def run_and_download():
        # every mesh on the host, addressable per block: both pools in one DMA into a (recycled) page-locked arena
        # (vx_host_meshes_acquire) + per level the block infos and the blocks' ranges in the pools
        poly.execute(levels)
        hm = poly.host_meshes()
        for l in range(levels):
            poly.level(l, with_data=False)
            poly.level_ranges(l)
        got = (hm.verts.size, hm.indices.size)
        hm.release()
        return got

    def run_and_copy_levels():
        # the older interface: per level four caller-owned arrays (vx_download_level), i.e. one more host copy
        poly.execute(levels)
        poly.all_levels()

    # (--serialize: a profiling run; no overlapped launch may end up in its kernel trace)
    e2e = None if args.serialize else {
        "polygonize_ms": timed(lambda: poly.execute(levels)),
        "polygonize_plus_host_block_lists_ms": timed(run_and_lists),
        "polygonize_plus_download_of_all_meshes_ms": timed(run_and_download, 3),
        "polygonize_plus_vx_download_level_copies_ms": timed(run_and_copy_levels, 2),
        "mesh_bytes": int(poly.info.total_verts) * 48 + int(poly.info.total_indices) * 4}

    # ---- what the timed step leaves out: the library's mirrors of the grid (brick order, lattice copies, sign summaries)
    #      are built where the grid changes, not by a polygonization.  A caller that changes the whole grid and runs once
    #      pays them: cold = mirrors + run, measured by declaring the resident grid changed (vx_grid_invalidate). ----------
    cold = None
    if not args.serialize:
        best = None
        for _ in range(3):
            poly.invalidate()
            device_sync()
            t = time.perf_counter()
            ci = poly.execute(levels)
            device_sync()
            wall = (time.perf_counter() - t) * 1e3
            if best is None or wall < best[0]:
                best = (wall, float(ci.mirror_ms) or (1.0 if dry else 0.0), float(ci.device_ms))
        vox = n * n * planes
        mirror_bytes = 6 * vox + sum(vox >> (3 * l) for l in range(1, 4))  # three fields read + written, lattice copies of levels 1..3 written
        cold = {"cold_execute_ms": round(best[0], 4), "mirror_build_ms": round(best[1], 4), "device_ms_after_mirrors": round(best[2], 4),
                "Mvoxels_per_s_cold": round(n ** 3 / (best[0] * 1e-3) / 1e6, 2),
                "mirror_roofline": {"kernel": "k_rebrick", "bytes": int(mirror_bytes), "achieved_GBps": round(mirror_bytes / (best[1] * 1e-3) / 1e9, 1),
                                    "frac_of_8TBps": round(mirror_bytes / (best[1] * 1e-3) / 8e12, 4)},
                "note": "the headline step runs on a resident, unchanged grid (SURVEY.md §8(d)); after a change of the whole grid "
                        "the first run also rebuilds the mirrors - the one place where all n^3 samples of the three fields are read"}

    # ---- the same step with the inputs NOT resident in the 256 MiB memory-side cache: the timed loop above polygonizes one
    #      grid over and over, so the ~0.4 GB of bricks a run reads may be served by the Infinity Cache (FETCH_SIZE counts
    #      fabric requests, MALL hits included).  Here two resident grids of different seeds (two contexts, two sets of
    #      mirrors and pools) are polygonized alternately: between two runs on a grid the other run moves ~1 GB (its inputs +
    #      its meshes) through the memory side.  Reported beside the headline; see DESIGN.md §6. ------------------------------
    uncached = None
    if world == 1 and not args.serialize and not args.no_extra:
        try:
            other = Polygonizer(device=local_rank)
            other.set_stream(torch.cuda.current_stream().cuda_stream)
            other.set_materials(synth.default_lut())
            other.create_terrain(n, seed + 1)
            for _ in range(3):
                poly.execute(levels); oi = other.execute(levels)
            device_sync()
            k = max(10, min(args.steps, 50))
            ta = time.perf_counter()
            for _ in range(k):
                poly.execute(levels); other.execute(levels)
            device_sync()
            alt_ms = (time.perf_counter() - ta) / (2 * k) * 1e3
            # the second grid alone, back to back (its surface differs a little from the headline grid's: the pair's expected mean)
            tb = time.perf_counter()
            for _ in range(k):
                other.execute(levels)
            device_sync()
            other_ms = (time.perf_counter() - tb) / k * 1e3
            other.close()
            expected = 0.5 * (ms_per_step + other_ms)
            uncached = {"ms_per_step_alternating_two_grids": round(alt_ms, 4), "ms_per_step_second_grid_alone": round(other_ms, 4),
                        "ms_per_step_headline_grid_alone": round(ms_per_step, 4), "slowdown_vs_mean_of_the_two_alone": round(alt_ms / expected, 4),
                        "second_grid": {"seed": seed + 1, "active_blocks": [int(x) for x in oi.active_blocks[:levels]], "verts": int(oi.total_verts), "indices": int(oi.total_indices)},
                        "Mvoxels_per_s": round(n ** 3 / (alt_ms * 1e-3) / 1e6, 2),
                        "note": "two resident %d^3 grids (seeds %d, %d) in two contexts, polygonized alternately: no input line of a run was "
                                "touched since the other grid's run moved its own inputs and ~0.5 GB of meshes through the memory side" % (n, seed, seed + 1)}
        except Exception as e:  # noqa: BLE001
            uncached = {"error": str(e)[-300:]}

    # ---- BASELINE config 5: 512^3 terrain, sphere carves (IT_Subtract, r = 20) at the surface, each followed by the incremental
    #      re-polygonization of its dirty box (vx_grid_inject_ball + vx_polygonize_dirty: three launches - k_dirty_head |
    #      k_main<true> | k_dirty_tail).  All reference levels (6 at 512^3).  Parity of this path: tests/test_gpu_parity.py. ------
    edit = None
    if world == 1 and not args.serialize and not args.no_extra:
        try:
            en = 512
            ep = Polygonizer(device=local_rank)
            ep.set_stream(torch.cuda.current_stream().cuda_stream)
            ep.set_materials(synth.default_lut())
            ep.create_terrain(en, seed)
            ei0 = ep.execute(0)
            ep.level(0, with_data=False)  # (the host copy of the block lists: fetched once after a full run)
            # (the surface's height from the resident grid: generating the grid on the host a second time would be a burst of
            # host threads right in front of the timed calls - in a container with a CPU quota that burst gets every thread of the
            # process put to sleep for the rest of a 100 ms period, profiles/r05_edit_stalls.txt)
            col = ep.column(en, en // 2, en // 2)
            zs = float(np.argmax(col >= 0)) if (col >= 0).any() else en * 0.5
            calls, devs, rebuilt, edit_ms, bytes_alg = [], [], [], [], []
            pv, pi = int(ei0.total_verts), int(ei0.total_indices)
            for k in range(14):
                pos = (en / 2.0 + 23.0 * (k % 4) - 30.0 + 0.37, en / 2.0 + 19.0 * (k // 4) - 20.0 + 0.61, zs + 2.0 * (k % 3) + 0.23)
                device_sync()
                t = time.perf_counter(); mn, mx = ep.inject_ball(pos, (44.0, 44.0, 44.0), 20.0, 2); device_sync(); te = time.perf_counter() - t
                t = time.perf_counter(); got = ep.execute_dirty(mn, mx); dt = time.perf_counter() - t
                nv, ni = int(ep.info.total_verts), int(ep.info.total_indices)
                if k >= 2:
                    calls.append(dt * 1e3); devs.append(float(ep.info.device_ms)); rebuilt.append(int(got.size)); edit_ms.append(te * 1e3)
                    # the dirty level-0 blocks' three fields + the meshes the run appended (a compaction in between resets the cursors: skipped)
                    if nv >= pv and ni >= pi:
                        bytes_alg.append(3 * 4096 * int(ep.info.active_blocks[0]) + 48 * (nv - pv) + 4 * (ni - pi))
                pv, pi = nv, ni
            ep.close()
            dev_ms_e = float(np.median(devs))
            balg = float(np.median(bytes_alg)) if bytes_alg else 0.0
            edit = {"workload": "512^3 terrain (seed %d), all %d reference levels, chain of IT_Subtract ball carves r = 20 at the surface (centres off the lattice), "
                                "each followed by vx_polygonize_dirty of its box" % (seed, int(ei0.levels)),
                    "ms_per_call_median": round(float(np.median(calls)), 4), "ms_per_call_mean": round(float(np.mean(calls)), 4), "ms_per_call_best": round(float(np.min(calls)), 4),
                    "ms_per_call_worst": round(float(np.max(calls)), 4), "device_ms_median": round(dev_ms_e, 4), "blocks_rebuilt_per_call": round(float(np.mean(rebuilt)), 1),
                    "device_edit_ms_median": round(float(np.median(edit_ms)), 4), "launches_per_call": 3,
                    "roofline": {"kernels": "k_dirty_head | k_main<true> | k_dirty_tail", "bound": "hbm", "algorithmic_bytes_per_call": balg,
                                 "achieved": round(balg / (dev_ms_e * 1e-3) / 1e9, 2) if dev_ms_e > 0 else None, "peak": 8000.0, "unit": "GB/s",
                                 "frac": round(balg / (dev_ms_e * 1e-3) / 8e12, 5) if dev_ms_e > 0 else None,
                                 "note": "a few hundred blocks: the three launches are bound by their dependent steps (five material levels "
                                         "one after the other, then regular and transition blocks), not by bytes"}}
        except Exception as e:  # noqa: BLE001
            edit = {"error": str(e)[-300:]}

    # ---- a second workload whose figure does not rest on a sparse surface: the "caves" style of the generator puts surface
    #      into a large share of all blocks (parity-tested like the terrain, tests/test_gpu_parity.py) ---------------------
    extra = None

    def extra_workload():
        poly.create_terrain(n, seed, 1)
        for _ in range(2):
            xi = poly.execute(levels)
        device_sync()
        t = time.perf_counter()
        k = 10
        for _ in range(k):
            xi = poly.execute(levels)
        device_sync()
        ms = (time.perf_counter() - t) / k * 1e3
        tot = np.zeros(4, np.uint64)
        lvl0 = None
        for l in range(xi.levels):
            lv = poly.level(l, with_data=False)
            row = np.array([lv.infos["n_verts"].sum(), lv.infos["n_idx"].sum(), lv.infos["n_tverts"].sum(), lv.infos["n_tidx"].sum()], np.uint64)
            lvl0 = row if lvl0 is None else lvl0
            tot += row
        sb = int(xi.active_blocks[0])
        out_bytes = 48 * (int(tot[0]) + int(tot[2])) + 4 * (int(tot[1]) + int(tot[3]))
        need = 4096 * int(xi.blocks_read) + 2 * 4096 * sb + out_bytes
        # the same roofline block as the headline workload's: serialised stage times, algorithmic bytes per kernel, the
        # dominant one priced against 8 TB/s (dense surfaces run the chain of launches with all capacity classes: a stage
        # is then a group of launches - the table-driven class, the 1536-cell class, the general pass in two classes)
        poly.set_stage_timing(True)
        st = np.zeros(8, np.float64)
        for _ in range(3):
            poly.execute(levels)
            st += poly.stage_times()
        st /= 3
        chain = poly.stage_layout() == 0
        poly.set_stage_timing(False)
        v0x, i0x = int(lvl0[0]), int(lvl0[1])
        if chain:
            names = ["reset", "k_classify", "k_hierarchy", "k_material", "k_regular0", "k_regular", "k_transition", "k_lists"]
            xalg = {"k_classify": float(4096 * int(xi.blocks_read)), "k_regular0": float(2 * 4096 * sb + 48 * v0x + 4 * i0x),
                    "k_regular": float(48 * (int(tot[0]) - v0x) + 4 * (int(tot[1]) - i0x)), "k_transition": float(48 * int(tot[2]) + 4 * int(tot[3]))}
        else:
            names = ["k_reset+k_run_head", "-", "--", "k_main", "k_after_level0", "k_after_upper", "---", "k_lists"]
            xalg = {"k_main": float(need)}
        xms = {k: round(float(v), 4) for k, v in zip(names, st) if not k.startswith("-")}
        xdom = max(xalg.keys(), key=lambda k: xms[k])
        xtraffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_caves_latest.json")))
            if pm.get("n") == n and pm.get("levels") == levels:
                xtraffic = pm.get("hbm_bytes_per_launch", {}).get(xdom)
        except Exception:
            xtraffic = None
        xach = xalg[xdom] / (xms[xdom] * 1e-3) / 1e9
        xroof = {"kernel": xdom, "bound": "hbm", "achieved": round(xach, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(xach / 8000.0, 5),
                 "traffic": xtraffic, "algorithmic_bytes_per_launch": xalg[xdom], "avg_launch_ms": xms[xdom],
                 "per_kernel": {k: {"ms": xms[k], "algorithmic_bytes": xalg[k], "frac": round(xalg[k] / (xms[k] * 1e-3) / 8e12, 5) if xms[k] > 0 else None} for k in xalg},
                 "stage_ms_serialized": xms}
        return {"roofline": xroof,"workload": "%d^3 'caves' style of the same generator (seed %d): 3-D noise isosurfaces in a band around the terrain height, "
                            "materials, LOD levels 0..%d with transition cells" % (n, seed, levels - 1),
                "ms_per_step": round(ms, 4), "Mvoxels_per_s": round(n ** 3 / (ms * 1e-3) / 1e6, 2),
                "surface_blocks": sb, "surface_block_share": round(sb / ((n // 16) ** 3), 4),
                "Mvoxels_per_s_over_surface_blocks": round(sb * 4096 / (ms * 1e-3) / 1e6, 2),
                "verts": int(tot[0]), "indices": int(tot[1]), "tverts": int(tot[2]), "tindices": int(tot[3]),
                "bytes_actually_touched": int(need), "achieved_GBps": round(need / (ms * 1e-3) / 1e9, 1), "frac_of_8TBps": round(need / (ms * 1e-3) / 8e12, 4)}

    if rank == 0:
        step_s = elapsed / args.steps
        out = {
            "metric": "Mvoxels/s polygonized (1024^3 grid, 4 LOD levels)" if (n == 1024 and levels == 4) else "Mvoxels/s polygonized (%d^3 grid, %d LOD levels)" % (n, levels),
            "value": round(n ** 3 / step_s / 1e6, 2),
            "unit": "Mvoxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "i8",
            "data": "synthetic",
            "dry_run": bool(dry),  # (true only for tests/test_bench_dry_run.py: the script's control flow on CPU, numbers meaningless)
            "config": {"workload": "%d^3 procedural noise terrain (seed %d), materials, LOD levels 0..%d with transition cells, "
                                   "%s-slab sharded over %d GPU(s)" % (n, seed, levels - 1, axis, world),
                       "grid": n, "levels": levels, "parallelism": "%sslab%d" % (axis, world),
                       "active_blocks": [int(x) for x in info.active_blocks[:levels]],
                       "verts": int(totals[0]), "indices": int(totals[1]), "tverts": int(totals[2]), "tindices": int(totals[3]),
                       "surface_only": {"surface_blocks_per_s": round(surface_blocks / step_s, 1),
                                        "Mvoxels_per_s_over_surface_blocks": round(surface_blocks * 4096 / step_s / 1e6, 2),
                                        "note": "a height-field terrain keeps its surface in %d of %d level-0 blocks; `value` counts every voxel of the grid, as the metric defines it" % (surface_blocks, (n // 16) ** 2 * (planes // 16))},
                       "stage_ms_serialized": stage_ms, "whole_execute": whole, "e2e_ms": e2e, "device_gen_s": round(t_gen, 3),
                       "halo_exchange_in_step": bool(world > 1 and args.halo_every_step), "halo_transport": halo_transport, "c_abi_rccl_error": comm_error,
                       "step_definition": "r03+: one vx_polygonize per step on a resident grid; with N > 1 the slab halo is exchanged once before "
                                          "the steps (vx_halo_exchange) unless --halo-every-step (r01/r02 exchanged it inside every step: "
                                          "compare those rounds with ms_per_step_with_halo_exchange)",
                       "multi_gpu_check": (multi_gpu_check["result"] if multi_gpu_check else None), "multi_gpu_check_detail": multi_gpu_check,
                       ("ms_per_step_without_halo_exchange" if args.halo_every_step else "ms_per_step_with_halo_exchange"): (round(ms_other, 4) if ms_other is not None else None),
                       "cold": cold, "steady_state_uncached": uncached, "edit": edit},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and not args.serialize:
            de = dropin_e2e(poly)
            if de:
                out["config"]["e2e_ms"]["libVoxels_Polygonizer_Execute"] = de
            cb = cpu_baseline(n, seed, levels)
            if cb:
                # SURVEY.md §8(c) at bench scale: the reference itself polygonized this very grid a moment ago - the digest of
                # its levels 0..levels-1 (every byte of every mesh incl. normals, block ids, corners, counts) must be the
                # digest of what the GPU path produces for the same grid.  A difference fails the bench.
                ref_digest = cb.pop("_digest", None)
                out["cpu_baseline"] = cb
                if ref_digest is not None:
                    from voxels_amd import digest
                    poly.create_terrain(n, seed)
                    poly.execute(levels)
                    mine = digest.surface_digest(poly.all_levels())
                    equal = digest.digests_equal(mine, ref_digest)
                    out["parity_vs_reference"] = "equal" if equal else "DIFFERENT"
                    out["config"]["parity_vs_reference_detail"] = {
                        "checked": "levels 0..%d of the %d^3 bench grid: GPU run vs the %s (oracle/_ref = the unmodified reference sources) run of cpu_baseline, "
                                   "order-independent 64-bit digest over every byte of every mesh (positions, secondary positions, normals, texture bytes, "
                                   "indices, transition meshes), block ids, corners and counts" % (levels - 1, n, cb["kind"]),
                        "totals_per_level_blocks_verts_indices_tverts_tindices": mine[0].tolist(), "hash": "%016x" % int(mine[1]),
                        "reference_totals": ref_digest[0].tolist(), "reference_hash": "%016x" % int(ref_digest[1])}
                    if not equal:
                        sys.stderr.write("parity vs reference FAILED: %s\n" % json.dumps(out["config"]["parity_vs_reference_detail"]))
                        print(json.dumps(out))
                        raise SystemExit(4)
                # the reference and the drop-in library polygonized the same grid in this run: their counts must agree
                rc = cb.get("counts")
                if rc and de and "verts" in de:
                    if (rc["levels"], rc["verts"], rc["indices"]) != (de["levels"], de["verts"], de["indices"]):
                        raise SystemExit("drop-in library and CPU reference disagree on the %d^3 grid: reference %s, libVoxels.so %s" % (n, rc, {k: de[k] for k in ("levels", "verts", "indices")}))
                    out["config"]["e2e_ms"]["libVoxels_Polygonizer_Execute"]["counts_equal_reference"] = True
        if world == 1 and not args.serialize and not args.no_extra:
            try:
                out["config"]["extra"] = extra_workload()
            except Exception as e:  # noqa: BLE001
                out["config"]["extra"] = {"error": str(e)[-300:]}
        print(json.dumps(out))
    if world > 1:
        dist_pkg.destroy_process_group()


if __name__ == "__main__":
    main()
