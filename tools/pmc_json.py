#!/usr/bin/env python3
"""profiles/pmc_latest.json from the per-pass summaries of tools/pmc_run.sh (FETCH_SIZE, WRITE_SIZE, TCC hit/miss passes).
Usage: tools/pmc_json.py <dir with pass1.txt pass2.txt [pass3.txt]> <n> <levels> <gpus> <label of the committed copies> [target json] [profiled command]"""
import json
import re
import sys

CLASSES = {"k_main": "k_main", "k_tail": "k_tail", "k_classify": "k_classify", "k_hierarchy": "k_classify", "k_material": "k_material", "k_transition": "k_transition",
           "k_regular0": "k_regular0", "k_regular": "k_regular", "k_run_head": "k_run_head", "k_reset": "k_run_head", "k_list": "k_lists"}


own_calls = {}  # of the last parse: dispatches of the kernels that run once per product run (the profiled command also runs a few
                # polygonizations through the chain of launches: those count as executes but launch no k_main)


def parse(path):
    """-> ({kernel class: {counter: sum}}, executes) ; executes = number of k_run_head dispatches (one per full run, whatever its pipeline)"""
    sums, executes = {}, 0
    own_calls.clear()
    for line in open(path):
        parts = line.split()
        if len(parts) == 5 and re.match(r"^[0-9.]+$", parts[-1]) and re.match(r"^\d+$", parts[2]):
            name, counter, samples, total = parts[0], parts[1], int(parts[2]), float(parts[3])
            for key, cls in CLASSES.items():
                if key in name:
                    if key == "k_regular" and "k_regular0" in name:
                        continue
                    sums.setdefault(cls, {}).setdefault(counter, 0.0)
                    sums[cls][counter] += total
                    if key == "k_run_head":
                        executes = samples
                    if key in ("k_main", "k_tail"):
                        own_calls[cls] = samples
    return sums, executes


def main():
    d, n, levels, gpus, label = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    target = sys.argv[6] if len(sys.argv) > 6 else "profiles/pmc_latest.json"
    command = sys.argv[7] if len(sys.argv) > 7 else "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-isolated"
    fetch, ex1 = parse(d + "/pass1.txt")
    write, ex2 = parse(d + "/pass2.txt")
    out = {"n": n, "levels": levels, "gpus": gpus,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum (separate passes, --kernel-trace only) of "
                     "`%s`; summaries committed as profiles/%s_pmc_*.txt" % (command, label),
           "units": "FETCH_SIZE/WRITE_SIZE are KiB, summed over the dispatches of a kernel and divided by the number of polygonizations (k_main, k_tail: by their own dispatches)",
           "correction": "calibrated on known access patterns (tools/pmc_calib.hip, profiles/r02a_pmc_calibration.txt): on gfx950 FETCH_SIZE "
                         "reports exactly half of the bytes of the 128-byte lines a kernel pulls in, for every access width tried (16 / 8 / 4 / 1 bytes "
                         "per lane, contiguous or strided up to one byte per line) - every kernel's FETCH_SIZE is doubled; WRITE_SIZE equals the bytes "
                         "written (16-byte, 4-byte and 48-byte-record stores) and is taken as reported."}
    fetch, ex1 = parse(d + "/pass1.txt")
    fk = {k: v.get("FETCH_SIZE", 0.0) / max(own_calls.get(k, ex1), 1) for k, v in fetch.items()}
    write, ex2 = parse(d + "/pass2.txt")
    wk = {k: v.get("WRITE_SIZE", 0.0) / max(own_calls.get(k, ex2), 1) for k, v in write.items()}
    out["fetch_kib_per_execute"] = {k: round(v, 1) for k, v in fk.items()}
    out["write_kib_per_execute"] = {k: round(v, 1) for k, v in wk.items()}
    out["read_bytes_per_launch"] = {k: int(fk[k] * 2.0 * 1024) for k in fk}
    out["write_bytes_per_launch"] = {k: int(wk.get(k, 0.0) * 1024) for k in fk}
    out["hbm_bytes_per_launch"] = {k: int((fk.get(k, 0.0) * 2.0 + wk.get(k, 0.0)) * 1024) for k in fk}
    try:
        tcc, _ = parse(d + "/pass3.txt")
        out["l2_hit_rate"] = {k: round(v.get("TCC_HIT_sum", 0.0) / max(v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 0.0), 1.0), 3) for k, v in tcc.items()}
    except OSError:
        pass
    json.dump(out, open(target, "w"), indent=1)
    print(json.dumps(out["hbm_bytes_per_launch"]))


if __name__ == "__main__":
    main()
