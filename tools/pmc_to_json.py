#!/usr/bin/env python3
"""Condense the rocprofv3 runs of tools/pmc_collect.sh into profiles/r03/:
  *_kernel_stats.csv   per-kernel duration statistics (from the kernel traces)
  pmc_counters.json    per-kernel counters per STEP + the FETCH_SIZE / WRITE_SIZE calibration, stamped
                       with the hash of the kernel sources (bench.py uses it only for the same sources)
usage: pmc_to_json.py <collect dir> <profiles dir> [frames per launch] [warm-up steps to leave out]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kname(s):
    return s.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()


STEP_KERNEL = "k_level_select"  # launched exactly once per step (batch): its dispatch count = steps of a run


SKIP_STEPS = 0  # warm-up steps of every run (argv[4]): their dispatches are left out -- the adaptive first pass
# re-partitions the FAST tile rows after the second batch (orbx_api.cpp, adapt_tile_rows)


def drop_warmup(rows, steps):
    """rows: a kernel's dispatches in dispatch order; drops the first SKIP_STEPS steps' share of them"""
    if not steps or SKIP_STEPS <= 0 or steps <= SKIP_STEPS or len(rows) % steps:
        return rows, steps
    per = len(rows) // steps
    return rows[SKIP_STEPS * per:], steps - SKIP_STEPS


def trace_stats(d):
    """kernel -> (calls, mean us, min us, max us, us per step) from a *_kernel_trace.csv.  A step of the
    production pipeline launches k_pyrblur and the FAST kernel TWICE (top rows first): `us per step` sums them."""
    raw = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            raw[kname(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    steps_all = len(raw.get(STEP_KERNEL, [])) or None
    out, steps = {}, steps_all
    for k, v in raw.items():
        rows, st = drop_warmup([x[1] for x in sorted(v)], steps_all)
        out[k] = rows
        if k == STEP_KERNEL:
            steps = st
    if steps_all and steps == steps_all and SKIP_STEPS > 0 and steps_all > SKIP_STEPS:
        steps = steps_all - SKIP_STEPS
    def per_step(k, v):
        # (a kernel whose dispatches could not be split by step keeps all of them: divide by all steps)
        return sum(v) / (steps if len(v) != len(raw[k]) or not SKIP_STEPS else steps_all) if steps_all else sum(v) / len(v)
    return {k: (len(v), sum(v) / len(v), min(v), max(v), per_step(k, v)) for k, v in out.items() if v}


def counters(d, per_step=True):
    """kernel -> counter -> mean per STEP (sum over the kernel's dispatches / steps of the run); per dispatch
    for runs without the step kernel (the calibration probes)"""
    raw = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            raw[kname(r["Kernel_Name"])][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    steps_all = 0
    if per_step and STEP_KERNEL in raw:
        steps_all = max(len(x) for x in raw[STEP_KERNEL].values())
    out = {}
    for k, v in raw.items():
        out[k] = {}
        for c, x in v.items():
            vals = [y[1] for y in sorted(x)]
            rows, st = drop_warmup(vals, steps_all) if per_step else (vals, 0)
            out[k][c] = sum(rows) / st if st else sum(rows) / len(rows)
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    import bench

    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    global SKIP_STEPS
    SKIP_STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    res = {"source_sha": bench.kernel_source_sha(), "batch": batch,
           "workload": "bench.py default: KITTI 1241x376, 8 levels, %d frames per launch" % batch,
           "kernels": {}, "configs": {}}
    for cfg in ("timed", "fullwork", "unfused", "fast3", "fast3timed", "b64_timed", "b64_fullwork", "b64_unfused", "b256_timed", "b256_fullwork",
                "b256_unfused", "hd_timed", "hd_fullwork"):
        st = trace_stats(os.path.join(src, cfg + "_stats"))
        if not st:
            continue
        with open(os.path.join(dst, cfg + "_kernel_stats.csv"), "w") as f:
            f.write("kernel,calls,avg_us,min_us,max_us,us_per_step\n")
            for k, (n, a, lo, hi, ps) in sorted(st.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
                f.write('"%s",%d,%.2f,%.2f,%.2f,%.2f\n' % (k, n, a, lo, hi, ps))
        cfgd = {"avg_us": {k: v[1] for k, v in st.items()}, "us_per_step": {k: v[4] for k, v in st.items()}}
        for part in ("sqa", "sqb", "fetch", "write"):
            c = counters(os.path.join(src, "%s_%s" % (cfg, part)))
            for k, v in c.items():
                cfgd.setdefault("counters", {}).setdefault(k, {}).update(v)
        res["configs"][cfg] = cfgd
    # calibration: FETCH_SIZE / WRITE_SIZE (KB) against the bytes bw_probe really moves
    calib = {}
    for mb in (97, 1600):
        f = counters(os.path.join(src, "calib_fetch_%d" % mb), per_step=False)
        w = counters(os.path.join(src, "calib_write_%d" % mb), per_step=False)
        if f or w:
            calib[str(mb)] = {"fetch_kb": {k: v.get("FETCH_SIZE") for k, v in f.items()},
                              "write_kb": {k: v.get("WRITE_SIZE") for k, v in w.items()}}
    res["calibration"] = calib
    # what bench.py reads: per kernel VALU instructions and corrected HBM traffic per launch
    def full_name(cfg, prefix):  # "k_fast3" -> "k_fast3<1>" as rocprofv3 prints the instantiation
        names = [k for k in res["configs"].get(cfg, {}).get("us_per_step", {}) if k == prefix or k.startswith(prefix + "<")]
        return max(names, key=lambda k: res["configs"][cfg]["us_per_step"][k]) if names else prefix

    def pick(cfg, kern, key):
        return res["configs"].get(cfg, {}).get("counters", {}).get(full_name(cfg, kern), {}).get(key)

    table = {"k_pyrblur": ("timed", "k_pyrblur"), "k_pyrblur_every_row": ("fullwork", "k_pyrblur"),
             "k_fast4": ("timed", "k_fast4"), "k_fast4_full_work": ("fullwork", "k_fast4"),
             "k_fast3_full_work": ("fast3", "k_fast3"), "k_level_select": ("timed", "k_level_select"),
             "k_describe2": ("timed", "k_describe2"), "k_pyramid2": ("unfused", "k_pyramid2"),
             "k_blur3": ("unfused", "k_blur3")}
    for name, (cfg, kern) in table.items():
        valu, fe, wr = pick(cfg, kern, "SQ_INSTS_VALU"), pick(cfg, kern, "FETCH_SIZE"), pick(cfg, kern, "WRITE_SIZE")
        e = {"valu": valu, "fetch_kb": fe, "write_kb": wr, "config": cfg,
             "us_per_step_rocprof": res["configs"].get(cfg, {}).get("us_per_step", {}).get(full_name(cfg, kern))}
        for extra in ("SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT",
                      "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            v = pick(cfg, kern, extra)
            if v is not None:
                e[extra.lower()] = v
        # gfx950: FETCH_SIZE reports exactly half of the bytes of a coalesced streaming read, at 4, 8 and
        # 16 B per lane alike, Infinity-Cache hits included (calibration below; MI355X_MICROARCH.md §HBM);
        # WRITE_SIZE reads the written bytes exactly
        if fe is not None and wr is not None:
            e["traffic_bytes"] = (2.0 * fe + wr) * 1024.0
        res["kernels"][name] = e
    json.dump(res, open(os.path.join(dst, "pmc_counters.json"), "w"), indent=1, sort_keys=True)
    for name in ("bench_default", "bench_batch64", "bench_fast3", "bench_1080p", "bw_probe_97", "bw_probe_1600"):
        for ext in (".json", ".txt"):
            f = os.path.join(src, name + ext)
            if os.path.exists(f) and os.path.getsize(f) > 0:
                open(os.path.join(dst, name + ext), "w").write(open(f).read())
    for name, e in res["kernels"].items():
        print(name, e)
    print("calibration", json.dumps(calib)[:2000])


main()
