"""Join the two passes of tools/chain_cost.sh: per kernel instance (symbol + grid) the in-step and the alone figures, and the attribution of the
difference.  usage: python tools/chain_cost.py <dir with step/ and alone/> <out.json>

rocprofv3 serialises dispatches while it collects counters: in the two --pmc passes every kernel had the chip to itself (~100 us apart).  So three
states of the same kernel instance (symbol + grid) are compared, per launch:
  alone      20 launches of the kernel back to back (pass b): warm caches, no neighbours
  isolated   the kernel in the step's order and memory state, but alone on the chip (pass a): what the step's DATA costs it (first touch of operands
             the previous kernels left in another XCD's L2 or in HBM)
  in_graph   the kernel between its real neighbours in the replayed graph (pass c, trace only): + what the NEIGHBOURS cost it (a predecessor still
             draining on some CUs, a concurrent stream's kernel), and `gap_us` = chip idle in front of it (start - latest end of any earlier dispatch)
cycles = GRBM_GUI_ACTIVE / 8 and parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES come from the --pmc passes.  (GRBM_GUI_ACTIVE / 8 / duration is NOT the shader clock
on dispatches this short — it reads 3-5 "GHz" on 7-25 us kernels, MI355X_MICROARCH.md "DVFS give-back" — so no clock attribution is attempted.)"""
import csv, glob, json, os, re, sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*\)$", "", name)[:72]


def load(d, need_counters=True):
    trace = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Kind") == "KERNEL_DISPATCH":
                trace[r["Dispatch_Id"]] = r
    cnt = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            cnt.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    rows = []
    order = sorted(trace.values(), key=lambda r: int(r["Start_Timestamp"]))
    latest_end = None
    for r in order:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        c = cnt.get(r["Dispatch_Id"], {})
        gap = 0.0 if latest_end is None else max(0, s - latest_end) / 1e3
        latest_end = e if latest_end is None else max(latest_end, e)
        if need_counters and c.get("GRBM_GUI_ACTIVE", 0) <= 0:
            continue
        grid = "%dx%dx%d" % (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        us = (e - s) / 1e3
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        rows.append({"key": short(r["Kernel_Name"]) + " grid " + grid, "us": us, "cycles": cyc, "ghz": cyc / us / 1e3, "gap_us": gap,
                     "parked": c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0),
                     "issue_stall": c.get("SQ_WAIT_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0)})
    return rows


def mean(rows, k):
    return sum(r[k] for r in rows) / len(rows)


def summarise(rows, skip_first=0):
    by = {}
    for r in rows:
        by.setdefault(r["key"], []).append(r)
    out = {}
    for k, v in by.items():
        v = v[skip_first:] if len(v) > skip_first + 2 else v
        out[k] = {"launches": len(v), **{f: mean(v, f) for f in ("us", "cycles", "ghz", "gap_us", "parked", "issue_stall")}}
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    step = summarise(load(os.path.join(src, "step")))
    alone = summarise(load(os.path.join(src, "alone")), skip_first=2)      # (the first launches of a run of 20 are the cold ones)
    graph = summarise(load(os.path.join(src, "graph"), need_counters=False))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tts_king_amd.lib import source_fingerprint
    doc = {"csrc_fingerprint": source_fingerprint(),
           "source": "tools/chain_cost.sh: rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace over (a) bench.py "
                     "--steps 10 --warmup 3 (the replayed train step) and (b) tools/debug/chain_alone.py (the same kernels alone, 20 launches back to back); "
                     "fields and attribution: tools/chain_cost.py",
           "kernels": []}
    for k, a in sorted(alone.items()):
        s, g = step.get(k), graph.get(k)
        if s is None or g is None or "copyBuffer" in k:
            continue
        rec = {"kernel": k, "alone": a, "isolated_in_step_state": s, "in_graph": {f: g[f] for f in ("launches", "us", "gap_us")},
               "attribution_us": {"data_state (isolated - alone)": s["us"] - a["us"], "neighbours (in_graph - isolated)": g["us"] - s["us"],
                                  "idle_in_front (in_graph gap)": g["gap_us"]}}
        doc["kernels"].append(rec)
        print("%-44s alone %5.1f us (parked %2.0f%%) | isolated, step state %5.1f us (parked %2.0f%%) | in graph %5.1f us, idle in front %4.1f us  =>  data %+4.1f, neighbours %+4.1f" % (
            k[:44], a["us"], 100 * a["parked"], s["us"], 100 * s["parked"], g["us"], g["gap_us"], s["us"] - a["us"], g["us"] - s["us"]))
    # the whole step, for context: every kernel instance's in-step figures
    doc["step_kernels_isolated_vs_in_graph"] = [{"kernel": k, "launches": v["launches"], "isolated_us": v["us"], "in_graph_us": graph[k]["us"], "idle_in_front_us": graph[k]["gap_us"],
                                                 "parked": v["parked"], "issue_stall": v["issue_stall"]}
                                                for k, v in sorted(step.items(), key=lambda kv: -kv[1]["us"] * kv[1]["launches"]) if k in graph and "copyBuffer" not in k][:40]
    tot_i = sum(r["isolated_us"] * r["launches"] for r in doc["step_kernels_isolated_vs_in_graph"])
    tot_g = sum(r["in_graph_us"] * r["launches"] for r in doc["step_kernels_isolated_vs_in_graph"])
    steps = max([v["launches"] for k, v in step.items() if "adam_pack_kernel" in k] + [1])
    doc["step_totals_us"] = {"isolated": tot_i / steps, "in_graph": tot_g / steps, "traced_steps": steps}
    print("the step's 40 heaviest kernel instances, per step: %.0f us isolated, %.0f us between their neighbours (%d traced steps)" % (tot_i / steps, tot_g / steps, steps))
    json.dump(doc, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main()
