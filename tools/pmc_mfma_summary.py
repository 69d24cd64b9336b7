"""Per-kernel MFMA utilisation from the rocprofv3 --pmc pass of tools/pmc_mfma.sh.

usage: python tools/pmc_mfma_summary.py <pmc_mfma dir> <out.json>

ROCm 7.2 ships no gfx950 section in its derived-counter files (MI355X_MICROARCH.md, rocprofv3 PMC slots), so the ratios are
formed here from raw counters, per dispatch, then averaged per kernel symbol:
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
      SQ_VALU_MFMA_BUSY_CYCLES = cycles a SIMD's matrix pipe is busy, summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE = busy
      cycles summed over the 8 XCDs (so / 8 = the dispatch's length in shader cycles).  1.0 = every matrix pipe busy all the time.
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra LDS cycles spent on bank conflicts)
  wave_stall_frac   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (share of wave lifetime stalled at issue), active = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
"""
import csv, glob, json, os, sys


def load(d):
    disp = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (f, r.get("Dispatch_Id"))
            e = disp.setdefault(k, {"kernel": (r.get("Kernel_Name") or r.get("Kernel") or "?").replace("(anonymous namespace)::", "")})
            e[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(disp.values())


def summarise(rows):
    agg = {}
    for e in rows:
        gui = e.get("GRBM_GUI_ACTIVE", 0.0)
        if gui <= 0:
            continue
        a = agg.setdefault(e["kernel"], {"n": 0, "mfma": 0.0, "cycles": 0.0, "conf": 0.0, "lds": 0.0, "stall": 0.0, "act": 0.0, "wave": 0.0})
        a["n"] += 1
        a["mfma"] += e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a["cycles"] += gui / 8.0
        a["conf"] += e.get("SQ_LDS_BANK_CONFLICT", 0.0); a["lds"] += e.get("SQ_LDS_IDX_ACTIVE", 0.0)
        a["stall"] += e.get("SQ_WAIT_INST_ANY", 0.0); a["act"] += e.get("SQ_ACTIVE_INST_ANY", 0.0); a["wave"] += e.get("SQ_WAVE_CYCLES", 0.0)
    out = []
    for k, a in agg.items():
        out.append({"kernel": k, "dispatches": a["n"], "avg_cycles": a["cycles"] / a["n"],
                    "mfma_busy_frac": a["mfma"] / (a["cycles"] * 1024.0) if a["cycles"] else None,
                    "lds_conflict_frac": a["conf"] / a["lds"] if a["lds"] else 0.0,
                    "wave_stall_frac": a["stall"] / a["wave"] if a["wave"] else None,
                    "wave_active_frac": a["act"] / a["wave"] if a["wave"] else None,
                    "share_of_cycles": a["cycles"]})
    tot = sum(o["share_of_cycles"] for o in out) or 1.0
    for o in out:
        o["share_of_cycles"] = o["share_of_cycles"] / tot
    out.sort(key=lambda o: -o["share_of_cycles"])
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tts_king_amd.lib import source_fingerprint
    doc = {"csrc_fingerprint": source_fingerprint(),
           "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY "
                     "SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace (tools/pmc_mfma.sh); ratios: see tools/pmc_mfma_summary.py",
           "fs2_train_step": summarise(load(os.path.join(src, "fs2"))), "hifi_gan": summarise(load(os.path.join(src, "hifi")))}
    json.dump(doc, open(dst, "w"), indent=1)
    for name in ("fs2_train_step", "hifi_gan"):
        print(name)
        for o in doc[name][:16]:
            if o["mfma_busy_frac"] is None:
                continue
            print("  %-64s n=%4d share %5.1f%%  mfma_busy %5.1f%%  lds_conflict %4.1f%%  stall %4.1f%%" % (
                o["kernel"][:64], o["dispatches"], 100 * o["share_of_cycles"], 100 * o["mfma_busy_frac"], 100 * o["lds_conflict_frac"],
                100 * (o["wave_stall_frac"] or 0)))


if __name__ == "__main__":
    main()
