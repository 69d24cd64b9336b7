"""Per-kernel HBM traffic from the two rocprofv3 --pmc passes of tools/pmc_bench.sh.

usage: python tools/pmc_summary.py <pmc_bench dir> <out.json>
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch.  Corrections of MI355X_MICROARCH.md (HBM section):
on gfx950 FETCH_SIZE tallies each 128-byte request of a wide coalesced read as 64 bytes, so it is doubled; WRITE_SIZE is
exact for 16-byte-per-lane stores.  Infinity-Cache hits are counted by both (they are requests on the fabric side of L2)."""
import csv, glob, json, os, sys


def _fingerprint():
    """Identity of the kernel sources the counters were taken with (tts_king_amd/lib.py:source_fingerprint); bench.py compares it."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tts_king_amd.lib import source_fingerprint
    return source_fingerprint()


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r.get("Kernel_Name") or r.get("Kernel") or "?"
            v = out.setdefault(k, [0.0, 0])
            v[0] += float(r["Counter_Value"]); v[1] += 1
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    fetch, write = per_kernel(os.path.join(src, "fetch"), "FETCH_SIZE"), per_kernel(os.path.join(src, "write"), "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, [0.0, 0]), write.get(k, [0.0, 0])
        n = max(f[1], w[1])
        rows.append({"kernel": k.replace("(anonymous namespace)::", ""), "dispatches": n,
                     "fetch_bytes_per_launch": 2.0 * 1024.0 * f[0] / max(f[1], 1),     # x2: gfx950 correction
                     "write_bytes_per_launch": 1024.0 * w[0] / max(w[1], 1)})
    for r in rows:
        r["hbm_bytes_per_launch"] = r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"]
    rows.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["dispatches"])
    json.dump({"csrc_fingerprint": _fingerprint(),
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 3 --warmup 2",
               "unit": "bytes per launch (FETCH_SIZE x2 per the gfx950 rule, + WRITE_SIZE)", "kernels": rows}, open(dst, "w"), indent=1)
    for r in rows[:16]:
        print("%-70s n=%5d  fetch %8.2f MB  write %8.2f MB" % (r["kernel"][:70], r["dispatches"], r["fetch_bytes_per_launch"] / 1e6, r["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
