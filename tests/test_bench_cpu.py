"""bench.py's ONE line: it must fit the driver's 8 KB stdout tail with both halves of the metric in it, and
`roofline.frac` must follow from the committed rocprofv3 summary under profiles/."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _full_records():
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[5-9]*_bench.json")))


def _worst_case(rec):
    """The same record with everything a multi-GPU / profiled run adds: six bucket sizes, long source strings, unknown switches."""
    r = json.loads(json.dumps(rec))
    r["config"].update({"collective_world_size": 8, "collective_backend": "nccl", "allreduce_bucket_bytes": [25165824] * 6,
                        "allreduces_per_step": 6, "collective_timeout_s": 300.0, "parallelism": "dp8"})
    r["unknown_switches"] = ["TTSK_SOMETHING_NOBODY_READS_%d" % i for i in range(6)]
    if r.get("roofline"):
        r["roofline"]["frac_source"] = "profiles/r06_bench_kernel_stats.csv: 11 symbols, 34.0 launches and 0.765 ms per step over 15 traced steps"
    return r


@pytest.mark.parametrize("path", _full_records())
def test_line_fits_the_driver_tail_with_both_halves(path):
    rec = json.load(open(path))
    for r in (rec, _worst_case(rec)):
        line = bench.compact_record(r, "gpurun_out/bench_full.json")
        assert len(line) <= bench.LINE_LIMIT <= 6144, len(line)
        assert "\n" not in line
        out = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in out, k
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in out["roofline"], k
        h = out["hifi_gan"]                      # the metric's second half, inside the last 8 KB whatever follows it
        assert h["rtf"] > 0 and h["ms_per_batch"] > 0 and "mfma_roofline_frac" in h and "cpu_baseline" in h
        assert line.rfind('"hifi_gan"') > len(line) - 8192
        assert abs(out["value"] - r["value"]) <= 1e-5 * r["value"] and abs(out["ms_per_step"] - r["ms_per_step"]) <= 1e-5 * r["ms_per_step"]


def test_line_refuses_to_grow_silently():
    rec = json.load(open(_full_records()[-1]))
    rec["config"]["workload"] = "x" * 7000
    with pytest.raises(RuntimeError):
        bench.compact_record(rec)


def test_roofline_frac_follows_from_the_committed_rocprof_summary():
    """family GF per step / (sum of the family's TotalDurationNs / traced steps) from profiles/r*_bench_kernel_stats.csv, recomputed here by hand."""
    import csv
    prof = bench.profile_family_time("win_conv_kernel")
    assert prof is not None
    rows = list(csv.DictReader(open(os.path.join(ROOT, prof["source"]))))
    steps = sum(int(r["Calls"]) for r in rows if "adam_pack_kernel" in r["Name"] or "adam_clip_kernel" in r["Name"])
    ns = sum(float(r["TotalDurationNs"]) for r in rows if "win_conv_kernel" in r["Name"])
    assert steps == prof["steps_traced"] and abs(prof["ms_per_step"] - ns / steps * 1e-6) < 1e-9
    tag = os.path.basename(prof["source"]).split("_bench_kernel_stats")[0]
    full = os.path.join(ROOT, "profiles", tag + "_bench.json")
    if os.path.exists(full):
        r = json.load(open(full))["roofline"]
        if r.get("family") == "win_conv" and "frac_source" in r and r["frac_source"].startswith("profiles/"):
            gf = [f for f in r["families"] if f["family"] == "win_conv"][0]["gflop"]
            assert abs(r["frac"] - gf / prof["ms_per_step"] / 2500.0) <= 0.03 * r["frac"]
