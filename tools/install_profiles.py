"""Copy one tools/round_profiles.sh output directory into profiles/<tag>_* and stamp the JSON summaries with the source fingerprint they
were measured on and the git head.  usage: python tools/install_profiles.py gpurun_out/round_r04c r04"""
import glob, json, os, shutil, subprocess, sys
sys.path.insert(0, os.getcwd())
from tts_king_amd import lib as L
src, tag = sys.argv[1], sys.argv[2]
fp = L.source_fingerprint()
head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
def one(pattern):
    m = sorted(glob.glob(os.path.join(src, pattern)))
    if not m: raise SystemExit("missing " + pattern)
    return m[0]
copies = {
    "bench.json": "%s_bench.json", "bench_line.json": "%s_bench_line.json", "fs2_trace/*kernel_stats.csv": "%s_bench_kernel_stats.csv", "hifi_trace/*kernel_stats.csv": "%s_hifigan_kernel_stats.csv",
    "mfma_util.json": "%s_mfma_util.json", "pmc_traffic.json": "%s_pmc_traffic.json", "pmc_traffic_hifi.json": "%s_pmc_traffic_hifi.json",
    "step_timeline.txt": "%s_step_timeline.txt", "step_timeline_real.txt": "%s_step_timeline_real.txt", "chain_cost.json": "%s_chain_cost.json",
}
for pat, dst in copies.items():
    d = os.path.join("profiles", dst % tag)
    shutil.copyfile(one(pat), d)
    if d.endswith(".json") and "_bench.json" not in d and "_bench_line.json" not in d:
        j = json.load(open(d))
        if j.get("csrc_fingerprint") != fp:
            raise SystemExit("%s was measured on source %s, the tree is %s" % (d, j.get("csrc_fingerprint"), fp))
        j["git_head"] = head
        json.dump(j, open(d, "w"), indent=1)
    print("installed", d)
# the kernel-stats CSV carries no fingerprint of its own: a sidecar says which sources it was measured on (the PMC summary of the same
# round_profiles.sh run has been checked against the tree above) and how many steps the trace holds (bench.py: profile_family_time)
import csv
stats = os.path.join("profiles", "%s_bench_kernel_stats.csv" % tag)
steps = sum(int(r["Calls"]) for r in csv.DictReader(open(stats)) if "adam_pack_kernel" in r["Name"] or "adam_clip_kernel" in r["Name"])
json.dump({"csrc_fingerprint": fp, "git_head": head, "steps": steps, "command": "rocprofv3 --kernel-trace --stats -- /usr/bin/python3 bench.py "
           "--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-mel --no-e2e --no-hifi --no-extra (tools/round_profiles.sh)"},
          open(os.path.join("profiles", "%s_bench_kernel_stats.meta.json" % tag), "w"), indent=1)
print("installed profiles/%s_bench_kernel_stats.meta.json (%d steps)" % (tag, steps))
