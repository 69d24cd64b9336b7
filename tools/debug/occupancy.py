"""Per kernel of a .hip file: LDS bytes, VGPRs, spills and the workgroups per CU each of them allows (512 VGPRs per SIMD lane, 160 KiB of LDS) —
an unintended register count above 256 / 170 / 128 silently halves the co-residency a kernel was designed for.
usage: python tools/debug/occupancy.py tts_king_amd/csrc/file.hip ..."""
import os, re, subprocess, sys, tempfile


def scan(src):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", os.path.basename(src), "-o", f.name],
                       capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(src)), check=True)
        t = open(f.name).read()
    rows = []
    for m in re.finditer(r"\.group_segment_fixed_size: (\d+).*?\.max_flat_workgroup_size: (\d+).*?\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count: (\d+)", t, re.S):
        lds, wg, name, vg, sp = m.groups()
        lds, wg, vg = int(lds), int(wg), int(vg)
        waves = wg // 64
        per_simd = min(8, 512 // max(vg, 1))
        rows.append((name, lds, wg, vg, int(sp), (per_simd * 4) // waves, 163840 // lds if lds else 99))
    return rows


if __name__ == "__main__":
    for src in sys.argv[1:]:
        for name, lds, wg, vg, sp, by_v, by_l in scan(src):
            short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:70]
            print("%-16s %-70s threads %4d lds %6d vgpr %3d spill %2d  workgroups/CU: %d by registers, %d by LDS" % (os.path.basename(src), short, wg, lds, vg, sp, by_v, by_l))
