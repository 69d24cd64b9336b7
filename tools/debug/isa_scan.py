"""Per kernel of a .hip file: how many MFMAs sit right behind an `s_waitcnt lgkmcnt(0|1)` (an LDS operand read just before its use) or a
`vmcnt(0|1)`, out of how many — the compiler's just-in-time operand reads that tapring.h / FragStream replace.
usage: python tools/debug/isa_scan.py tts_king_amd/csrc/file.hip ...        (tests/test_isa_cpu.py holds the hot kernels to their counts)"""
import os, re, subprocess, sys, tempfile

HIPCC = "/opt/rocm/bin/hipcc"


def scan(src):
    """{mangled kernel name: (MFMAs, behind lgkmcnt(0|1), behind vmcnt(0|1))} for every kernel of `src` with at least one MFMA."""
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", os.path.basename(src), "-o", f.name],
                       capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(src)), check=True)
        lines = open(f.name).read().splitlines()
    name, rows, prev = None, {}, []
    for line in lines:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1); rows[name] = [0, 0, 0]; prev = []; continue
        if name is None:
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if t.startswith("v_mfma"):
            rows[name][0] += 1
            if any(re.match(r"s_waitcnt.*lgkmcnt\([01]\)", p) for p in prev[-2:]): rows[name][1] += 1
            if any(re.match(r"s_waitcnt vmcnt\([01]\)", p) for p in prev[-2:]): rows[name][2] += 1
        prev.append(t)
    return {k: tuple(v) for k, v in rows.items() if v[0]}


if __name__ == "__main__":
    for src in sys.argv[1:]:
        for k, (n, a, b) in scan(src).items():
            print("%-28s %-80s mfma %4d  behind lgkmcnt(0|1) %4d  behind vmcnt(0|1) %3d" % (os.path.basename(src), k[:80], n, a, b))
