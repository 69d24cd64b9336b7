"""Per kernel of a .hip file: how many MFMAs sit right behind an `s_waitcnt lgkmcnt(0|1)` (an LDS operand read just before its use) or a
`vmcnt(0|1)`, out of how many — the compiler's just-in-time operand reads that tapring.h / FragStream replace.  usage: python tools/debug/isa_scan.py csrc/file.hip ..."""
import re, subprocess, sys, os
for src in sys.argv[1:]:
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", src, "-o", "-"],
                         capture_output=True, text=True, cwd=os.path.dirname(src) or ".").stdout if False else None
    out = "/tmp/isa_scan.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", os.path.basename(src), "-o", out],
                   capture_output=True, text=True, cwd=os.path.dirname(src) or ".")
    name, rows = None, {}
    prev = []
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1); rows[name] = [0, 0, 0]; prev = []; continue
        if name is None: continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith("."): continue
        if t.startswith("v_mfma"):
            rows[name][0] += 1
            for p in prev[-2:]:
                if re.match(r"s_waitcnt.*lgkmcnt\([01]\)", p): rows[name][1] += 1; break
            for p in prev[-2:]:
                if re.match(r"s_waitcnt vmcnt\([01]\)", p): rows[name][2] += 1; break
        prev.append(t)
    for k, (n, a, b) in rows.items():
        if n: print("%-28s %-80s mfma %4d  behind lgkmcnt(0|1) %4d  behind vmcnt(0|1) %3d" % (os.path.basename(src), k[:80], n, a, b))
