"""Safety check of attention4w.hip's asm-owned AGPR block.  The kernel keeps its O^T accumulators, Q and V^T fragments in AGPRs
that are NAMED in inline asm: the register allocator knows them only as clobbers, so a compiler-generated v_accvgpr_* (an AGPR
used as VGPR spill space) or any scratch access inside attn4w_kernel would silently corrupt them.  This compiles the file to
assembly with the library's flags and fails when either appears.      python tools/check_attn4w_isa.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "instancediffusion_amd", "csrc", "attention4w.hip")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form".split()


def check(extra=(), src=SRC, kernel="attn4w_kernel"):
    """src / kernel: the same scan for another file with asm-owned AGPRs (mlp_fused.hip's mlp320w_kernel)"""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "a.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "--cuda-device-only", "-S", src, "-o", out], check=True,
                       cwd=os.path.dirname(src))
        txt = open(out).read()
    report = {}
    for name, body in re.findall(r"\n(_ZN\S*" + kernel + r"\S*):[^\n]*\n(.*?)\n\ts_endpgm", txt, flags=re.S):
        in_asm, stray, scratch, mfma = False, [], 0, 0
        for line in body.split("\n"):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif t.startswith("v_accvgpr") and not in_asm:
                stray.append(t)
            elif t.startswith("scratch_"):
                scratch += 1
            elif t.startswith("v_mfma"):
                mfma += 1
        report[name] = dict(stray_accvgpr=stray, scratch_ops=scratch, mfma=mfma)
    return report


if __name__ == "__main__":
    rep = check(sys.argv[1:])
    bad = False
    for k, v in rep.items():
        print(k, "stray v_accvgpr:", len(v["stray_accvgpr"]), "scratch ops:", v["scratch_ops"], "mfma:", v["mfma"])
        bad |= bool(v["stray_accvgpr"]) or v["scratch_ops"] > 0
    if not rep:
        print("no attn4w kernel found")
        bad = True
    sys.exit(1 if bad else 0)
