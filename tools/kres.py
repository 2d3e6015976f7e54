#!/usr/bin/env python3
"""Register / scratch / code-size table of every kernel in kernels.hip (hipcc cross-compiles; no GPU needed).
  python tools/kres.py [extra hipcc flags ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from poseidon252_amd import build as b
    b._gen_assets()
    out = os.path.join(tempfile.mkdtemp(prefix="p252_kres_"), "kernels.s")
    cmd = [b._hipcc()] + [f for f in b.HIPCC_FLAGS if f != "-fPIC"] + sys.argv[1:] + ["-S", "--cuda-device-only", "-o", out, os.path.join(b.CSRC, "kernels.hip")]
    if os.environ.get("KRES_ASM"):
        out = os.environ["KRES_ASM"]
    else:
        subprocess.check_call(cmd, cwd=b.CSRC, stderr=subprocess.DEVNULL)
    text = open(out).read()
    print("%-46s %5s %5s %5s %7s %6s %5s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "lds", "occ", "code B"))
    pat = (r"^(_Z\w+):.*?; codeLenInByte = (\d+).*?; TotalNumSgprs: (\d+).*?; NumVgprs: (\d+).*?; NumAgprs: (\d+).*?"
           r"; ScratchSize: (\d+).*?; LDSByteSize: (\d+).*?; Occupancy: (\d+)")
    for m in re.finditer(pat, text, re.S | re.M):
        name, code, sg, vg, ag, sc, lds, occ = m.groups()
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0].replace("p252::", "").replace("void ", "")
        print("%-46s %5s %5s %5s %7s %6s %5s %8s" % (short, vg, ag, sg, sc, lds, occ, code))
    print("asm:", out)


if __name__ == "__main__":
    main()
