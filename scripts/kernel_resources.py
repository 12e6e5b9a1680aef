#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel instantiation in the library (compiles the device code to assembly;
no GPU needed).  usage: python scripts/kernel_resources.py [filter-substring] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mcbrat3d_amd", "csrc")


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    flt = argv[0] if argv else ""
    keep = os.environ.get("KRES_DIR")
    tmp = keep or tempfile.mkdtemp(prefix="kres")
    os.makedirs(tmp, exist_ok=True)
    asm = os.path.join(tmp, "mcbrat_api.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
           "-o", asm, os.path.join(CSRC, "mcbrat_api.hip")] + extra
    subprocess.check_call(cmd)
    text = open(asm).read()
    # the metadata block at the end lists every kernel
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
        blk = m.group(0)
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt and flt not in dem:
            continue
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        short = re.sub(r"mcbrat::|\(mcbrat::DevParams\)|\(mcbrat::FinishParams\)", "", dem)
        print("%-70s vgpr %3d sgpr %3d scratch %4d lds %6d spill_sgpr %3d spill_vgpr %3d" % (
            short[:70], g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"),
            g("sgpr_spill_count"), g("vgpr_spill_count")))
    print("assembly:", asm)


if __name__ == "__main__":
    main()
