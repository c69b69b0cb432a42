#!/usr/bin/env python3
"""Resource table of every kernel in liborbx.so (VGPRs, SGPRs, static LDS, scratch, workgroup size) read from the code
object's notes -- what decides which kernels can be resident on a CU together.  No GPU needed.
usage: python tools/kernel_resources.py [path/to/liborbx.so]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(so):
    tmp = "/tmp/orbx_co"
    os.makedirs(tmp, exist_ok=True)
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + so,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + tmp + "/co.o"], check=False,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = []
    # a shared library holds one bundle per object: roc-obj-ls lists them, but llvm-readelf on the extracted
    # fat section is simpler -- take every code object out of .hip_fatbin
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, tmp + "/fat.bin"],
                         check=True)
    data = open(tmp + "/fat.bin", "rb").read()
    # code objects are ELF images inside the bundle
    idx = [m.start() for m in re.finditer(b"\x7fELF\x02\x01\x01\x40", data)]
    for n, a in enumerate(idx):
        b = idx[n + 1] if n + 1 < len(idx) else len(data)
        p = "%s/co%d.o" % (tmp, n)
        open(p, "wb").write(data[a:b])
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", p], capture_output=True, text=True).stdout
        for m in re.finditer(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)", txt, re.S):
            blk = m.group(0)
            g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]  # noqa: E731
            rows.append((g("name"), g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"),
                         g("max_flat_workgroup_size")))
    return rows


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "monoorbslam3_amd", "lib", "liborbx.so")
    print("%-90s %5s %5s %8s %7s %6s" % ("kernel", "vgpr", "sgpr", "lds", "scratch", "wg"))
    for r in sorted(table(so)):
        name = subprocess.run(["c++filt", r[0]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)[:90]
        print("%-90s %5s %5s %8s %7s %6s" % (name, r[1], r[2], r[3], r[4], r[5]))
