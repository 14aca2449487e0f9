#!/usr/bin/env python3
"""Registers, spills and scratch of the kernels in the built library.

    python tools/kernel_resources.py [substring ...]

Pulls the gfx950 code object out of libdeconv3d_hip.so's offload bundle and prints
the AMDGPU metadata notes of every kernel whose (demangled) name holds one of the
substrings (default: all).  Runs without a GPU.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "deconv3d_amd", "csrc", "libdeconv3d_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def device_code_objects(path):
    """Every gfx950 code object of the library: one offload bundle per translation unit."""
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, at = [], data.find(magic)
    while at >= 0:
        (n,) = struct.unpack_from("<Q", data, at + len(magic))
        off = at + len(magic) + 8
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl]
            off += tl
            if b"gfx950" in triple:
                out.append(data[at + o:at + o + size])
        at = data.find(magic, at + len(magic))
    if not out:
        raise SystemExit("no gfx950 code object in %s" % path)
    return out


def main(argv):
    wanted = argv[1:]
    notes = ""
    for blob in device_code_objects(os.environ.get("DECONV3D_HIP_LIB", LIB)):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            notes += subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name],
                                    capture_output=True, text=True, check=True).stdout
    rows = []
    for block in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        get = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, block).group(1))
        rows.append((name, get("vgpr_count"), get("sgpr_count"), get("vgpr_spill_count"),
                     get("private_segment_fixed_size"), get("group_segment_fixed_size")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows),
                           capture_output=True, text=True, check=True).stdout.splitlines()
    print("%-6s %-6s %-6s %-8s %-8s  kernel" % ("vgpr", "sgpr", "spill", "scratch", "lds"))
    for (_, v, s, sp, sc, lds), name in zip(rows, names):
        name = name.replace("void d3d::", "").split("(")[0]
        if wanted and not any(w in name for w in wanted):
            continue
        print("%-6d %-6d %-6d %-8d %-8d  %s" % (v, s, sp, sc, lds, name))


if __name__ == "__main__":
    main(sys.argv)
