#!/usr/bin/env python3
"""kernel descriptors of the gfx950 code objects inside a host object / shared library: registers, spills, scratch, LDS.

  tools/kernel_meta.py [file ...] [--grep SUBSTR] [--spills]

Reads the clang offload bundles (`__CLANG_OFFLOAD_BUNDLE__`) of each file (default: porla_amd/libmultiexp.so), writes every
gfx950 code object to a temporary file and prints the `.amdhsa` note fields llvm-readelf shows for each kernel:
name, .vgpr_count, .vgpr_spill_count, .sgpr_spill_count, .private_segment_fixed_size (scratch bytes per lane),
.group_segment_fixed_size (static LDS).  `--spills`: only kernels with spills or scratch."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    data = open(path, "rb").read()
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", data, pos + len(MAGIC))[0]
        off = pos + len(MAGIC) + 8
        for _ in range(n):
            o, size, tlen = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24:off + 24 + tlen].decode()
            off += 24 + tlen
            if "gfx950" in triple and size:
                yield data[pos + o:pos + o + size]
        pos += len(MAGIC)


def kernels(blob):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(blob)
        f.flush()
        txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    cur = {}
    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):     # first key of a kernel record in the note
            if "vgpr_count" in cur:
                yield cur
            cur = {}
        cur[k] = v.strip("'\"")
        if k == "wavefront_size" and "vgpr_count" in cur and "name" in cur:
            yield cur
            cur = {}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    grep = None
    if "--grep" in sys.argv:
        grep = sys.argv[sys.argv.index("--grep") + 1]
        args = [a for a in args if a != grep]
    only_spills = "--spills" in sys.argv
    files = args or [os.path.join(ROOT, "porla_amd", "libmultiexp.so")]
    rows = []
    for path in files:
        for blob in code_objects(path):
            for k in kernels(blob):
                rows.append(k)
    names = subprocess.run([CXXFILT], input="\n".join(r.get("name", "?") for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("%6s %6s %6s %8s %8s  %s" % ("vgpr", "vspill", "sspill", "scratch", "lds", "kernel"))
    for r, nm in zip(rows, names):
        if grep and grep not in nm:
            continue
        sp = int(r.get("vgpr_spill_count", 0)) + int(r.get("private_segment_fixed_size", 0))
        if only_spills and sp == 0:
            continue
        nm = re.sub(r"^void ", "", nm)
        print("%6s %6s %6s %8s %8s  %s" % (r.get("vgpr_count"), r.get("vgpr_spill_count"), r.get("sgpr_spill_count"),
                                            r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size"), nm[:150]))


if __name__ == "__main__":
    main()
