#!/usr/bin/env python3
"""Registers / scratch / LDS of the kernels of a built object or library, read from the gfx950 code object's metadata
(no GPU needed).  usage: tools/kernel_resources.py <file.o | lib.so> [substring filter ...]
The ISA vgpr count is what occupancy follows (allocation granule 8; rocprofv3's VGPR_Count column shows half of it)."""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(path, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
    co = os.path.join(tmp, "k.co")
    subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--input=" + fat, "--output=" + co, "--unbundle"], check=True, stderr=subprocess.DEVNULL)
    return co


def kernels(path):
    with tempfile.TemporaryDirectory() as tmp:
        co = code_object(path, tmp)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == "agpr_count" and cur.get("name"):
            out.append(cur)
            cur = {}
        if k in ("name", "vgpr_count", "sgpr_count", "agpr_count", "private_segment_fixed_size", "group_segment_fixed_size",
                 "max_flat_workgroup_size"):
            if k == "name" and "name" in cur and "vgpr_count" in cur:
                out.append(cur)
                cur = {}
            if k == "name" and v.startswith("_Z") is False and "name" in cur:
                continue
            cur[k] = v
    if cur.get("name"):
        out.append(cur)
    return out


def main():
    path, filt = sys.argv[1], sys.argv[2:]
    seen = set()
    for k in kernels(path):
        if "vgpr_count" not in k:
            continue
        name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
        if name in seen or (filt and not all(f in name for f in filt)):
            continue
        seen.add(name)
        print("%-100s vgpr %3s agpr %2s sgpr %3s scratch %4s B/lane  static LDS %6s" % (
            name[:100], k.get("vgpr_count"), k.get("agpr_count", "0"), k.get("sgpr_count"),
            k.get("private_segment_fixed_size"), k.get("group_segment_fixed_size")))


if __name__ == "__main__":
    main()
