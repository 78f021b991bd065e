#!/usr/bin/env python3
"""Register / scratch use of every gfx950 kernel in the built library, read from the code objects inside libaicam.so.

    python tools/kernel_resources.py [path/to/libaicam.so] [--scratch-only]

The library carries one clang offload bundle per .hip source in its `.hip_fatbin` section; each bundle holds the gfx950 ELF whose
AMDGPU metadata note lists, per kernel, `.vgpr_count`, `.agpr_count` and `.private_segment_fixed_size` (scratch bytes per lane).
A kernel with scratch is a kernel that spills: tests/test_host_logic.py keeps the conv kernels at zero (the fp32
patch kernels once went to 11 KB per lane and 8x slower without any test noticing).  No GPU needed.
"""
from __future__ import annotations

import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib: Path, arch: str = "gfx950"):
    """-> the gfx950 ELF images of every bundle in the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fat = Path(td) / "fat.bin"
        subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)], check=True)
        blob = fat.read_bytes()
    out = []
    for m in re.finditer(MAGIC, blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, pos)
            triple = blob[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if arch in triple and size:
                out.append(blob[base + off:base + off + size])
    return out


def kernel_table(lib: Path):
    """-> {kernel name: dict(vgpr, agpr, sgpr, scratch, lds, vgpr_spills)} over all code objects."""
    import yaml
    table = {}
    for img in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", f.name], check=True, capture_output=True, text=True).stdout
        for doc in re.findall(r"^\s*---\n(.*?)^\s*\.\.\.", txt, re.S | re.M):
            meta = yaml.safe_load(doc)
            for k in (meta or {}).get("amdhsa.kernels", []):
                table[k[".name"]] = dict(vgpr=k.get(".vgpr_count", 0), agpr=k.get(".agpr_count", 0), sgpr=k.get(".sgpr_count", 0),
                                         scratch=k.get(".private_segment_fixed_size", 0), lds=k.get(".group_segment_fixed_size", 0),
                                         vgpr_spills=k.get(".vgpr_spill_count", 0))
    return table


def demangle(names):
    """Readable names through binutils' c++filt when it is on PATH (llvm-cxxfilt is not in this image); else the symbols as they are."""
    import shutil
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if tool is None:
        return {n: n for n in names}
    p = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, p.stdout.splitlines()))


def main(argv):
    args = [a for a in argv if not a.startswith("--")]
    lib = Path(args[0]) if args else Path(__file__).resolve().parent.parent / "ai-camera_amd" / "libaicam.so"
    tab = kernel_table(lib)
    names = demangle(sorted(tab))
    for k in sorted(tab, key=lambda k: (-tab[k]["scratch"], k)):
        r = tab[k]
        if "--scratch-only" in argv and not r["scratch"]:
            continue
        print(f"{r['vgpr']:4d} v {r['agpr']:4d} a {r['scratch']:6d} B scratch {r['lds']:7d} B lds  {names[k][:150]}")
    print(f"{len(tab)} kernels, {sum(1 for r in tab.values() if r['scratch'])} with scratch")


if __name__ == "__main__":
    main(sys.argv[1:])
