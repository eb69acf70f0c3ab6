#!/usr/bin/env python
"""Census, in the BUILT library, of the packed-f32 instructions that are unsafe next to another kernel's MFMA waves on the same SIMD.

Round 6 (NOTEBOOK.md section 16.7, profiles/r06/p_pk_opsel_probe.md): on the MI355X boxes of this pool `v_pk_{mul,add,fma}_f32` with a
VGPR src1 read with op_sel[1] = 1 (the LOW result takes src1's HIGH register) returns wrong values in lanes 48-63 while a wave of another
kernel on the same SIMD runs MFMAs with AGPR accumulators (k_conv_bx, k_wgrad_bx*).  Alone on the SIMD, or with the select on src0 / src2 /
an SGPR source, the instruction is right.  The library must not contain the form: any of its kernels may share a CU with the convolutions
of another stream (weight gradients beside the backward pass; pool batches on two streams).

    python tools/isa_opsel_census.py [path/to/lib.so]       # default: mulactseg_amd/libmulactseg_hip.so; exit status 1 if any is found
    python tools/isa_opsel_census.py $(python -c "import torch, os; print(os.path.dirname(torch.__file__))")/lib/libtorch_hip.so   # ~5 min

tests/test_isa_cpu.py runs the same census on the library `__graft_entry__.build()` produced."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PK = re.compile(r"^\s+(v_pk_(?:mul|add|fma)_f32)\s+([^/;]*)")
OPSEL = re.compile(r"op_sel:\[([01,]+)\]")
LABEL = re.compile(r"^[0-9a-f]+ <([^>]+)>:")


def unsafe(line):
    """True for a disassembled packed-f32 instruction whose src1 is a VGPR pair with op_sel[1] set."""
    m = PK.match(line)
    if not m:
        return False
    rest = m.group(2)
    sel = OPSEL.search(rest)
    if not sel:
        return False
    bits = sel.group(1).split(",")
    if len(bits) < 2 or bits[1] != "1":
        return False
    ops = [o.strip() for o in rest.split(" op_sel")[0].split(",")]       # vdst, src0, src1 [, src2]
    return len(ops) >= 3 and ops[2].startswith("v")


def code_objects(path, arch="gfx950"):
    """The device code objects (bytes) of every offload bundle embedded in a host shared library."""
    data = open(path, "rb").read()
    out = []
    for m in re.finditer(MAGIC, data):
        base = m.start()
        n = struct.unpack_from("<Q", data, base + 24)[0]
        q = base + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            q += 24
            triple = data[q:q + tl].decode()
            q += tl
            if size and triple.endswith(arch):
                out.append(data[base + off:base + off + size])
    # compressed bundles ("CCOB", version 2: u16 version, u16 method, u32 total size, u32 uncompressed size, u64 hash, payload) -- how
    # PyTorch's own libraries ship their kernels; clang-offload-bundler unpacks them
    for m in re.finditer(b"CCOB", data):
        i = m.start()
        ver, _meth, total, _unc, _hash = struct.unpack_from("<HHIIQ", data, i + 4)
        if ver != 2 or total <= 24 or i + total > len(data):
            continue
        with tempfile.TemporaryDirectory() as tmp:
            blob, co = os.path.join(tmp, "b.bin"), os.path.join(tmp, "b.co")
            open(blob, "wb").write(data[i:i + total])
            r = subprocess.run([BUNDLER, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--" + arch, "--input=" + blob, "--output=" + co],
                               capture_output=True)
            if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co):
                out.append(open(co, "rb").read())
    return out


def census_library(path):
    """({kernel symbol: count of unsafe instructions}, kernels seen, packed-f32 instructions seen) over the library's device code."""
    found, kernels, packed = {}, 0, 0
    for blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            text = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in text.splitlines():
            lab = LABEL.match(line)
            if lab:
                cur = lab.group(1)
                kernels += 1
            elif PK.match(line):
                packed += 1
                if unsafe(line):
                    found[cur] = found.get(cur, 0) + 1
    return found, kernels, packed


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "mulactseg_amd", "libmulactseg_hip.so")
    found, kernels, packed = census_library(path)
    for sym, n in sorted(found.items()):
        print("%5d  %s" % (n, sym[:160]))
    print("%s: %d functions, %d packed-f32 instructions, %d in the unsafe form (%d functions)" % (os.path.relpath(path, ROOT), kernels, packed,
                                                                                                sum(found.values()), len(found)))
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
