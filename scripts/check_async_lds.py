"""Static check of the asynchronous LDS reads in csrc/pair_block.hip (run by __graft_entry__.build() and by hand).

The kernel issues its weight-fragment and table reads through inline asm and waits with counted `s_waitcnt lgkmcnt(N)`; the compiler
believes an asm output is valid at once, so a register copy / AGPR spill it inserts between the read and the covering wait would move
stale data.  This script replays the in-order LDS return queue over the generated ISA of every pair_kernel instantiation (linear scan;
the loops of the kernel re-enter with the same queue shape) and reports any instruction that reads or writes a VGPR which is still
the destination of an LDS read in flight.  Exit status 1 on a finding.

usage: python scripts/check_async_lds.py [file.s]     (without an argument: compiles pair_block.hip to assembly itself)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def vregs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check(path):
    findings, fn, pending, nreads = [], None, [], 0
    for m in re.finditer(r"\.private_segment_fixed_size:\s*(\d+)", open(path).read()):
        if int(m.group(1)) != 0:
            findings.append(f"{path}: a kernel uses {m.group(1)} bytes of scratch per lane (register spills: VMEM traffic the counted waits do not cover)")
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":") and not line.startswith("."):
            fn = line[:-1] if "pair_kernel" in line else None
            pending = []
            continue
        if fn is None:
            continue
        op = line.split()[0]
        ops = line[len(op):]
        if op.startswith("s_waitcnt"):
            m = re.search(r"lgkmcnt\((\d+)\)", line)
            if m:
                n = int(m.group(1))
                while len(pending) > n:
                    pending.pop(0)
            continue
        touched = vregs(ops)
        for dest in pending:
            if dest & touched:
                findings.append(f"{path}:{ln}: {fn}: `{line}` touches v{sorted(dest & touched)} while an LDS read into it is in flight")
        if op.startswith("ds_read"):
            first = ops.split(",")[0]
            pending.append(vregs(first))
            nreads += 1
        elif op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            pending.append(set())          # counted by lgkmcnt, no vector destination
        elif op in ("s_endpgm",):
            fn = None
    return findings, nreads


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = os.path.join(tempfile.mkdtemp(), "pair_block.s")
        src = os.path.join(ROOT, "prediff_amd", "csrc", "pair_block.hip")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", path],
                       check=True, stderr=subprocess.DEVNULL)
    findings, nreads = check(path)
    for f in findings[:40]:
        print(f)
    print(f"check_async_lds: {nreads} LDS reads replayed, {len(findings)} finding(s)")
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
