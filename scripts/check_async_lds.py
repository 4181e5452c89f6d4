"""Static check of the asynchronous LDS reads in csrc/pair_block.hip (run by __graft_entry__.build() and by hand).

The kernel issues its weight-fragment and table reads through inline asm and waits with counted `s_waitcnt lgkmcnt(N)`; the compiler
believes an asm output is valid at once, so a register copy / AGPR spill it inserts between the read and the covering wait would move
stale data.  This script replays the in-order LDS return queue over the generated ISA of every pair_kernel instantiation (along every
path of its control-flow graph) and reports any instruction that reads or writes a VGPR which is still
the destination of an LDS read in flight.  Exit status 1 on a finding.

usage: python scripts/check_async_lds.py [file.s ...]
Without an argument it checks prediff_amd/csrc/pair_block.isa.s and pair_block_f16.isa.s: the device assembly the Makefile keeps from the
very compilations that produced pair_block.o / pair_block_f16.o (hipcc -save-temps with the Makefile's HIPCC, ARCH and CXXFLAGS), i.e. the
code that is in libprediff_hip.so -- not a re-compilation with flags of its own."""
import os
import re
import sys

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


def parse_functions(path):
    """-> {kernel name: [(line number, text)]} for every pair_kernel instantiation (labels kept, comments and directives dropped)"""
    fns, fn = {}, None
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":") and not line.startswith("."):
            fn = line[:-1] if "pair_kernel" in line else None
            if fn:
                fns[fn] = []
            continue
        if fn is None:
            continue
        if line.startswith(".") and not line.endswith(":"):
            continue                                   # assembler directive
        fns[fn].append((ln, line))
        if line.split()[0] == "s_endpgm":
            fn = None
    return fns


def check(path):
    """Walks the control-flow graph of every instantiation (basic blocks split at labels and branches; the state carried along an edge is
    the in-order queue of LDS operations in flight) -- block PLACEMENT in the file says nothing about what runs after what."""
    findings, nreads = [], 0
    for m in re.finditer(r"\.private_segment_fixed_size:\s*(\d+)", open(path).read()):
        if int(m.group(1)) != 0:
            findings.append(f"{path}: a kernel uses {m.group(1)} bytes of scratch per lane (register spills: VMEM traffic the counted waits do not cover)")
    for fn, lines in parse_functions(path).items():
        label_at = {l[:-1]: k for k, (_, l) in enumerate(lines) if l.endswith(":")}
        seen, reported, reads = set(), set(), set()
        work = [(0, ())]
        while work:
            k, pending = work.pop()
            pending = list(pending)
            while k < len(lines):
                ln, line = lines[k]
                if line.endswith(":"):                  # a label: a join point -- continue only with a state not seen here yet
                    key = (k, tuple(pending))
                    if key in seen:
                        break
                    seen.add(key)
                    k += 1
                    continue
                op = line.split()[0]
                ops = line[len(op):]
                if op.startswith("s_waitcnt"):
                    m = re.search(r"lgkmcnt\((\d+)\)", line)
                    if m:
                        pending = pending[max(0, len(pending) - int(m.group(1))):]
                    k += 1
                    continue
                if op in ("s_branch", "s_endpgm") or op.startswith("s_cbranch"):
                    if op != "s_endpgm":
                        tgt = ops.strip()
                        if tgt in label_at:
                            work.append((label_at[tgt], tuple(pending)))
                    if op.startswith("s_cbranch"):
                        k += 1
                        continue
                    break
                touched = vregs(ops)
                for dest in pending:
                    if dest & touched and ln not in reported:
                        reported.add(ln)
                        findings.append(f"{path}:{ln}: {fn}: `{line}` touches v{sorted(dest & touched)} while an LDS read into it is in flight")
                if op.startswith("ds_read"):
                    pending.append(frozenset(vregs(ops.split(",")[0])))
                    reads.add(ln)
                elif op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
                    pending.append(frozenset())        # counted by lgkmcnt, no vector destination
                del pending[:-15]                      # (the 4-bit counter: a 16th operation is not issued before the oldest returned)
                k += 1
        nreads += len(reads)
    return findings, nreads


def main():
    if len(sys.argv) > 1:
        paths = sys.argv[1:]
    else:
        csrc = os.path.join(ROOT, "prediff_amd", "csrc")
        paths = [os.path.join(csrc, "pair_block.isa.s"), os.path.join(csrc, "pair_block_f16.isa.s")]
        for path in paths:
            obj = path.replace(".isa.s", ".o")
            if not os.path.exists(path) or (os.path.exists(obj) and os.path.getmtime(path) + 60 < os.path.getmtime(obj)):
                print(f"check_async_lds: {path} is missing or older than its object: run `make -C prediff_amd/csrc` (the Makefile emits it)")
                return 1
    total, nall = 0, 0
    for path in paths:
        findings, nreads = check(path)
        for f in findings[:40]:
            print(f)
        if nreads == 0:
            print(f"check_async_lds: {path}: no pair_kernel instantiation found")
            return 1
        total += len(findings)
        nall += nreads
    print(f"check_async_lds: {len(paths)} file(s), {nall} LDS reads replayed, {total} finding(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
