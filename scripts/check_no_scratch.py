#!/usr/bin/env python3
"""No kernel of the BUILT library may use scratch memory or spill a vector register.

A spill reload is a VMEM load: the fused kernels (pair_block, attn_block, ffn, igemm*) pace their weight / operand DMA with counted
`s_waitcnt vmcnt(N)`, the queue is in order, so a reload in front of a wait drains the DMA that was meant to stay in flight (round 4: the
multi-key-tile forms of attn_block_kernel reloaded eight fragment addresses per head).  This reads the code objects inside
prediff_amd/libprediff_hip.so itself -- what the GPU box maps -- not a recompilation: `llvm-objdump --offloading` unbundles them,
`llvm-readelf --notes` prints each kernel's metadata (.private_segment_fixed_size, .vgpr_spill_count, .sgpr_spill_count, .vgpr_count).
Exit status 1 (and one line per offender) if any kernel has scratch or VGPR spills.  SGPR spills (to VGPR lanes: v_writelane, no memory)
are reported, not refused.  Usage: check_no_scratch.py [path/to/lib.so]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def kernels_of(lib):
    """[(name, scratch bytes, vgpr spills, sgpr spills, vgprs)] over every gfx950 code object bundled in `lib`."""
    tmp = tempfile.mkdtemp(prefix="pd_scratch_")
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        out = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True, capture_output=True,
                                   text=True).stdout
            # one YAML map per kernel under amdhsa.kernels; the keys of a map come in alphabetical order, `.name` is unique per map
            for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                def field(key, cast=int):
                    m = re.search(r"^\s*" + re.escape(key) + r":\s*(\S+)", block, re.M)
                    return cast(m.group(1)) if m else None
                out.append((field(".name", str), field(".private_segment_fixed_size") or 0, field(".vgpr_spill_count") or 0,
                            field(".sgpr_spill_count") or 0, field(".vgpr_count") or 0))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def demangle(names):
    try:
        r = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.split("\n")
    except Exception:
        return names


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "prediff_amd", "libprediff_hip.so")
    ks = kernels_of(lib)
    if not ks:
        print(f"check_no_scratch: no gfx950 kernels found in {lib}")
        return 1
    nice = dict(zip([k[0] for k in ks], demangle([k[0] for k in ks])))
    bad = [k for k in ks if k[1] > 0 or k[2] > 0]
    for name, scratch, vsp, ssp, vg in bad:
        print(f"check_no_scratch: {nice[name][:140]}: scratch {scratch} B, {vsp} spilled VGPRs ({vg} VGPRs)")
    sg = [k for k in ks if k[3] > 0 and k not in bad]
    print(f"check_no_scratch: {len(ks)} kernels in {os.path.basename(lib)}, {len(bad)} with scratch / VGPR spills, {len(sg)} with SGPR spills only "
          f"(lane writes, no memory), max VGPRs {max(k[4] for k in ks)}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
