#!/usr/bin/env python3
"""ISA audit of the HIP kernels (no GPU needed): compile every csrc/*.hip to gfx950 assembly with the Makefile's flags
and report, per kernel, what round 5 found twice in hot loops without any symptom but time:
  * scratch  (private_segment_fixed_size > 0, VGPR spills) — conv_bx_kernel kept its loader state there: ~15 scratch
    accesses per K step and, on every tap wrap, a flat_load behind s_waitcnt vmcnt(0) lgkmcnt(0);
  * flat_load / flat_store — pointer provenance lost (a pointer-merged increment, a table of generic pointers);
  * compiler-inserted `s_waitcnt vmcnt(0)` inside a loop that also issues LDS-DMA loads (buffer_load ... lds) — hipcc
    drains vmcnt in front of every LDS access that MAY read what an outstanding LDS-DMA writes (intrinsics without a
    memory operand, e.g. ds_read_b64_tr_b16, always "may"): conv_bx_wgrad_kernel ran one DMA round trip per K step.
usage: tools/isa_audit.py [file.hip ...] [--all]      (default: kernels with a finding only)
exit code 1 if a kernel on the HOT list has scratch or a flat access."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "efficient-slowfast_amd", "csrc")
HOT = ("conv_bx_kernel", "conv_pw_bx_kernel", "conv_bx_wgrad_kernel", "conv_wave_kernel", "conv_wgrad_wave_kernel",
       "attn_bwd_bx_kernel", "attn_bwd_bxp_kernel", "attn_bwd_bx2_kernel", "attn_fwd_bx_kernel", "attn_fwd_bxp_kernel",
       "attn_fwd_bx2_kernel", "conv_stem", "conv_wgrad_stem", "conv_wgrad_rows_kernel", "bn_bwd_", "affine_flat")
# kernels that are allowed scratch (not launched by default / debug variants)
EXEMPT = ("conv_wave_p_kernelILi7ELi4", "conv_wave_p_kernelILi13ELi2", "attn_bwd_bxpp_kernel", "attn_bwd_fused_kernelILi64",
          "attn_bwd_dq_kernelILi128", "attn_bwd_dkv_kernelILi128", "attn_bwd_dq_kernelILi64", "attn_bwd_dkv_kernelILi64",
          "attn_bwd_bx2_kernel")


def flags_for(name):
    f = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-mllvm",
         "-amdgpu-mfma-vgpr-form", "--cuda-device-only", "-S"]
    if name == "attn_bwd.hip":
        f.append("-fno-slp-vectorize")
    return f


def audit(path, show_all):
    name = os.path.basename(path)
    with tempfile.NamedTemporaryFile(suffix=".s", delete=False) as t:
        out = t.name
    try:
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags_for(name) + [path, "-o", out], check=True, cwd=CSRC,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(out).read()
    finally:
        os.unlink(out)
    meta, cur = {}, None
    for line in text.splitlines():  # the amdhsa.kernels metadata: .name, then that kernel's counts
        t = line.strip()
        if t.startswith(".name:"):
            cur = t.split()[1]
            meta[cur] = [0, 0, 0]
        elif cur and t.startswith(".private_segment_fixed_size:"):
            meta[cur][0] = int(t.split()[1])
        elif cur and t.startswith(".vgpr_count:"):
            meta[cur][1] = int(t.split()[1])
        elif cur and t.startswith(".vgpr_spill_count:"):
            meta[cur][2] = int(t.split()[1])
    bad = 0
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        kern, body = m.group(1), m.group(2)
        scratch, vgpr, spills = meta.get(kern, (0, 0, 0))
        flat = len(re.findall(r"^\s+flat_(?:load|store)", body, re.M))
        dma = len(re.findall(r"buffer_load_\w+ .* lds$", body, re.M))
        # compiler-inserted vmcnt(0) (not inside an inline-asm block) in kernels that use LDS-DMA
        outside = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", body, flags=re.S)
        drains = len(re.findall(r"s_waitcnt vmcnt\(0\)", outside)) if dma else 0
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", kern)
        hot = any(h in short for h in HOT) and not any(e in short for e in EXEMPT)
        finding = scratch or flat or spills
        if show_all or finding or (dma and drains):
            print("%-14s %-64s vgpr %3d scratch %3d B spills %2d flat %2d lds-dma %2d compiler vmcnt(0) %3d%s" % (
                name, short[:64], vgpr, scratch, spills, flat, dma, drains, "   <-- HOT" if hot and (scratch or flat) else ""))
        if hot and (scratch or flat):
            bad += 1
    return bad


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    files = args or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    for f in files:
        bad += audit(os.path.join(CSRC, f) if not os.path.isabs(f) else f, "--all" in sys.argv)
    print("%d hot kernel(s) with scratch or flat accesses" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
