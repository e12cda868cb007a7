#!/usr/bin/env python3
"""ISA audit of the HIP kernels (no GPU needed): compile every csrc/*.hip to gfx950 assembly with the Makefile's flags
and report, per kernel, what round 5 found twice in hot loops without any symptom but time:
  * scratch  (private_segment_fixed_size > 0, VGPR spills) — conv_bx_kernel kept its loader state there: ~15 scratch
    accesses per K step and, on every tap wrap, a flat_load behind s_waitcnt vmcnt(0) lgkmcnt(0);
  * flat_load / flat_store — pointer provenance lost (a pointer-merged increment, a table of generic pointers);
  * compiler-inserted `s_waitcnt vmcnt(0)` inside a loop that also issues LDS-DMA loads (buffer_load ... lds) — hipcc
    drains vmcnt in front of every LDS access that MAY read what an outstanding LDS-DMA writes (intrinsics without a
    memory operand, e.g. ds_read_b64_tr_b16, always "may"): conv_bx_wgrad_kernel ran one DMA round trip per K step.
  * a compiler-inserted `s_waitcnt vmcnt(0)` in a basic block that also holds MFMAs (warning only: the matrix pipe waits
    for the whole load queue there).
usage: tools/isa_audit.py [file.hip ...] [--all] [--asm-dir DIR]     (default: kernels with a finding only)
  --asm-dir DIR: read DIR/<name>-hip-amdgcn-amd-amdhsa-gfx950.s as the build left them (the Makefile compiles with
                 -save-temps=obj and runs this as part of `make`: the audit costs no second compilation)
exit code 1 if a kernel on the HOT list has scratch or a flat access; exempted hot kernels are printed as WAIVED."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "efficient-slowfast_amd", "csrc")
HOT = ("conv_rows_kernel", "conv_bx_kernel", "conv_pw_bx_kernel", "conv_bx_wgrad_kernel", "conv_wave_kernel", "conv_wgrad_wave_kernel",
       "attn_bwd_bx_kernel", "attn_bwd_bxp_kernel", "attn_bwd_bx2_kernel", "attn_fwd_bx_kernel", "attn_fwd_bxp_kernel",
       "attn_fwd_bx2_kernel", "conv_stem", "conv_wgrad_stem", "conv_wgrad_rows_kernel", "bn_bwd_", "affine_flat")
# kernels that are allowed scratch (not launched by default / debug variants)
EXEMPT = ("attn_bwd_bxpp_kernel",          # parked ping-pong variant: only behind sf_attn_tune(2, 1) / SF_ATTN_BX_PP=1
          "attn_bwd_fused_kernelILi64",    # f32-input sweep for 32 < d <= 64: only with SF_ATTN_BX=0 (A/B runs)
          "attn_bwd_dq_kernelILi64", "attn_bwd_dkv_kernelILi64")  # two-kernel f32 form for d = 64: SF_ATTN_BX=0 only


def flags_for(name):
    f = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-mllvm",
         "-amdgpu-mfma-vgpr-form", "--cuda-device-only", "-S"]
    if name == "attn_bwd.hip":
        f.append("-fno-slp-vectorize")
    return f


def audit(path, show_all, asm_dir=None):
    name = os.path.basename(path)
    if asm_dir is not None:
        text = open(os.path.join(asm_dir, name[:-len(".hip")] + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    else:
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
        # basic blocks (label to label) that hold MFMAs AND a compiler-inserted full drain of the load queue
        mixed = sum(1 for blk in re.split(r"^\.LBB\w+:", outside, flags=re.M)
                    if "v_mfma" in blk and re.search(r"s_waitcnt vmcnt\(0\)", blk))
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", kern)
        listed = any(h in short for h in HOT)
        waived = listed and any(e in short for e in EXEMPT)
        hot = listed and not waived
        finding = scratch or flat or spills
        if show_all or finding or (dma and drains) or (listed and mixed):
            note = ""
            if hot and (scratch or flat):
                note = "   <-- HOT"
            elif waived and (scratch or flat):
                note = "   WAIVED (not launched by default: see EXEMPT)"
            elif listed and mixed:
                note = "   warning: %d MFMA block(s) with vmcnt(0)" % mixed
            print("%-14s %-64s vgpr %3d scratch %3d B spills %2d flat %2d lds-dma %2d compiler vmcnt(0) %3d%s" % (
                name, short[:64], vgpr, scratch, spills, flat, dma, drains, note))
        if hot and (scratch or flat):
            bad += 1
    return bad


def main():
    argv = sys.argv[1:]
    asm_dir = None
    if "--asm-dir" in argv:
        i = argv.index("--asm-dir")
        asm_dir = argv[i + 1]
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    files = args or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    for f in files:
        bad += audit(os.path.join(CSRC, f) if not os.path.isabs(f) else f, "--all" in argv, asm_dir)
    print("%d hot kernel(s) with scratch or flat accesses" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
