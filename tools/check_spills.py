#!/usr/bin/env python3
"""Build-time guard (jpeg-encoder_amd/csrc/build.sh): fails the build when a k_group_code or k_blocks_fast instantiation
spills SGPRs AND ALSO spills VGPRs (SGPR spills alone - also those that end up in scratch - pass every parity test).  hipcc 7.2 mis-compiles that combination (SGPR spills parked in VGPRs
that are themselves spilled: wrong scan bytes and memory faults in the SIMD-variant instantiations at a 5-wave register
budget, fused_kernel_impl.hip.h) - the waves_per_eu budgets that avoid it were picked by hand, so a compiler bump, an
EXTRA_HIPCC_FLAGS variant or a new instantiation must not bring it back unnoticed.  Also reports scratch in any scalar-variant
instantiation of the pixels -> bits kernel, and FAILS on scratch in any RGB-family instantiation (CONV = true: Rgb / Rgba / Bgr /
Bgra / CmykAsYcck, either FDCT variant) of either kernel: round 4 removed the simd variant's 68 bytes per lane there, and a
scratch allocation per wave is what a register-allocation accident looks like from outside.

usage: check_spills.py FILE...   (the stderr of hipcc -Rpass-analysis=kernel-resource-usage, one file per translation unit)"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.splitlines()
    except Exception:
        return names


def main(paths):
    kernels = {}
    for path in paths:
        cur = None
        for line in open(path, errors="replace"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = kernels.setdefault(m.group(1), {})
                continue
            m = re.search(r"remark:\s+(SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|VGPRs|Occupancy \[waves/SIMD\]): (\d+)", line)
            if m and cur is not None:
                cur[m.group(1)] = int(m.group(2))
    names = sorted(kernels)
    pretty = dict(zip(names, demangle(names)))
    bad, noted, scratchy = [], [], []
    for n in names:
        k, name = kernels[n], pretty[n]
        if "k_group_code" not in name and "k_blocks_fast" not in name:
            continue
        sg, vg, sc = k.get("SGPRs Spill", 0), k.get("VGPRs Spill", 0), k.get("ScratchSize [bytes/lane]", 0)
        rgb_family = re.search(r"<\d+, \d+, \d+, \d+, true", name) is not None
        if sg and vg:
            bad.append(f"{name}: {sg} SGPRs spilled AND {vg} VGPRs spilled ({sc} B scratch per lane)")
        elif rgb_family and sc:
            scratchy.append(f"{name}: {sc} B scratch per lane ({sg} SGPRs, {vg} VGPRs spilled)")
        elif "k_group_code" in name and sc:
            noted.append(f"{name}: {sc} B scratch per lane")
    checked = sum(1 for n in names if "k_group_code" in pretty[n] or "k_blocks_fast" in pretty[n])
    for line in noted:
        print("check_spills: note: " + line)
    if bad:
        print("check_spills: the SGPR-spill-into-spilled-VGPR combination hipcc 7.2 mis-compiles is back:", file=sys.stderr)
        for line in bad:
            print("  " + line, file=sys.stderr)
        return 1
    if scratchy:
        print("check_spills: scratch in an RGB-family instantiation:", file=sys.stderr)
        for line in scratchy:
            print("  " + line, file=sys.stderr)
        return 1
    print(f"check_spills: {checked} k_group_code / k_blocks_fast instantiations, none spills SGPRs together with VGPRs, no scratch in the RGB family")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
