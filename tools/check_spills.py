#!/usr/bin/env python3
"""Build-time guard (jpeg-encoder_amd/csrc/build.sh): fails the build when a k_group_code or k_blocks_fast instantiation
spills SGPRs AND ALSO spills VGPRs (SGPR spills alone - also those that end up in scratch - pass every parity test).  hipcc 7.2 mis-compiles that combination (SGPR spills parked in VGPRs
that are themselves spilled: wrong scan bytes and memory faults in the SIMD-variant instantiations at a 5-wave register
budget, fused_kernel_impl.hip.h) - the waves_per_eu budgets that avoid it were picked by hand, so a compiler bump, an
EXTRA_HIPCC_FLAGS variant or a new instantiation must not bring it back unnoticed.  Also reports scratch in any scalar-variant
instantiation of the pixels -> bits kernel, and FAILS on scratch in any RGB-family instantiation (CONV = true: Rgb / Rgba / Bgr /
Bgra / CmykAsYcck, either FDCT variant) of either kernel: round 4 removed the simd variant's 68 bytes per lane there, and a
scratch allocation per wave is what a register-allocation accident looks like from outside.  One exception, listed as a note:
the six-wave layouts (2x2 sampling) of the pixels -> bits kernel at their 6-wave register budget may keep up to 32 bytes
(a few values parked once per block, outside every loop) - the budget is worth more than they cost (fused_kernel_impl.hip.h).

usage: check_spills.py FILE...   (the stderr of hipcc -Rpass-analysis=kernel-resource-usage, one file per translation unit)"""
import re
import subprocess
import sys


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
            if len(out) == len(names):
                return out
        except Exception:
            pass
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
        if "k_group_code" not in name and "k_blocks_fast" not in name and "k_blocks_444" not in name and "k_blocks_420" not in name:
            continue
        sg, vg, sc = k.get("SGPRs Spill", 0), k.get("VGPRs Spill", 0), k.get("ScratchSize [bytes/lane]", 0)
        # CONV = true (5th template argument), in the demangled or - without a demangler - the mangled name
        rgb_family = (re.search(r"<\d+, \d+, \d+, \d+, true", name) is not None or re.search(r"ILi\d+ELi\d+ELi\d+ELi\d+ELb1ELb[01]EE", n) is not None
                      or "k_blocks_444" in name or "k_blocks_420" in name)                               # (the 4:4:4 kernel of the RGB family: no scratch either)
        m = re.search(r"<\d+, (\d+), (\d+), \d+, (?:true|false)", name) or re.search(r"ILi\d+ELi(\d+)ELi(\d+)ELi\d+ELb[01]ELb[01]EE", n)
        six_wave_layout = bool(m) and int(m.group(1)) * int(m.group(2)) == 4
        if sg and vg:
            bad.append(f"{name}: {sg} SGPRs spilled AND {vg} VGPRs spilled ({sc} B scratch per lane)")
        elif rgb_family and sc and not (six_wave_layout and "k_group_code" in name and sc <= 32):
            scratchy.append(f"{name}: {sc} B scratch per lane ({sg} SGPRs, {vg} VGPRs spilled)")
        elif rgb_family and sc:
            noted.append(f"{name}: {sc} B scratch per lane at the 6-wave budget of the six-wave layouts ({vg} VGPRs spilled outside the loops; measured faster than 5 waves without)")
        elif "k_group_code" in name and sc:
            noted.append(f"{name}: {sc} B scratch per lane")
    checked = sum(1 for n in names if "k_group_code" in pretty[n] or "k_blocks_fast" in pretty[n] or "k_blocks_444" in pretty[n] or "k_blocks_420" in pretty[n])
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
    exempt = sum(1 for line in noted if "six-wave layouts" in line)
    tail = ("no scratch in the RGB family" if not exempt else
            f"scratch <= 32 B per lane in {exempt} six-wave RGB-family instantiations of k_group_code (exempt: listed above), none in the other RGB-family kernels")
    print(f"check_spills: {checked} k_group_code / k_blocks_fast / k_blocks_444 / k_blocks_420 instantiations, none spills SGPRs together with VGPRs, {tail}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
