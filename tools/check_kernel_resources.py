"""Build-time guard of the one-pass kernel's register budget (run by totsu_amd/csrc/Makefile on the remarks hipcc prints with
-Rpass-analysis=kernel-resource-usage while compiling thip_sweep.hip).

sweep_k<7,1,2,1,3> -- the headline's instance -- sits on the 256-VGPR cliff by design (two waves per SIMD share the 512
registers of a lane): 2 spilled VGPRs cost nothing measurable, 14 cost 2 %, 56 halve the rate.  A compiler
bump that moves it over the edge must fail the BUILD, not halve the bench silently:

    every sweep_k instance the library launches by default: scratch <= 16 bytes per lane and >= 2 waves per SIMD;
    SGPR spills (moves to VGPR lanes, not memory -- but each one is a VALU op on the service wave's chain, and their growth
    is how a compiler bump shows first): at most MAX_SGPR_SPILL[element kind] -- the round-4 build's worst instances (f32: 78,
    16-bit: 117) plus a margin wide enough for the +-25 that any edit of the kernel's prologue moves them by.

usage: check_kernel_resources.py <remarks file> [--report]"""
import re
import sys

# <slots, columns per panel, LAGL, DLAG, LS, element kind (0 f32, 1 bf16, 2 f16)>
# instances that are experiment variants only (thip_sweep_test.variant): reported, not enforced
VARIANTS = set()
MAX_SCRATCH, MIN_OCC, MAX_SGPR_SPILL = 16, 2, {0: 128, 1: 160, 2: 160}


def parse(txt):
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("vgpr_spill", r"VGPRs Spill: (\d+)"), ("sgpr_spill", r"SGPRs Spill: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                out[cur][key] = int(m.group(1))
    return out


def main():
    txt = open(sys.argv[1]).read()
    res = parse(txt)
    bad, seen = [], 0
    for name, r in sorted(res.items()):
        m = re.search(r"sweep_kILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
        if not m or "scratch" not in r:
            continue
        inst = tuple(int(v) for v in m.groups())
        seen += 1
        enforced = inst not in VARIANTS
        ok = r["scratch"] <= MAX_SCRATCH and r.get("occupancy", 0) >= MIN_OCC and r.get("sgpr_spill", 0) <= MAX_SGPR_SPILL[inst[5]]
        if "--report" in sys.argv or not ok:
            print("sweep_k<%s>: %d VGPRs, %d spilled, %d SGPRs spilled, scratch %d B/lane, %d waves/SIMD%s"
                  % (",".join(map(str, inst)), r.get("vgprs", -1), r.get("vgpr_spill", -1), r.get("sgpr_spill", -1), r["scratch"],
                     r.get("occupancy", -1), "" if enforced else "  (experiment variant: not enforced)"))
        if enforced and not ok:
            bad.append(inst)
    if seen == 0:
        print("check_kernel_resources: no sweep_k instance in the remarks -- was -Rpass-analysis=kernel-resource-usage passed?")
        return 2
    if bad:
        print("check_kernel_resources: FAILED for %s: scratch > %d B/lane, < %d waves/SIMD or > %s spilled SGPRs -- the one-pass "
              "kernel has fallen off its register budget with this compiler (thip_sweep.hip; DESIGN.md 4.7)"
              % (bad, MAX_SCRATCH, MIN_OCC, sorted(set(MAX_SGPR_SPILL.values()))))
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
