#!/usr/bin/env python3
"""tools/check_mfma_overlap.py [file.hip ...] -- CPU: compile kernel files to gfx950 assembly and list every matrix instruction whose result
registers overlap its A or B operand registers.  The register allocator produces such an instruction when the C operand is the constant 0
(the result is tied to no source) and an operand dies at the instruction: `v_mfma_f32_32x32x16_f16 v[0:15], v[0:3], v[160:163], 0`.  The
narrow two-half mix has carried one per pass since round 3 with every client of every parity run right; round 6 suspected it of a wrong-result
hazard in xl_mixh2.hip and cleared it (the cause was the read of the sums too close behind the last matrix instruction:
profiles/r06_mix_wide_result_hazard.txt).  A listing tool, kept for the next suspicion.  Exit code 1 if any."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "--cuda-device-only", "-S"]
PAT = re.compile(r"\s*(v_mfma_\S+)\s+[va]\[(\d+):(\d+)\], [va]\[(\d+):(\d+)\], [va]\[(\d+):(\d+)\]")


def scan(path):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["hipcc"] + FLAGS + ["-fno-slp-vectorize"] + [path, "-o", out], check=True, capture_output=True)  # (csrc/Makefile: MIX_FLAGS)
        total, bad, kernel = 0, [], "?"
        for line in open(out):
            if line.startswith("_Z") and line.rstrip().endswith(":") is False and ":" in line:
                kernel = line.split(":")[0]
            m = PAT.match(line)
            if not m:
                continue
            total += 1
            d0, d1, a0, a1, b0, b1 = map(int, m.groups()[1:])
            if not (a1 < d0 or a0 > d1) or not (b1 < d0 or b0 > d1):
                bad.append((kernel, line.strip()))
        return total, bad


def main():
    files = sys.argv[1:] or [os.path.join(ROOT, "sdr-server_amd", "csrc", f) for f in ("xl_mixh.hip", "xl_mixh2.hip", "xl_mixf32.hip")]
    rc = 0
    for f in files:
        total, bad = scan(f)
        print(f"{os.path.basename(f)}: {total} matrix instructions, {len(bad)} with a result overlapping an A / B operand")
        for k, l in bad[:10]:
            print("   ", k, "|", l)
        rc |= 1 if bad else 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
