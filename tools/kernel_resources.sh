#!/bin/bash
# tools/kernel_resources.sh [out.txt] -- ISA evidence: registers, spills, LDS and scratch of every gfx950 kernel the library
# ships, from the code objects' own metadata (llvm-readelf --notes), plus where the SGPR spill code of a kernel sits
# (v_writelane / v_readlane counts inside vs outside its loops).  CPU only (hipcc cross-compiles).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/profiles/r05_kernel_resources.txt}
W=$(mktemp -d /tmp/xl_kres.XXXX)
LLVM=/opt/rocm/lib/llvm/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt"
{
echo "# kernel resources of sdr-server_amd/csrc/*.hip for gfx950 (hipcc $FLAGS), from llvm-readelf --notes of the code objects"
echo "# columns: vgpr agpr sgpr sgpr_spill vgpr_spill lds_bytes scratch_bytes  kernel"
for f in xl_kernels xl_polyphase xl_mixf32 xl_mixh xl_mixh2 xl_inv8 xl_inv32; do
  case $f in xl_mixh|xl_mixh2|xl_mixf32) X="-fno-slp-vectorize";; *) X="";; esac   # (csrc/Makefile: MIX_FLAGS)
  hipcc $FLAGS $X --cuda-device-only -c $ROOT/sdr-server_amd/csrc/$f.hip -o $W/$f.co
  $LLVM/clang-offload-bundler --unbundle --type=o --input=$W/$f.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$W/$f.elf
  echo "## $f.hip"
  $LLVM/llvm-readelf --notes $W/$f.elf | python3 -c '
import re, subprocess, sys
txt = sys.stdin.read()
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    agpr = blk.split()[0]
    name = g("name")
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    print("%4s %4s %4s %4s %4s %7s %6s  %s" % (g("vgpr_count"), agpr, g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"),
          g("group_segment_fixed_size"), g("private_segment_fixed_size"), name))
'
done
echo
echo "## SGPR spill code placement (v_writelane_b32 / v_readlane_b32 per kernel: total, and inside loop bodies = between a label that is"
echo "## the target of a backward branch and that branch)"
for f in xl_kernels xl_polyphase xl_mixf32 xl_mixh xl_mixh2 xl_inv8 xl_inv32; do
  $LLVM/llvm-objdump -d --no-show-raw-insn $W/$f.elf | python3 -c '
import re, sys
kern, lines = None, {}
for l in sys.stdin:
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
    if m:
        kern = m.group(1); lines[kern] = []
    elif kern and l.strip():
        lines[kern].append(l)
for k, ls in lines.items():
    if k.startswith("L") or k.startswith("."): continue
    # positions: index of each instruction; backward branches define loop bodies
    addr = {}
    ins = []
    for l in ls:
        m = re.match(r"\s*(\S.*?)\s+//\s*([0-9A-Fa-f]+):", l)
        if m: ins.append((int(m.group(2), 16), m.group(1)))
    loops = []
    for a, t in ins:
        m = re.search(r"(s_cbranch_\w+|s_branch)\s+(\d+|0x[0-9a-f]+|\S+)", t)
        if m:
            mm = re.search(r"<\S+\+0x([0-9a-f]+)>|<(\S+)>", l)
    # simpler and robust: use the branch offset encoded in the simm16 operand
    for a, t in ins:
        m = re.match(r"(s_cbranch_\w+|s_branch)\s+(-?\d+)", t)
        if m:
            off = int(m.group(2))
            off = off - 65536 if off >= 32768 else off
            tgt = a + 4 + 4 * off
            if tgt <= a: loops.append((tgt, a))
    sp = [(a, t) for a, t in ins if t.startswith("v_writelane_b32") or t.startswith("v_readlane_b32")]
    if not sp: continue
    inloop = sum(1 for a, t in sp if any(lo <= a <= hi for lo, hi in loops))
    # the HOT loops: innermost loops (no other loop inside) that are mostly packed arithmetic
    hot = []
    for lo, hi in loops:
        if any((l2, h2) != (lo, hi) and lo <= l2 and h2 <= hi for l2, h2 in loops): continue
        body = [t for a, t in ins if lo <= a <= hi]
        npk = sum(1 for t in body if t.startswith("v_pk_") or t.startswith("v_fma"))
        if npk >= 48: hot.append((len(body), npk, sum(1 for a, t in sp if lo <= a <= hi)))
    hots = "; ".join("%d instr / %d packed-FP / %d spill ops" % h for h in hot) or "none"
    print("%5d spill-lane ops, %5d inside any loop (%d loops); hot (innermost arithmetic) loops: %s  %s" % (len(sp), inloop, len(loops), hots, k))
'
done
} > $OUT
rm -rf $W
echo "wrote $OUT"
