#!/usr/bin/env python3
"""Issue time of the vector-instruction MIX of a product kernel, measured (VERDICT r5 #4a: "use MEASURED cycles per op-mix").

For every kernel named on the command line (a substring of its mangled name): the histogram of its VALU opcodes is taken
from the gfx950 assembly of pyfastx_amd/csrc/fxgpu.hip (static counts: the hot loops are nearly all of these kernels), a
probe kernel is generated whose loop body holds the same opcodes in the same proportions (~600 instructions, shuffled, on a
pool of 24 registers so that neighbours rarely depend on each other), and that probe runs with the device full (8 waves per
SIMD).  Printed per kernel:   MIX <kernel> ns_per_instr=<wall time x SIMDs / wave64 instructions issued>
-- the time one wave64 instruction of THIS mix occupies a SIMD.  bench.py's `roofline_issue` multiplies it with the
instructions per granule the PMC pass counted (SQ_INSTS_VALU).  Opcodes without a template are issued as v_add_u32 and listed.
usage: python tools/mixprobe.py k_fastq_linesE k_fastq_lines_compILb0E k_span_scanILi0E k_scan_comp   (needs a GPU)"""
import os
import random
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

# opcode -> asm text; {d} destination VGPR, {a} {b} {c} source VGPRs, vcc / s[40:41] scratch scalars
T3 = "{op} {d}, {a}, {b}, {c}"
T2 = "{op} {d}, {a}, {b}"
T1 = "{op} {d}, {a}"
TEMPL = {}
for op in ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32",
           "v_min_i32", "v_max_i32", "v_min_u32", "v_max_u32", "v_mul_u32_u24", "v_pk_max_u16", "v_pk_min_u16", "v_mul_lo_u32", "v_bcnt_u32_b32",
           "v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32"):
    TEMPL[op] = T2
for op in ("v_mov_b32", "v_not_b32", "v_ffbl_b32", "v_ffbh_u32"):
    TEMPL[op] = T1
for op in ("v_dot4_u32_u8", "v_perm_b32", "v_add3_u32", "v_lshl_or_b32", "v_lshl_add_u32", "v_and_or_b32", "v_or3_b32", "v_bfe_u32", "v_mad_u32_u24",
           "v_alignbit_b32", "v_add_lshl_u32", "v_xad_u32"):
    TEMPL[op] = T3
TEMPL["v_bitop3_b32"] = "v_bitop3_b32 {d}, {a}, {b}, {c} bitop3:0x96"
TEMPL["v_cndmask_b32"] = "v_cndmask_b32 {d}, {a}, {b}, vcc"
TEMPL["v_addc_co_u32"] = "v_addc_co_u32 {d}, vcc, {a}, {b}, vcc"
TEMPL["v_add_co_u32"] = "v_add_co_u32 {d}, vcc, {a}, {b}"
TEMPL["v_subrev_co_u32"] = "v_subrev_co_u32 {d}, vcc, {a}, {b}"
TEMPL["v_readlane_b32"] = "v_readlane_b32 s40, {a}, 5"
TEMPL["v_readfirstlane_b32"] = "v_readfirstlane_b32 s40, {a}"
TEMPL["v_lshl_add_u64"] = "v_lshl_add_u64 {d2}, {a2}, 3, {b2}"
TEMPL["v_lshlrev_b64"] = "v_lshlrev_b64 {d2}, {a}, {b2}"
TEMPL["v_lshrrev_b64"] = "v_lshrrev_b64 {d2}, {a}, {b2}"
for cmp_ in ("eq", "ne", "lt", "gt", "le", "ge"):
    for ty in ("u32", "i32", "u16", "i16"):
        TEMPL["v_cmp_%s_%s" % (cmp_, ty)] = "v_cmp_%s_%s vcc, {a}, {b}" % (cmp_, ty)
    for ty in ("u64", "i64"):
        TEMPL["v_cmp_%s_%s" % (cmp_, ty)] = "v_cmp_%s_%s vcc, {a2}, {b2}" % (cmp_, ty)


# the instructions tools/valuprobe.hip measured at ~1.8 ns per wave64 instruction and SIMD with the device full (v_dot4, v_perm, v_alignbit, v_bfe,
# v_mul_lo, v_bcnt, v_mbcnt, v_readlane + its consumer, the 64-bit shifts and adds); everything else is taken at the rate of
# v_sub_u32 / v_add_u32 (~1.05 ns) -- also what was not measured one by one, so that the sum is a LOWER bound of the mix's issue time
SLOW = ("v_dot4_u32_u8", "v_perm_b32", "v_alignbit_b32", "v_bfe_u32", "v_mul_lo_u32", "v_bcnt_u32_b32", "v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32",
        "v_readlane_b32", "v_readfirstlane_b32", "v_lshl_add_u64", "v_lshlrev_b64", "v_lshrrev_b64", "v_mad_u64_u32", "v_mul_hi_u32")


def histogram(asm, pat):
    m = re.search(r"^(_ZN2fx\d+\w*%s\w*):[^\n]*\n" % re.escape(pat), asm, re.M)
    if not m:
        raise SystemExit("no kernel matches %r" % pat)
    body = asm[m.end():asm.index(".Lfunc_end", m.end())]
    ins = [ln.split()[0] for ln in body.split("\n") if ln.startswith("\t") and not ln.strip().startswith((".", ";"))]
    valu = [re.sub(r"_e(32|64)$|_sdwa$|_dpp$", "", x) for x in ins if x.startswith("v_")]
    return m.group(1), Counter(valu), len(ins)


def body_for(hist, target=600, seed=1):
    total = sum(hist.values())
    seq, missing = [], Counter()
    for op, c in hist.items():
        k = max(1, round(c * target / total)) if c * target / total >= 0.5 else 0
        if op not in TEMPL:
            missing[op] += c
            op = "v_add_u32"
        seq += [op] * k
    rng = random.Random(seed)
    rng.shuffle(seq)
    lines = []
    for i, op in enumerate(seq):
        d = 8 + (i * 7) % 24                                   # v8..v31: singles; pairs from even registers
        a, b, c = 8 + rng.randrange(24), 8 + rng.randrange(24), 8 + rng.randrange(24)
        ev = lambda x: 8 + ((x - 8) // 2 * 2) % 24             # noqa: E731
        txt = TEMPL[op].format(op=op, d="v%d" % d, a="v%d" % a, b="v%d" % b, c="v%d" % c,
                               d2="v[%d:%d]" % (ev(d), ev(d) + 1), a2="v[%d:%d]" % (ev(a), ev(a) + 1), b2="v[%d:%d]" % (ev(b), ev(b) + 1))
        lines.append(txt)
    return lines, missing


def main():
    pats = sys.argv[1:] or ["k_fastq_linesE", "k_fastq_lines_compILb0E", "k_span_scanILi0E", "k_scan_comp"]
    asm_path = "/tmp/fxgpu_mix.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-Wno-everything", "-S", "-o", asm_path,
                           os.path.join(ROOT, "pyfastx_amd", "csrc", "fxgpu.hip")], stderr=subprocess.DEVNULL)
    asm = open(asm_path).read()
    src = ["#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <cstdint>"]
    meta = []
    for i, pat in enumerate(pats):
        name, hist, n_all = histogram(asm, pat)
        lines, missing = body_for(hist, seed=i + 1)
        meta.append((pat, name, hist, len(lines), missing, n_all))
        clob = ", ".join('"v%d"' % r for r in range(8, 32)) + ', "vcc", "s40", "s41"'
        asm_txt = "\\n\\t".join(lines)
        src.append("__global__ __launch_bounds__(256) void k_mix_%d(uint32_t *out, int iters) {\n"
                   "  for (int it = 0; it < iters; ++it) { asm volatile(\"%s\" ::: %s); }\n"
                   "  if (iters < 0) out[threadIdx.x] = 1;\n}" % (i, asm_txt, clob))
    src.append("int main() {\n  uint32_t *d; if (hipMalloc((void **)&d, 4096) != hipSuccess) return 1;\n  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);\n"
               "  const int iters = 400, blocks = 2048;")
    for i, (pat, name, hist, nl, missing, n_all) in enumerate(meta):
        src.append("  { float best = 1e9f; for (int r = 0; r < 4; ++r) { hipEventRecord(a, 0); hipLaunchKernelGGL(k_mix_%d, dim3(blocks), dim3(256), 0, 0, d, iters);"
                   " hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }\n"
                   "    const double instr_per_simd = (double)blocks * 4 / 1024.0 * iters * %d;\n"
                   "    printf(\"MIX %s ns_per_instr=%%.4f slow_class_fraction=%.4f  (probe: %d instructions per iteration, %%.3f ms)\\n\", best * 1e6 / instr_per_simd, best); }"
                   % (i, nl, pat, sum(c for op, c in hist.items() if op in SLOW) / max(sum(hist.values()), 1), nl))
    src.append("  return 0;\n}")
    hip = "/tmp/mixprobe.hip"
    open(hip, "w").write("\n".join(src) + "\n")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-everything", "-o", "/tmp/mixprobe", hip])
    print("# tools/mixprobe.py: the VALU opcode mix of product kernels as probe kernels, device full (2048 workgroups x 256 threads = 8 waves per SIMD);")
    print("# ns_per_instr = wall time x 1024 SIMDs / wave64 VALU instructions issued = the time one instruction of the mix occupies a SIMD")
    for pat, name, hist, nl, missing, n_all in meta:
        top = ", ".join("%s %d" % kv for kv in hist.most_common(12))
        print("# %s: %d VALU of %d instructions in the kernel; most frequent: %s%s" % (pat, sum(hist.values()), n_all, top,
              ("; WITHOUT a template (issued as v_add_u32): " + ", ".join("%s %d" % kv for kv in missing.items())) if missing else ""))
    sys.stdout.flush()
    subprocess.check_call(["/tmp/mixprobe"])


if __name__ == "__main__":
    main()
