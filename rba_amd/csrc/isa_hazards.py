"""Scan the gfx950 ISA that hipcc emitted for librba_hip.so for wait-state hazards the compiler cannot see inside inline asm.

hipcc's hazard recogniser inserts the s_nop a gfx950 wait-state rule needs between two instructions it emitted itself; it does not decode
the text of an `asm` statement, so a hand-written instruction can end up back to back with its producer or consumer.  Rules checked (each is
"one wait state": ANY instruction in between, s_nop included, satisfies it):

  D  a VALU whose destination is 16 bits of a register (v_fma_mixlo_f16 / v_fma_mixhi_f16, SDWA with dst_sel != DWORD, VOP3 op_sel with the
     destination bit) followed IMMEDIATELY by a VALU (MFMA included) that reads that register -- v_fma_mixhi_f16 / v_fma_mixlo_f16 of the same
     register count as readers (they keep the other half).  Round 3's rba_reduce_up4_mx_kernel had 40 of these (mixlo, mixhi back to back).
  T  a transcendental (v_rcp/v_rsq/v_sqrt/v_exp/v_log/v_sin/v_cos) followed IMMEDIATELY by a non-transcendental VALU that reads its result.

  P  (round 5, not a documented rule: an observation) a packed fp32 VALU (v_pk_mul/add/fma_f32) whose LOW lane takes the HIGH register of source 1 (`op_sel:[x,1,...]`).
     In the GroupNorm-folded projection `v_pk_mul_f32 vD, v_gamma, v_(mean,rstd) op_sel:[0,1]` gave a LOW product of exactly 0 in lanes 48-63 a few hundred times per
     launch; the same products as v_mul_f32, as a packed multiply on a broadcast pair, or with the select on source 0 (`op_sel:[1,0]`) never did
     (tools/gnf_asm_probe.py, profiles/r05_gnfold_select.txt); tools/micro/pk_opsel_after_load.hip reproduces it standalone, v_pk_add_f32 included (profiles/r05_pk_opsel_erratum.txt).  The library is kept free of the form: split_linear_gnf.hip is compiled without packed fp32, the one other
     producer (the LayerNorm prologue of mlp_fused_h3.h) sums channel pairs instead of quads.

Round 6: `gate(path)` is the post-link step of rba_amd/csrc/build.py for BOTH librba_hip.so and librba_hip_knobs.so -- a library with a hit is deleted and the build fails.

Usage: python tools/isa_hazards.py [lib.so | file.o ...]   (default: rba_amd/csrc/librba_hip.so); exit status 1 if anything is found.
`scan_library(path)` returns {"code_objects": n, "mix": n, "D": [...], "T": [...], "P": [...]} for tests/test_host_cpu.py."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PK_SRC1_CROSS = re.compile(r"^v_pk_(mul|add|fma)_f32\b.*\bop_sel:\[[01],1[,\]]")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag)?_(f32|f16|legacy_f32)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def disassemble(path):
    """[(code object name, [instruction text, ...]), ...] for every gfx950 code object bundled in a hipcc object / shared library"""
    tmp = tempfile.mkdtemp(prefix="isa_")
    try:
        local = os.path.join(tmp, "x.bin")
        shutil.copy(path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "x.bin"], cwd=tmp, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        out = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", os.path.join(tmp, f)], check=True,
                                 stdout=subprocess.PIPE, text=True).stdout
            ins, kernel = [], ""
            for ln in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
                if m:
                    kernel = m.group(1)
                    ins.append(("label", kernel))
                    continue
                if re.match(r"^\s+[a-z_0-9]+", ln):
                    ins.append((ln.split("//")[0].strip(), kernel))
            out.append((f, ins))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _regs(operand):
    s = set()
    for a, lo, hi in REG.findall(operand):
        if a:
            s.add(int(a))
        else:
            s.update(range(int(lo), int(hi) + 1))
    return s


def _split(text):
    mnem, _, rest = text.partition(" ")
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        if ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return mnem, ops


def _half_dst(mnem, text):
    if mnem in ("v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_fma_mixlo_bf16", "v_fma_mixhi_bf16"):
        return True
    if "_sdwa" in mnem and re.search(r"dst_sel:(WORD|BYTE)", text):
        return True
    m = re.search(r"op_sel:\[([01,]+)\]", text)
    return bool(m and len(m.group(1).split(",")) == 4 and m.group(1).split(",")[3] == "1")


def _reads(mnem, ops, text):
    """registers a VALU instruction reads: every operand but the first, plus the destination where it is also a source"""
    r = set()
    for o in ops[1:]:
        r |= _regs(o.split(" ")[0])
    if mnem.startswith(("v_fma_mixlo", "v_fma_mixhi", "v_fmac", "v_mac", "v_pk_fmac", "v_dot2c", "v_dot4c", "v_dot8c")) or "dst_unused:UNUSED_PRESERVE" in text:
        r |= _regs(ops[0])
    return r


def scan(ins):
    found = {"D": [], "T": [], "P": [], "mix": 0}
    prev = None
    for text, kernel in ins:
        if text == "label":
            prev = None
            continue
        mnem, ops = _split(text)
        if mnem.startswith("v_fma_mix"):
            found["mix"] += 1
        if PK_SRC1_CROSS.match(text):
            found["P"].append((kernel, text, ""))
        if prev is not None and mnem.startswith("v_") and ops:
            pm, pops, ptext = prev
            rd = _reads(mnem, ops, text)
            if _half_dst(pm, ptext) and (_regs(pops[0]) & rd):
                found["D"].append((kernel, ptext, text))
            if TRANS.match(pm) and not TRANS.match(mnem) and (_regs(pops[0]) & rd):
                found["T"].append((kernel, ptext, text))
        prev = (mnem, ops, text) if ops else None
    return found


def scan_library(path):
    total = {"code_objects": 0, "mix": 0, "D": [], "T": [], "P": []}
    for name, ins in disassemble(path):
        r = scan(ins)
        total["code_objects"] += 1
        total["mix"] += r["mix"]
        total["D"] += [(name,) + x for x in r["D"]]
        total["T"] += [(name,) + x for x in r["T"]]
        total["P"] += [(name,) + x for x in r["P"]]
    return total


def gate(path, rules=("D", "T", "P")):
    """post-link gate of rba_amd.csrc.build: raise (the caller removes the library) when a code object of `path` holds a form the kernels must not contain.
    Returns the scan summary; a tool chain without llvm-objdump cannot be gated and raises too -- an unscanned library must not ship silently."""
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        raise RuntimeError(f"{LLVM}/llvm-objdump not found: cannot scan {path} for the gfx950 hazards of rba_amd/csrc/isa_hazards.py")
    r = scan_library(path)
    if r["code_objects"] == 0:
        raise RuntimeError(f"{path}: no gfx950 code object found to scan")
    hits = [(k,) + x for k in rules for x in r[k]]
    if hits:
        lines = "\n".join(f"  [{k}] {kernel[:100]}: {a} ; {b}" for k, _, kernel, a, b in hits[:20])
        raise RuntimeError(f"{path}: {len(hits)} ISA hazard form(s) (D = 16-bit destination -> reader, T = transcendental -> reader, P = packed fp32 with the "
                           f"cross select on source 1, wrong results on MI355X):\n{lines}\nCompile the translation unit without packed fp32 "
                           "(build.py UNPACKED_ALWAYS) or rewrite the expression.")
    return r


def main(argv):
    paths = argv or [os.path.join(REPO, "rba_amd", "csrc", "librba_hip.so")]
    bad = 0
    for p in paths:
        r = scan_library(p)
        print(f"{p}: {r['code_objects']} code objects, {r['mix']} v_fma_mix*; hazards: 16-bit destination -> reader {len(r['D'])}, "
              f"transcendental -> reader {len(r['T'])}, packed fp32 with the cross select on source 1 {len(r['P'])}")
        for kind in ("D", "T", "P"):
            for name, kernel, a, b in r[kind][:20]:
                print(f"  [{kind}] {kernel[:90]}\n        {a}\n        {b}")
        bad += len(r["D"]) + len(r["T"]) + len(r["P"])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
