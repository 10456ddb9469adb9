#!/usr/bin/env python3
"""Round 5, the GroupNorm-fold failures (profiles/r05_gnfold_select.txt): run HAND-EDITED assembly of the failing probe build, one edit at a time.

  in the container (needs the ROCm LLVM tools):   python tools/gnf_asm_probe.py make     -> tools/micro/bin/gnfasm/<variant>.co  (+ variants.txt)
  on the GPU box:                                 python tools/gnf_asm_probe.py run [reps]

`make` compiles csrc/tune/gnf_form0_dbg1.hip (the product arithmetic + the dump of every thread's staged values: it fails as compiled) to assembly, applies each
variant's edit to the loop head of split_linear_h3l_kernel<NCHW, GNF> and assembles a code object; `run` loads each with hipModuleLoad, launches the kernel on the
probe problem of tools/gnfold_probe.py and counts wrong coefficients a = gamma * rstd (from the dump) and wrong output pixels."""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "micro", "bin", "gnfasm")           # git-ignored (*.co), but it travels to the GPU box
KERNEL = "_ZN12_GLOBAL__N_123split_linear_h3l_kernelILi0ELi4ELi0ELb0ELb0ELb0ELb1ELb1ELb0EEEvPKfPKDv4_jS2_PfiiiiiPyNS_9ConvShapeES2_iNS_6GnFoldENS_9GnMomentsE"
LLVM = "/opt/rocm/lib/llvm/bin"

PK1 = "\tv_pk_mul_f32 v[166:167], v[130:131], v[182:183] op_sel:[0,1]\n"
PK2 = "\tv_pk_mul_f32 v[168:169], v[132:133], v[182:183] op_sel:[0,1]\n"
ST4 = "\tds_write_b128 v197, v[166:169] offset:12288\n"


def variants(head):
    """name -> edited loop head.  `head` = the text from the loop label to the first v_pk_fma (contains the four weight stores and the two packed multiplies)."""
    assert PK1 in head and PK2 in head and ST4 in head, "the compiler's output changed: re-derive the edits"
    v = {"base": head}
    v["nop_before_pk"] = head.replace(PK1, "\ts_nop 7\n\ts_nop 7\n" + PK1)
    v["one_nop_before_pk"] = head.replace(PK1, "\ts_nop 0\n" + PK1)
    v["nop_between_pk"] = head.replace(PK2, "\ts_nop 7\n" + PK2)
    v["plain_mul"] = head.replace(PK1, "\tv_mul_f32 v166, v130, v183\n\tv_mul_f32 v167, v131, v183\n").replace(PK2, "\tv_mul_f32 v168, v132, v183\n\tv_mul_f32 v169, v133, v183\n")
    v["plain_mul_first_only"] = head.replace(PK1, "\tv_mul_f32 v166, v130, v183\n\tv_mul_f32 v167, v131, v183\n")
    v["pk_before_store4"] = head.replace(PK1, "").replace(PK2, "").replace(ST4, ST4.replace("v[166:169]", "v[166:169]") )   # placeholder, replaced below
    # the packed multiplies into OTHER registers (not the 4th store's data registers), results moved afterwards
    v["pk_other_dest"] = head.replace(PK1, PK1.replace("v[166:167]", "v[220:221]")).replace(PK2, PK2.replace("v[168:169]", "v[222:223]") +
                                      "\tv_mov_b32 v166, v220\n\tv_mov_b32 v167, v221\n\tv_mov_b32 v168, v222\n\tv_mov_b32 v169, v223\n")
    # lgkmcnt(0) after the weight stores (the stores have read their data before the packed instruction issues)
    v["wait_lgkm_after_stores"] = head.replace(ST4, ST4 + "\ts_waitcnt lgkmcnt(0)\n")
    # all loads landed before the packed instruction
    v["wait_vm0_before_pk"] = head.replace(PK1, "\ts_waitcnt vmcnt(0)\n" + PK1)
    # the same op_sel, the (mean, rstd) pair copied by plain VALU first
    v["copy_pair_first"] = head.replace(PK1, "\tv_mov_b32 v220, v182\n\tv_mov_b32 v221, v183\n" + PK1.replace("v[182:183]", "v[220:221]")).replace(PK2, PK2.replace("v[182:183]", "v[220:221]"))
    # no op_sel: rstd broadcast into both halves of a pair first
    v["no_op_sel"] = head.replace(PK1, "\tv_mov_b32 v220, v183\n\tv_mov_b32 v221, v183\n\tv_pk_mul_f32 v[166:167], v[130:131], v[220:221]\n").replace(PK2, "\tv_pk_mul_f32 v[168:169], v[132:133], v[220:221]\n")
    del v["pk_before_store4"]
    # the matrix pipe drained first: 40 wait states, then a read of the last MFMA's accumulator (v[18:33]: issued right before the loop's closing branch)
    v["mfma_drained_before_pk"] = head.replace(PK1, "\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 v220, v18\n\tv_mov_b32 v221, v50\n\tv_mov_b32 v222, v82\n" + PK1)
    # the same products with the operands swapped: op_sel:[1,0] (src0 = (mean, rstd), its HIGH half for the low product)
    v["swapped_op_sel_10"] = head.replace(PK1, "\tv_pk_mul_f32 v[166:167], v[182:183], v[130:131] op_sel:[1,0]\n").replace(PK2, "\tv_pk_mul_f32 v[168:169], v[182:183], v[132:133] op_sel:[1,0]\n")
    # rstd moved into the LOW half of a pair by plain VALU, broadcast with op_sel_hi (low register for both products): the form the passing builds use
    v["op_sel_hi_broadcast"] = head.replace(PK1, "\tv_mov_b32 v220, v183\n\tv_pk_mul_f32 v[166:167], v[220:221], v[130:131] op_sel_hi:[0,1]\n").replace(PK2, "\tv_pk_mul_f32 v[168:169], v[220:221], v[132:133] op_sel_hi:[0,1]\n")
    # ---- what else does it need?  Parts of the loop removed (the GEMM's result is then garbage; the dumped coefficient a does not depend on it)
    drop = lambda pat: "".join(l for l in head.splitlines(True) if not re.search(pat, l))
    v["loop_without_mfma"] = drop(r"^\tv_mfma_")
    v["loop_without_ds_read"] = drop(r"^\tds_read_b128")
    v["loop_without_ds_write"] = drop(r"^\tds_write_b128")
    v["loop_without_barrier"] = drop(r"^\ts_barrier")
    v["loop_without_mfma_ds"] = drop(r"^\tv_mfma_|^\tds_read_b128|^\tds_write_b128|^\ts_barrier")
    # ---- towards a minimal loop: MFMAs kept, everything else that can go removed step by step (every s_waitcnt vmcnt(n) becomes vmcnt(0) once loads are dropped)
    DS = r"^\tds_read_b128|^\tds_write_b128|^\ts_barrier"
    ROWLOADS = r"^\tglobal_load_dwordx4 v\[(146:149|162:165|150:153|166:169|138:141|142:145|158:161|154:157)\]"
    SPLIT = r"^\tv_cvt_pk_f16_f32|^\tv_fma_mix(lo|hi)_f16"
    vm0 = lambda t: re.sub(r"s_waitcnt vmcnt\(\d+\)", "s_waitcnt vmcnt(0)", t)
    v["min_A_no_lds"] = drop(DS)
    v["min_B_no_lds_no_row_weight_loads"] = vm0(drop(DS + "|" + ROWLOADS))
    v["min_C_B_no_split_arith"] = vm0(drop(DS + "|" + ROWLOADS + "|" + SPLIT))
    keep = [0]
    def sixth(l):
        if re.search(r"^\tv_mfma_", l):
            keep[0] += 1
            return keep[0] % 6 == 1
        return True
    v["min_D_every_6th_mfma"] = "".join(l for l in head.splitlines(True) if sixth(l))
    # ---- victim and aggressor apart: the loop reduced to its scalar control flow + MFMAs (blocks 1-7 are then not staged at all: only block 0, staged by the
    # untouched prologue while the CU's OTHER workgroup runs this loop, is meaningful)
    only = lambda pat: "".join(l for l in head.splitlines(True) if (not l.startswith("\t")) or re.search(pat, l))
    v["loop_only_mfma"] = only(r"^\ts_(?!barrier|waitcnt)|^\tv_mfma_")
    v["loop_only_scalar"] = only(r"^\ts_(?!barrier|waitcnt)")
    MINC = DS + "|" + ROWLOADS + "|" + SPLIT
    v["aggr_minC_no_stores"] = vm0(drop(MINC + r"|^\tglobal_store"))
    v["aggr_minC_no_fold_arith"] = vm0(drop(MINC + r"|^\tv_pk_(mul|fma|add)_f32|^\tv_fma_f32|^\tv_max_f32|^\tv_add_f32|^\tv_fmac_f32"))
    v["aggr_minC_no_loads"] = vm0(drop(MINC + r"|^\tglobal_load"))
    v["aggr_minC_no_stores_no_loads"] = vm0(drop(MINC + r"|^\tglobal_load|^\tglobal_store"))
    # ---- where does the zero come from?  The dump's `a` slot (v[166:169], stored at offset:128) is replaced by a SNAPSHOT of gamma (v[130:133]) ...
    GLOAD = "\tglobal_load_dwordx4 v[130:133], v[136:137], off\n"
    DUMP_A = "\tglobal_store_dwordx4 v[180:181], v[166:169], off offset:128\n"
    assert GLOAD in head and DUMP_A in head
    dump_snap = DUMP_A.replace("v[166:169]", "v[220:223]")
    snap = "\tv_mov_b32 v220, v130\n\tv_mov_b32 v221, v131\n\tv_mov_b32 v222, v132\n\tv_mov_b32 v223, v133\n"
    # ... taken at the loop top, right behind the wait that covers the gamma load
    v["snap_gamma_at_top"] = head.replace(PK1, snap + PK1).replace(DUMP_A, dump_snap)
    # ... taken right behind the load itself (wait for it there): what the load delivered
    v["snap_gamma_at_load"] = head.replace(GLOAD, GLOAD + "\ts_waitcnt vmcnt(0)\n" + snap).replace(DUMP_A, dump_snap)
    # ... taken at the loop top but stored in the dump's FIRST raw-row slot (offset 0), so that `a` stays visible next to it
    DUMP_X0 = "\tglobal_store_dwordx4 v[180:181], v[138:141], off\n"
    assert DUMP_X0 in head
    v["snap_gamma_beside_a"] = head.replace(PK1, snap + PK1).replace(DUMP_X0, DUMP_X0.replace("v[138:141]", "v[220:223]"))
    # gamma loaded into spare registers instead of v[130:133] (the B operand of an MFMA issued five instructions before the load)
    # 16 wait states between that MFMA's neighbourhood and the load
    v["nops_before_gamma_load"] = head.replace(GLOAD, "\ts_nop 7\n\ts_nop 7\n" + GLOAD)
    return v


def make():
    os.makedirs(OUT, exist_ok=True)
    tmp = os.path.join(OUT, "tmp")
    os.makedirs(tmp, exist_ok=True)
    src = os.path.join(ROOT, "rba_amd", "csrc", "tune", "gnf_form0_dbg1.hip")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result", "-DRBA_TUNE_KNOBS", "-save-temps=obj",
                    "-c", src, "-o", os.path.join(tmp, "x.o")], check=True, cwd=os.path.join(ROOT, "rba_amd", "csrc"), stderr=subprocess.DEVNULL)
    s = open(os.path.join(tmp, "gnf_form0_dbg1-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    k0 = s.index("\n" + KERNEL + ":")
    l0 = s.index("; =>This Inner Loop Header", k0)
    l0 = s.rindex("\n.LBB", k0, l0)
    l1 = s.index("s_cbranch_scc1", s.index("s_barrier", l0))                          # the whole loop body, up to its closing branch
    head = s[l0:l1]
    # four spare registers for the edits: the kernel allocates 219 (accum_offset 220, no AGPRs)
    d0 = s.index(".amdhsa_kernel " + KERNEL)
    d1 = s.index(".end_amdhsa_kernel", d0)
    desc = s[d0:d1]
    assert ".amdhsa_next_free_vgpr 219" in desc and ".amdhsa_accum_offset 220" in desc, "register count changed: re-derive the spare registers"
    s = s[:d0] + desc.replace(".amdhsa_next_free_vgpr 219", ".amdhsa_next_free_vgpr 224").replace(".amdhsa_accum_offset 220", ".amdhsa_accum_offset 224") + s[d1:]
    names = []
    for name, edited in variants(head).items():
        t = s[:l0] + edited + s[l1:]
        sp, op, cp = (os.path.join(tmp, name + e) for e in (".s", ".o", ".co"))
        open(sp, "w").write(t)
        subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", sp, "-o", op], check=True)
        subprocess.run([LLVM + "/ld.lld", "-shared", op, "-o", os.path.join(OUT, name + ".co")], check=True)
        names.append(name)
    open(os.path.join(OUT, "variants.txt"), "w").write("\n".join(names) + "\n")
    open(os.path.join(OUT, "loop_head_base.s"), "w").write(head)
    print("built", names)


class ConvShape(ctypes.Structure):
    _fields_ = [("H", ctypes.c_int), ("W", ctypes.c_int), ("Cin", ctypes.c_int)]


class GnFold(ctypes.Structure):
    _fields_ = [("mr", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("G", ctypes.c_int), ("cpg", ctypes.c_int), ("relu", ctypes.c_int)]


class GnMoments(ctypes.Structure):
    _fields_ = [("ws", ctypes.c_void_p), ("G", ctypes.c_int), ("cpg", ctypes.c_int), ("P", ctypes.c_int)]


def run(reps):
    sys.path.insert(0, ROOT)
    import torch
    import torch.nn.functional as F
    from rba_amd import ops

    hip = ctypes.CDLL("libamdhip64.so")
    g = torch.Generator().manual_seed(1 + 131072 + 256 + 256)
    B, P, K, N, G = 1, 131072, 256, 256, 32
    x = torch.randn(B, P, K, generator=g) * 3 + 0.7
    w, b = torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g)
    ga, be = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g)
    xd = x.cuda()
    p3 = ops.split_weight(w.cuda(), mode="f16x3")
    gac, bec, bc = ga.cuda(), be.cuda(), b.cuda()
    yref = F.group_norm(xd.double().permute(0, 2, 1), G, gac.double(), bec.double(), 1e-5).permute(0, 2, 1)
    ref = (yref.reshape(B * P, K) @ w.cuda().double().t() + bc.double()).view(B, P, N).permute(0, 2, 1)
    mr = ops.group_norm_nhwc_stats(xd, G, 1e-5)
    M, NT, NB = B * P, N // 128, K // 32
    MT = M // 128
    tid = torch.arange(256, device="cuda")
    chn = (32 * torch.arange(NB, device="cuda")[:, None, None] + 4 * (tid & 7)[None, :, None] + torch.arange(4, device="cuda")[None, None, :])
    rstd = mr.view(G, 2)[:, 1]
    a_exp = gac[chn] * rstd[chn // (K // G)]
    out = torch.empty(B, N, P, device="cuda")
    dbg = torch.zeros(MT * NT * NB * 256 * 40, device="cuda")
    for name in open(os.path.join(OUT, "variants.txt")).read().split():
        mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipModuleLoad(ctypes.byref(mod), os.path.join(OUT, name + ".co").encode()) == 0, name
        assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, KERNEL.encode()) == 0, name
        res = []
        for rep in range(reps):
            out.zero_()
            dbg.zero_()
            vals = [ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(p3.data_ptr()), ctypes.c_void_p(bc.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_int(M), ctypes.c_int(N), ctypes.c_int(K),
                    ctypes.c_int(MT), ctypes.c_int(NT), ctypes.c_void_p(dbg.data_ptr()), ConvShape(0, 0, 0), ctypes.c_void_p(0), ctypes.c_int(P),
                    GnFold(mr.data_ptr(), gac.data_ptr(), bec.data_ptr(), G, K // G, 0), GnMoments(0, 1, 1, 128)]
            params = (ctypes.c_void_p * len(vals))(*[ctypes.cast(ctypes.pointer(v), ctypes.c_void_p) for v in vals])
            rc = hip.hipModuleLaunchKernel(fn, MT * NT, 1, 1, 256, 1, 1, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), params, None)
            assert rc == 0, (name, rc)
            torch.cuda.synchronize()
            e = (out.double() - ref).abs().amax(dim=(0, 1))
            a = dbg.view(MT, NT, NB, 256, 40)[..., 32:36]
            if name in ("snap_gamma_at_top", "snap_gamma_at_load"):                                       # the slot holds the snapshot of gamma for the block being staged
                bad_a = a != gac[chn][None, None]
                bad_a[:, :, :2] = False                                                # blocks 0 and 1 are staged from the prologue's loads (no snapshot there)
            else:
                bad_a = (a - a_exp[None, None]).abs() > 1e-5
            which = sorted(set((torch.nonzero(bad_a)[:, 4]).tolist()))
            if name == "snap_gamma_beside_a":
                snapg = dbg.view(MT, NT, NB, 256, 40)[..., 0:4]
                sel = bad_a.any(dim=-1)                                                 # (tile, block, thread) entries with a wrong coefficient
                sg, ag = snapg[sel], a[sel]
                want_g = gac[chn][None, None].expand(MT, NT, NB, 256, 4)[sel]
                res.append(f"[entries with a wrong a: {int(sel.sum())}; of these the gamma snapshot taken just before the multiply is wrong in {int((sg != want_g).any(dim=-1).sum())}; snapshot zero where a is zero: {int(((sg == 0) & (ag == 0)).sum())}]")
            if name.startswith("loop_only") or name.startswith("aggr_"):
                bad_a[:, :, 1:] = False                                                # only block 0 was staged
            per_blk = bad_a.any(dim=-1).sum(dim=(0, 1, 3)).tolist()                    # block 0 is staged by the prologue (not edited), blocks 1 .. by the loop
            res.append(f"bad pixels {int((e > 1e-3).sum())}, wrong a {int(bad_a.sum())} (components {which}; entries per k block {per_blk})")
            if rep == 0 and int(bad_a.sum()) and os.environ.get("GNF_HEX"):
                idx = torch.nonzero(bad_a)
                for (mt_, nt_, blk_, tid_, i_) in idx[:: max(1, idx.shape[0] // 12)][:12].tolist():
                    d = dbg.view(MT, NT, NB, 256, 40)[mt_, nt_, blk_, tid_]
                    av = d[32:36].view(torch.int32).tolist()
                    bv = d[36:40].tolist()
                    print(f"    tile ({mt_},{nt_}) blk {blk_} tid {tid_} i {i_}: a bits {[hex(v & 0xffffffff) for v in av]} a {d[32:36].tolist()} want {a_exp[blk_, tid_].tolist()} b {bv} beta {bec[chn[blk_, tid_]].tolist()}")
        print(f"{name:28s} " + " | ".join(res), flush=True)
        hip.hipModuleUnload(mod)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "make":
        make()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 2)
