#!/usr/bin/env python3
"""Writes iac_amd/csrc/lfe_chain_asm.inc: the main loop of lfe_chain_kernel (render_lfe.hpp) as ONE inline-asm statement.

The recurrence y = (u - b1*y1) - b2*y2 (reference h2m_rdr.c:1218-1222, left to right, f32) is three dependent operations
per sample however it is written; what the compiler adds around them is what this removes: with the products of a new y
by b1 and b2 made by ONE v_pk_mul_f32 that reads y where it lies in the output quad (op_sel picks the half of the pair),
a step is v_sub, v_sub, v_pk_mul and nothing else - no copies into aligned pairs, no copies into the store's quad, no
wait states behind asm statements.  Registers are assigned by hand:

  v0        lane * 16 (the lane's byte offset inside a 1 KiB row)
  v[2:3]    b1, b2
  v[4:5] v[6:7] v[8:9] v[10:11]   K0..K3: (b1*o_i, b2*o_i) of the current quad's o0..o3; between quads K3 = (b1*y1, b2*y1),
                                  K2 = (b1*y2, b2*y2)
  v[16 + 4i : 19 + 4i], i < R     the ring: quad i of a rotation, u on arrival, y when stored (in place)
  s[40:41]  first row of the rotation      s[42:43] / s[46:47]  store / load base of a group of 8 quads (offsets -4096..3072)
  s44       rotations left                  s45  byte distance to the rows the rotation prefetches (0 in the last one)

Memory: quad q of the block is the 1 KiB row at base + 1024 q (u_t, render_lfe.hpp).  A rotation = R quads; while quad i of
rotation r is computed, the loads of quads i+1.. of r and 0..i-1 of r+1 are in flight behind the stores of r (vector-memory
operations retire in order, one s_waitcnt vmcnt(2R - 8) per four quads).  The last rotation prefetches its own rows again
(in bounds, never used).
"""
import os

FMA = os.environ.get("LFE_ASM_FMA", "0") == "1"   # experiment: products as v_pk_fma_f32 with a (-0, -0) addend
R = 32            # quads per rotation (vmcnt is a 6-bit counter: 2 (R - 1) <= 63)
RING0 = 16
K = [(4, 5), (6, 7), (8, 9), (10, 11)]
NV = RING0 + 4 * R


def quad(slot, off, wait):
    a = RING0 + 4 * slot
    lines = []
    if wait is not None:
        lines.append("s_waitcnt vmcnt(%d)" % wait)
    # (src of u - b1*y1, src of - b2*y2, pair holding the new y, op_sel half, destination pair)
    steps = [
        (K[3][0], K[2][1], (a, a + 1), 0, K[0]),
        (K[0][0], K[3][1], (a, a + 1), 1, K[1]),
        (K[1][0], K[0][1], (a + 2, a + 3), 0, K[2]),
        (K[2][0], K[1][1], (a + 2, a + 3), 1, K[3]),
    ]
    for i, (p, q, pair, half, dst) in enumerate(steps):
        lines.append("v_sub_f32 v%d, v%d, v%d" % (a + i, a + i, p))
        lines.append("v_sub_f32 v%d, v%d, v%d" % (a + i, a + i, q))
        if FMA:   # y * b + (-0) is y * b bit for bit (a -0 product stays -0); v_pk_fma_f32 issues faster than v_pk_mul_f32
            lines.append("v_pk_fma_f32 v[%d:%d], v[%d:%d], v[2:3], v[12:13] op_sel:[%d,0,0] op_sel_hi:[%d,1,1]"
                         % (dst[0], dst[1], pair[0], pair[1], half, half))
        else:
            lines.append("v_pk_mul_f32 v[%d:%d], v[%d:%d], v[2:3] op_sel:[%d,0] op_sel_hi:[%d,1]"
                         % (dst[0], dst[1], pair[0], pair[1], half, half))
    lines.append("global_store_dwordx4 v0, v[%d:%d], s[42:43] offset:%d" % (a, a + 3, off))
    lines.append("global_load_dwordx4 v[%d:%d], v0, s[46:47] offset:%d" % (a, a + 3, off))
    return lines


def body():
    L = []
    L += ["s_mov_b64 s[40:41], %[base]", "s_mov_b32 s44, %[nrot]", "v_mov_b32 v0, %[voff]", "v_mov_b32 v2, %[b1]",
          "v_mov_b32 v3, %[b2]", "v_mov_b32 v12, 0x80000000", "v_mov_b32 v13, 0x80000000",
          "v_mul_f32 v10, v2, %[y1]", "v_mul_f32 v11, v3, %[y1]", "v_mul_f32 v9, v3, %[y2]",
          "s_add_u32 s46, s40, 4096", "s_addc_u32 s47, s41, 0"]
    for g in range(R // 8):      # the first rotation's rows
        for j in range(8):
            a = RING0 + 4 * (8 * g + j)
            L.append("global_load_dwordx4 v[%d:%d], v0, s[46:47] offset:%d" % (a, a + 3, (j - 4) * 1024))
        L += ["s_add_u32 s46, s46, 8192", "s_addc_u32 s47, s47, 0"]
    L += ["s_waitcnt vmcnt(0)", "Llfe_rot%=:"]
    L += ["s_sub_u32 s44, s44, 1", "s_cmp_lg_u32 s44, 0", "s_cselect_b32 s45, %d, 0" % (R * 1024),
          "s_add_u32 s42, s40, 4096", "s_addc_u32 s43, s41, 0", "s_add_u32 s46, s42, s45", "s_addc_u32 s47, s43, 0"]
    for g in range(R // 8):
        for j in range(8):
            L += quad(8 * g + j, (j - 4) * 1024, 2 * R - 8 if j % 4 == 0 else None)
        L += ["s_add_u32 s42, s42, 8192", "s_addc_u32 s43, s43, 0", "s_add_u32 s46, s46, 8192", "s_addc_u32 s47, s47, 0"]
    L += ["s_add_u32 s40, s40, %d" % (R * 1024), "s_addc_u32 s41, s41, 0", "s_cmp_lg_u32 s44, 0",
          "s_cbranch_scc1 Llfe_rot%=", "s_waitcnt vmcnt(0)"]
    last = RING0 + 4 * (R - 1)
    L += ["v_mov_b32 %%[y1], v%d" % (last + 3), "v_mov_b32 %%[y2], v%d" % (last + 2)]
    return L


def main(out=None):
    out = out or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "iac_amd", "csrc", "lfe_chain_asm.inc")
    lines = body()
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_lfe_chain_asm.py - do not edit.  The main loop of lfe_chain_kernel (render_lfe.hpp)\n")
        f.write("// as one inline-asm statement with hand-assigned registers; see the generator for the register plan.\n")
        f.write("#define IAMF_LFE_RING %d\n" % R)
        f.write("#define IAMF_LFE_ASM_BODY \\\n")
        for l in lines:
            f.write('  "%s\\n\\t" \\\n' % l.replace("%%", "%"))
        f.write('  ""\n')
        f.write("#define IAMF_LFE_ASM_CLOBBERS \\\n  ")
        names = ['"v%d"' % i for i in range(NV)] + ['"s%d"' % i for i in range(40, 48)] + ['"scc"', '"memory"']
        f.write(", ".join(names) + "\n")
    print("wrote", out, len(lines), "instructions")


if __name__ == "__main__":
    import sys
    main(sys.argv[1] if len(sys.argv) > 1 else None)
