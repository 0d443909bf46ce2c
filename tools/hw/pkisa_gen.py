"""DESIGN.md 4.7, ISA-level bisect.  Takes the device assembly of round 1's detect_ops.hip built WITH packed-f32 ops
(`hipcc ... --cuda-device-only -S`; its crop_resize_norm is the kernel that fails beside the conv kernels) and writes
variants in which chosen v_pk_*_f32 instructions of that kernel are replaced by two scalar ops through temporaries
(all sources are read before any result is written, as the packed op does), or padded with s_nop.  Each variant is
assembled into a code object (tools/hw/pkco/<name>.co) that tools/hw/pkisa_run.py launches through the module API.
usage: pkisa_gen.py <r1.s>"""
import os, re, subprocess, sys

SRC = sys.argv[1]
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pkco")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_Z16crop_resize_normPKhiiPKfPKiiiPf"
lines = open(SRC).read().split("\n")
k0 = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
k1 = next(i for i in range(k0, len(lines)) if "s_endpgm" in lines[i])
d0 = next(i for i, l in enumerate(lines) if ".amdhsa_kernel " + KERNEL in l)
pk = [i for i in range(k0, k1) if re.search(r"\bv_pk_(mul|add)_f32\b", lines[i])]
mov = [i for i in range(k0, k1) if "v_pk_mov_b32" in lines[i]]
print("packed mul/add at", [i - k0 for i in pk], "pk_mov at", [i - k0 for i in mov])


def parse(l):
    m = re.match(r"\s*v_pk_(mul|add)_f32 v\[(\d+):(\d+)\], ([vs])\[(\d+):(\d+)\], ([vs])\[(\d+):(\d+)\](.*)", l)
    op, d, _, t0, a, _, t1, b, _, rest = m.groups()
    sel = re.search(r"op_sel:\[(\d),(\d)\]", rest)
    selh = re.search(r"op_sel_hi:\[(\d),(\d)\]", rest)
    sel = [int(x) for x in sel.groups()] if sel else [0, 0]
    selh = [int(x) for x in selh.groups()] if selh else [1, 1]
    return op, int(d), (t0, int(a)), (t1, int(b)), sel, selh


def scalarise(l):
    op, d, (t0, a), (t1, b), sel, selh = parse(l)
    r = lambda t, base, h: f"{t}{base + h}"
    out = [f"\tv_{op}_f32_e64 v32, {r(t0, a, sel[0])}, {r(t1, b, sel[1])}",
           f"\tv_{op}_f32_e64 v33, {r(t0, a, selh[0])}, {r(t1, b, selh[1])}",
           f"\tv_mov_b32_e32 v{d}, v32", f"\tv_mov_b32_e32 v{d + 1}, v33"]
    return out


def is_sgpr(l):
    return " s[" in l


def is_cross(l):
    op, d, s0, s1, sel, selh = parse(l)
    return sel != [0, 0] or selh != [1, 1]


def variant(name, replace=(), nop_after=(), nop="s_nop 7"):
    out = list(lines)
    for i in sorted(set(replace) | set(nop_after), reverse=True):
        new = scalarise(lines[i]) if i in replace else [lines[i]]
        if i in nop_after:
            new = new + ["\t" + nop]
        out[i:i + 1] = new
    # two temporaries: v32, v33
    for j in range(d0, d0 + 60):
        if ".amdhsa_next_free_vgpr" in out[j + (len(out) - len(lines))] if False else False:
            pass
    txt = "\n".join(out)
    seg = txt[txt.index(".amdhsa_kernel " + KERNEL):]
    seg2 = re.sub(r"\.amdhsa_next_free_vgpr \d+", ".amdhsa_next_free_vgpr 40", seg, count=1)
    seg2 = re.sub(r"\.amdhsa_accum_offset \d+", ".amdhsa_accum_offset 40", seg2, count=1)
    txt = txt[:txt.index(".amdhsa_kernel " + KERNEL)] + seg2
    s = os.path.join(OUT, name + ".s")
    open(s, "w").write(txt)
    subprocess.check_call([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s,
                           "-o", s[:-2] + ".o"])
    subprocess.check_call([f"{LLVM}/ld.lld", "-shared", s[:-2] + ".o", "-o", s[:-2] + ".co"])
    os.remove(s[:-2] + ".o")
    print("built", name, "replaced", len(replace), "nop after", len(nop_after))


sg = [i for i in pk if is_sgpr(lines[i])]
cr = [i for i in pk if is_cross(lines[i]) and not is_sgpr(lines[i])]
plain = [i for i in pk if i not in sg and i not in cr]
print("sgpr-source:", [i - k0 for i in sg], "cross-half:", [i - k0 for i in cr], "plain:", [i - k0 for i in plain])

def swapfix(l):
    """keep the op packed but feed it through a half-swapped COPY of the operand instead of op_sel"""
    op, d, (t0, a), (t1, b), sel, selh = parse(l)
    # build temp pair v[32:33] = (src1[sel[1]], src1[selh[1]]) and v[34:35] = (src0[sel[0]], src0[selh[0]])
    return [f"\tv_mov_b32_e32 v32, {t1}{b + sel[1]}", f"\tv_mov_b32_e32 v33, {t1}{b + selh[1]}",
            f"\tv_mov_b32_e32 v34, {t0}{a + sel[0]}", f"\tv_mov_b32_e32 v35, {t0}{a + selh[0]}",
            f"\tv_pk_{op}_f32 v[{d}:{d + 1}], v[34:35], v[32:33]"]


import sys
mode = sys.argv[2] if len(sys.argv) > 2 else "round2"
for f in os.listdir(OUT):
    os.remove(os.path.join(OUT, f))
variant("w0_as_built")
for k, i in enumerate(cr):
    variant(f"w1_only_cross{k}_scalar_line{i - k0}", replace=[i])
for k, i in enumerate(cr):
    variant(f"w2_all_cross_but{k}_scalar_line{i - k0}", replace=[j for j in cr if j != i])
_scalarise = scalarise
scalarise = swapfix
variant("w3_cross_kept_packed_operands_preswapped", replace=cr)
scalarise = _scalarise
