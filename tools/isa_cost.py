"""VALU issue cost of a piece of gfx950 ISA, in SIMD cycles per wave: tools/isa_cost.py file.s [first_line last_line].

Weights from tools/valu_rates.hip on MI355X (profiles/r04/valu_rates.txt), two or more waves per SIMD: the VOP2 / VOP1 f32
add / sub / mul / fmac, v_fma_f32, 32-bit and / or / xor / add / sub / right shifts and v_mov_b32 occupy the SIMD for ~2.2
cycles (1 unit); v_rcp_f32 and the permlane swaps for ~8.2 (4 units); every other vector instruction — all v_pk_*,
conversions, floor / fract, min / max, left shifts, 24-bit and 32-bit multiplies, three-operand integer ops, v_alignbyte,
v_perm, SDWA / DPP forms, v_mov_b64, compares, v_readlane — for ~4.2 (2 units)."""
import collections
import re
import sys

FULL = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_fma_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mac_f32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32",
        "v_not_b32"}
QUARTER = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_permlane32_swap_b32",
           "v_permlane16_swap_b32", "v_rcp_f64", "v_mul_f64", "v_add_f64", "v_fma_f64"}


def unit(op, line):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.endswith("_dpp") or op.endswith("_sdwa") or "row_" in line or "quad_perm" in line:
        return 2
    if base in QUARTER:
        return 4
    if base in FULL:
        return 1
    return 2


def main():
    lines = open(sys.argv[1]).read().split("\n")
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(lines)
    cnt = collections.Counter()
    units = collections.Counter()
    for ln in lines[lo - 1:hi]:
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
        cnt[kind] += 1
        if kind == "valu":
            u = unit(op, t)
            units[u] += 1
    total = sum(k * v for k, v in units.items())
    print(f"lines {lo}-{hi}: VALU {cnt['valu']} instructions = {total} units ({units[1]} x1, {units[2]} x2, {units[4]} x4) "
          f"~ {total * 2.15:.0f} cycles; SALU {cnt['salu']}, LDS {cnt['lds']}, VMEM {cnt['vmem']}")


if __name__ == "__main__":
    main()
