#!/usr/bin/env python3
"""Generates rans_decode_tile.inc: the inline-asm body of the rANS decode fast tile (gfx950).

Costs measured on MI355X with scripts/probes/latency_probe.hip (one wave alone on its SIMD):
  any SALU or VALU instruction ~4 cycles; the first SALU instruction after a VALU instruction that wrote
  an SGPR (v_readlane) stalls ~20 cycles whatever it reads, while further VALU instructions do not;
  an untaken branch ~10 cycles, a taken one ~25.
Hence: exactly one VALU->SALU crossing per symbol (both table readlanes back to back, the record
v_writelane right behind them, inside the stall), no branch in the per-symbol code (renormalisation shift from an SGPR table
indexed by the leading-zero count), and the window refill test once per two symbols with the refill
itself out of line.

Register plan (all literal, all listed as clobbers by the including statement):
  v[64:127]   F'[slot] = freq << 20 (row = slot[11:6], lane = slot[5:0])
  v[128:191]  B'[slot] = slot - cum - ((freq * slot) >> 12), so that x' = umulhi(F', x) + B' = freq * (x >> 12) + slot - cum
  v[192:224]  stream window, big-endian dwords (dword d in row d >> 6, lane d & 63)
  v225        record: pre-update state of the 64 symbols of the current block
  s[60:61]    {window copy : x} shift pair      s[62:63] 64-bit big-endian byte window      s[64:65] refill pair
  s67 F'   s69 B'   s70 product   s71 shift   s72 valid window bits - 33   s73 next window dword
  s74 blocks left   s75 byte-swap selector   s76 row   s77 shift of the odd symbol   s78 scratch
  s79..s99    renormalisation shift (0, 8 or 16) by count-leading-zeros of the state (0 .. 20: the updated state is at
              least 2^11).  Not s100 and up: the compiler keeps those for itself and ignores them in a clobber list.
  s59         M0 of the surrounding code.  M0 is written by every s_set_gpr_idx_on (the row) and by s_flbit (the index of
              s_movrels); the compiler reserves it and does not honour an "m0" clobber, so the statement saves it first and
              restores it last -- whatever the compiler keeps in M0 survives, and the clobber lists need not name it.
Operands: %[tp] (SGPR pair: the chain's RansDecSlots in global memory, F' then B'), %[l4] (VGPR: 4 * lane),
          %[wa] (VGPR: per-lane LDS byte address of the window), %[ra] (VGPR in/out: record address, 2 B per lane),
          %[xi] %[pi] %[nb] (SGPR in), %[xo] %[po] (SGPR out).
The tables are re-read from global memory (L2) at the top of every tile: 128 loads per 4096 symbols, well under 1 % of a
tile's time, and nothing of them lives in LDS, which is what bounds the number of chains a CU can host.
"""
import os

WIN_ROWS = 33   # stream window: 33 VGPRs x 64 lanes x 4 B = 8448 bytes >= 2 * 4096 symbols + refill look-ahead
REC = 192 + WIN_ROWS

L = []
def e(s): L.append(s)

M0_SAVE = "s59"

def table_loads(emit):
    """F' then B' (contiguous in RansDecSlots): 128 rows of 256 bytes, row r into v[64 + r], lane l <- dword 64 r + l.
    global_load's immediate offset is 13 bits signed, so the scalar base moves on every 16 rows (the address is read
    when the load issues)."""
    emit(f"s_mov_b32 {M0_SAVE}, m0")
    emit("s_mov_b64 s[64:65], %[tp]")
    for r in range(128):
        emit(f"global_load_dword v{64 + r}, %[l4], s[64:65] offset:{256 * (r % 16)}")
        if r % 16 == 15 and r != 127:
            emit("s_add_u32 s64, s64, 0x1000")
            emit("s_addc_u32 s65, s65, 0")

table_loads(e)
for r in range(WIN_ROWS):
    e(f"ds_read_b32 v{192 + r}, %[wa] offset:{256 * r}")
e("s_mov_b32 s75, 0x00010203")
e("s_waitcnt vmcnt(0) lgkmcnt(0)")
for r in range(WIN_ROWS):
    e(f"v_perm_b32 v{192 + r}, v{192 + r}, v{192 + r}, s75")
for c in range(21):
    sh = 0 if c <= 8 else (8 if c <= 16 else 16)
    e(f"s_mov_b32 s{79 + c}, {sh}")
# scalar state: x, 64-bit window primed with two dwords
e("s_mov_b32 s61, %[xi]")
e("s_mov_b32 s74, %[nb]")
e("s_lshr_b32 s73, %[pi], 2")
e("s_and_b32 s71, %[pi], 3")
e("s_lshl_b32 s71, s71, 3")
e("s_lshr_b32 s76, s73, 6")
e("s_set_gpr_idx_on s76, 0x1")
e("s_nop 1")
e("v_readlane_b32 s63, v192, s73")
e("s_add_u32 s73, s73, 1")
e("s_lshr_b32 s76, s73, 6")
e("s_set_gpr_idx_on s76, 0x1")
e("s_nop 1")
e("v_readlane_b32 s62, v192, s73")
e("s_add_u32 s73, s73, 1")
e("s_lshl_b64 s[62:63], s[62:63], s71")
e("s_sub_u32 s72, 31, s71")   # 64 - shift - 33
# Code placement: the block loop is pinned to a 64-byte boundary + 4 bytes.  Measured (scripts/chain_probe.py): loop
# starts at 4 mod 8 bytes decode 1.3 % faster than starts at 0 mod 8, and without the pin the phase is whatever the
# compiler-generated code in front of the asm statement happens to leave.  (The assembler pads with s_nop.)
e(".p2align 6")
e("s_nop 0")
e("2:")
# s72 holds (valid window bits - 33): the borrow of the one subtraction per symbol pair is the refill condition
# Symbols go in pairs.  At the start of a pair the window holds at least 33 valid bits, so its upper dword s63 is all
# stream.  s60 takes a copy of it ONCE per pair: the even symbol's shift of {s60:s61} feeds the state and leaves
# s60 = s63 << sh with 32 - sh >= 16 valid bits on top, enough for the odd symbol (a symbol consumes at most 16 bits);
# the window itself is shifted once per pair by the sum of the two shifts, which the refill test needs anyway.
# (Half an issue slot per symbol less than copying and shifting the window for every symbol.)
for lane in range(64):
    sh = "s77" if lane & 1 else "s71"
    e("s_bfe_u32 s76, s61, 0x60006")
    e("s_set_gpr_idx_on s76, 0x1")
    e("v_readlane_b32 s67, v64, s61")
    e("v_readlane_b32 s69, v128, s61")
    e(f"v_writelane_b32 v{REC}, s61, {lane}")   # VALU work issues during the VALU->SALU stall; SALU work would not
    e("s_mul_hi_u32 s70, s67, s61")
    e("s_add_u32 s61, s70, s69")
    e("s_flbit_i32_b32 m0, s61")
    # one instruction must sit between the SALU write of M0 and s_movrels (wait state): the copy for the even symbol,
    # a no-op for the odd one
    e("s_mov_b32 s60, s63" if not lane & 1 else "s_nop 0")
    e(f"s_movrels_b32 {sh}, s79")
    e(f"s_lshl_b64 s[60:61], s[60:61], {sh}")
    if lane & 1:
        e("s_add_u32 s78, s71, s77")
        e("s_lshl_b64 s[62:63], s[62:63], s78")
        e("s_sub_u32 s72, s72, s78")            # SCC = borrow <=> fewer than 33 valid bits left
        e(f"s_cbranch_scc1 3{lane:02d}f")
        e(f"4{lane:02d}:")
e("s_set_gpr_idx_off")
e(f"ds_write_b16 %[ra], v{REC}")
e("v_add_u32_e32 %[ra], 0x80, %[ra]")
e("s_sub_u32 s74, s74, 1")
e("s_cmp_lg_u32 s74, 0")
e("s_cbranch_scc1 2b")
e("s_branch 5f")
for lane in range(1, 64, 2):   # out-of-line refills
    e(f"3{lane:02d}:")
    e("s_lshr_b32 s76, s73, 6")
    e("s_set_gpr_idx_on s76, 0x1")
    e("s_mov_b32 s64, 0")
    e("v_readlane_b32 s65, v192, s73")
    e("s_add_u32 s73, s73, 1")
    e("s_add_u32 s78, s72, 33")                 # true number of valid bits (s72 has wrapped below zero)
    e("s_lshr_b64 s[64:65], s[64:65], s78")
    e("s_or_b64 s[62:63], s[62:63], s[64:65]")
    e("s_add_u32 s72, s72, 32")
    e(f"s_branch 4{lane:02d}b")
e("5:")
e("s_lshl_b32 s70, s73, 2")
e("s_add_u32 s71, s72, 33")
e("s_lshr_b32 s71, s71, 3")
e("s_sub_u32 %[po], s70, s71")
e("s_mov_b32 %[xo], s61")
e("s_waitcnt lgkmcnt(0)")
e(f"s_mov_b32 m0, {M0_SAVE}")

# ---- "dry" tile: the stream is exhausted (pos >= len), so the reference's refill loop reads nothing any more
# (src/rans.rs:365-368) and the state simply evolves: the same lookup and update, no window, no shift.
D = []
def d(s): D.append(s)
table_loads(d)
d("s_mov_b32 s61, %[xi]")
d("s_mov_b32 s74, %[nb]")
d("s_waitcnt vmcnt(0)")
d(".p2align 6")
d("s_nop 0")
d("2:")
for lane in range(64):
    d("s_bfe_u32 s76, s61, 0x60006")
    d("s_set_gpr_idx_on s76, 0x1")
    d("v_readlane_b32 s67, v64, s61")
    d("v_readlane_b32 s69, v128, s61")
    d(f"v_writelane_b32 v{REC}, s61, {lane}")
    d("s_mul_hi_u32 s70, s67, s61")
    d("s_add_u32 s61, s70, s69")
d("s_set_gpr_idx_off")
d(f"ds_write_b16 %[ra], v{REC}")
d("v_add_u32_e32 %[ra], 0x80, %[ra]")
d("s_sub_u32 s74, s74, 1")
d("s_cmp_lg_u32 s74, 0")
d("s_cbranch_scc1 2b")
d("s_mov_b32 %[xo], s61")
d("s_waitcnt lgkmcnt(0)")
d(f"s_mov_b32 m0, {M0_SAVE}")
dry_clob = ["memory", "scc", "s59", "s61", "s64", "s65", "s67", "s69", "s70", "s74", "s76"] + [f"v{r}" for r in range(64, 192)] + [f"v{REC}"]

clob = ["memory", "scc"] + [f"s{i}" for i in range(59, 79)] + [f"s{79 + c}" for c in range(21)]
clob += [f"v{r}" for r in range(64, REC + 1)]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rans_decode_tile.inc")
with open(out, "w") as f:
    f.write("// GENERATED by gen/gen_rans_decode_asm.py -- do not edit.\n")
    f.write("#define ALICE_DEC_TILE_ASM \\\n")
    for s in L:
        f.write(f'    "{s}\\n\\t" \\\n')
    f.write('    ""\n')
    f.write("#define ALICE_DEC_TILE_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + "\n")
    f.write("#define ALICE_DEC_DRY_TILE_ASM \\\n")
    for s in D:
        f.write(f'    "{s}\\n\\t" \\\n')
    f.write('    ""\n')
    f.write("#define ALICE_DEC_DRY_TILE_CLOBBERS " + ", ".join(f'"{c}"' for c in dry_clob) + "\n")
print("wrote", out, len(L), "asm lines")
