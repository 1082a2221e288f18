"""RV32IM completeness (format v15): mulh, mulhsu on the multiplier chip, div / divu / rem / remu on the divider chip.  The
committed guest executes none of them (round 3's tracer refused them); a hand-assembled guest does, on the spec's corner
cases.  The traced results are the spec's, the oracle proves the run, the product's verifier accepts it, and a wrong
quotient, remainder or high word - alone or with a matching chip row - is refused."""
import os

import numpy as np
import pytest

import toy_guest as tg

NQ, POW = 4, 4
CASES = [(7, 3), (-7, 3), (7, -3), (-7, -3), (123456789, 0), (-5, 0), (0, 0), (-(1 << 31), -1), (-(1 << 31), 1), (0, 5),
         ((1 << 31) - 1, 2), (0xFFFFFFFF, 0xFFFFFFFF), (0x80000000, 0x7FFFFFFF), (0x12345678, 0x9ABCDEF0), (1, -(1 << 31)),
         (0x10000, 0x10000), (0xFFFF0000, 0x0000FFFF)]
OPS = ["mul", "mulh", "mulhsu", "mulhu", "div", "divu", "rem", "remu"]


def program():
    ins = []
    for b, c in CASES:
        ins += tg.li(6, b) + tg.li(7, c)
        for k, op in enumerate(OPS):
            ins.append(tg.muldiv(op, 8 + (k & 1), 6, 7))  # results alternate between x8 and x9
    return ins + tg.epilogue()


@pytest.fixture(scope="module")
def toy(zk, built_lib, oracle):
    client = zk.ProverClient(device=-1, num_queries=NQ, pow_bits=POW)
    pk, vk = client.setup(tg.elf_of(program()))
    t = client.machine_trace(pk, zk.SP1Stdin())
    return client, pk, vk, t


def test_traced_results_follow_the_spec(zk, toy, oracle):
    client, pk, vk, t = toy
    cyc, prog = t["cycles"], t["program"]
    rows = prog[(cyc[:, 0] - prog[0, 0]) // 4]
    code_of = {27: "mul", 28: "mulhu", 31: "mulh", 32: "mulhsu", 33: "div", 34: "divu", 35: "rem", 36: "remu"}
    seen = {}
    for i in np.nonzero(np.isin(rows[:, 1], list(code_of)))[0]:
        name = code_of[int(rows[i, 1])]
        assert int(cyc[i, 1]) == tg.semantics(name, int(cyc[i, 2]), int(cyc[i, 3])), (name, hex(int(cyc[i, 2])), hex(int(cyc[i, 3])))
        seen[name] = seen.get(name, 0) + 1
    assert seen == {op: len(CASES) for op in OPS}
    assert t["info"].exit_code == 0 and t["public_values"] == b""
    # the divider chip's rows are the div / rem cycles, as the oracle derives them on its own; the multiplier chip also gets
    # the |q| * |d| products of the divisions by a non-zero divisor (low and high word each)
    div_cycles = np.nonzero(np.isin(rows[:, 1], [33, 34, 35, 36]))[0]
    assert np.array_equal(t["div_idx"], div_cycles) and np.array_equal(oracle.machine_events(t, 4), div_cycles)
    nz = int((cyc[div_cycles, 3] != 0).sum())
    assert len(t["muls"]) == 4 * len(CASES) + 2 * nz
    heights = oracle.machine_heights(t)
    assert heights[zk.MACHINE_CHIP_NAMES.index("divider")] == 7 and heights[zk.MACHINE_CHIP_NAMES.index("mul")] >= 7


def test_oracle_proof_of_the_toy_guest_verifies(zk, toy, oracle):
    client, pk, vk, t = toy
    proof = zk.SP1ProofWithPublicValues.from_bytes(oracle.machine_prove(t, num_queries=NQ, pow_bits=POW))
    client.verify(proof, vk)
    assert proof.public_values == b""
    # the committed guest's key is another program
    _, vk2 = client.setup(zk.merkle_elf())
    with pytest.raises(zk.ZkspError):
        client.verify(proof, vk2)


def _forced(oracle, t):
    os.environ["ZKSP_ORACLE_FORCE"] = "1"
    try:
        return oracle.machine_prove(t, num_queries=NQ, pow_bits=POW)
    finally:
        del os.environ["ZKSP_ORACLE_FORCE"]


@pytest.mark.parametrize("name", ["div", "remu", "mulh", "mulhsu"])
def test_wrong_results_are_refused(zk, toy, oracle, name):
    """The CPU row claims another result for one instruction (and writes it): the chip that computes the instruction from its
    operands sends another tuple - the ALU bus does not balance."""
    client, pk, vk, t = toy
    cyc, prog = t["cycles"], t["program"]
    rows = prog[(cyc[:, 0] - prog[0, 0]) // 4]
    code = {"mulh": 31, "mulhsu": 32, "div": 33, "remu": 36}[name]
    i = int(np.nonzero(rows[:, 1] == code)[0][3])
    c2 = cyc.copy()
    c2[i, 1] ^= 1
    t2 = dict(t, cycles=c2)
    with pytest.raises(RuntimeError):
        oracle.machine_prove(t2, num_queries=NQ, pow_bits=POW)
    with pytest.raises(zk.VerificationError):
        client.verify(zk.SP1ProofWithPublicValues.from_bytes(_forced(oracle, t2)), vk)


# ---- the divider chip row by row -------------------------------------------------------------------------------------------
P = 0x78000001
DV = dict(REAL=0, F=1, N=5, D=7, A=9, SN=11, SD=12, NH=13, DH=14, AN=15, AD=17, AQ=19, AR=21, CN=23, CD=24, CQ=25, CR=26, Q=27, R=29,
          SQ=31, SR=32, XS=33, PL=34, K=36, E=37, BE=39, NZD=40, INVD=41, NZQ=42, INVQ=43, NZR=44, INVR=45, W=46)
RANGED = [DV["A"], DV["A"] + 1, DV["AN"], DV["AN"] + 1, DV["AR"], DV["AR"] + 1, DV["E"], DV["E"] + 1]


def _inv(x):
    return pow(x % P, P - 2, P) if x % P else 0


def divider_row(op, n, d, aq, ar, sq=None, sr=None):
    """A divider-chip row for `op n, d` claiming |q| = aq, |r| = ar (the honest ones when they are the division's): every other
    column is what the chip's fill derives from them.  Limbs of values outside 0..2^32 are taken as signed field elements,
    the best a forger can do."""
    sg = op in ("div", "rem")
    sn, sd = (n >> 31 if sg else 0), (d >> 31 if sg else 0)
    an, ad = ((-n) & tg.M32 if sn else n), ((-d) & tg.M32 if sd else d)
    if sq is None:
        sq = 1 if d == 0 else (sn ^ sd) & (aq != 0)
    if sr is None:
        sr = sn & (ar != 0)
    q, rm = ((-aq) & tg.M32 if sq else aq), ((-ar) & tg.M32 if sr else ar)
    if d == 0:
        q = tg.M32
    row = [0] * DV["W"]

    def limbs(col, v):
        if 0 <= v < (1 << 32):
            row[col], row[col + 1] = v & 0xFFFF, v >> 16
        else:  # out of range: low limb in range, the rest (possibly negative) in the high one
            row[col], row[col + 1] = v & 0xFFFF, (v >> 16) % P

    row[DV["REAL"]] = 1
    row[DV["F"] + ["div", "divu", "rem", "remu"].index(op)] = 1
    limbs(DV["N"], n); limbs(DV["D"], d); limbs(DV["A"], q if op in ("div", "divu") else rm)
    row[DV["SN"]], row[DV["SD"]] = sn, sd
    row[DV["NH"]], row[DV["DH"]] = (n >> 16) - 32768 * sn, (d >> 16) - 32768 * sd
    limbs(DV["AN"], an); limbs(DV["AD"], ad); limbs(DV["AQ"], aq); limbs(DV["AR"], ar)
    row[DV["CN"]], row[DV["CD"]] = int(sn and (n & 0xFFFF) != 0), int(sd and (d & 0xFFFF) != 0)
    row[DV["CQ"]], row[DV["CR"]] = int(sq and (q & 0xFFFF) != 0), int(sr and (rm & 0xFFFF) != 0)
    limbs(DV["Q"], q); limbs(DV["R"], rm)
    row[DV["SQ"]], row[DV["SR"]], row[DV["XS"]] = sq, sr, sn ^ sd
    pl = 0 if d == 0 else (aq * ad) & tg.M32
    limbs(DV["PL"], pl)
    row[DV["K"]] = int(d != 0 and (pl & 0xFFFF) + (ar & 0xFFFF) > 0xFFFF)
    limbs(DV["E"], 0 if d == 0 else ad - ar - 1)
    row[DV["BE"]] = int(d != 0 and (ad & 0xFFFF) < (ar & 0xFFFF) + 1)
    for nz, iv, v in ((DV["NZD"], DV["INVD"], d), (DV["NZQ"], DV["INVQ"], aq), (DV["NZR"], DV["INVR"], ar)):
        row[nz], row[iv] = int(v != 0), _inv((v & 0xFFFF) + (v >> 16))
    return [x % P for x in row]


def _violations(oracle, zk, row):
    chip = zk.MACHINE_CHIP_NAMES.index("divider")
    c = oracle.machine_constraints(chip, None, np.array(row, np.uint32), np.zeros(DV["W"], np.uint32), 1, 0, 1)
    out_of_range = [k for k in RANGED if row[k] >= 1 << 16]
    # the multiplier chip must be able to produce |q| * |d|: its operands are bit-decomposed 32-bit words and the high word
    # the divider claims is zero
    aq = row[DV["AQ"]] + (row[DV["AQ"] + 1] << 16)
    ad = row[DV["AD"]] + (row[DV["AD"] + 1] << 16)
    if row[DV["NZD"]] and (row[DV["AQ"] + 1] >= 1 << 16 or aq * ad >= 1 << 32):
        out_of_range.append(DV["AQ"])
    return int(np.count_nonzero(c)), out_of_range


def test_divider_rows_admit_only_the_quotient_and_remainder(zk, toy, oracle):
    client, pk, vk, t = toy
    chip = zk.MACHINE_CHIP_NAMES.index("divider")
    _, main = oracle.machine_fill(t, chip)
    cyc, prog = t["cycles"], t["program"]
    rows = prog[(cyc[:, 0] - prog[0, 0]) // 4]
    names = {33: "div", 34: "divu", 35: "rem", 36: "remu"}
    for r, i in enumerate(t["div_idx"]):
        op, n, d = names[int(rows[i, 1])], int(cyc[i, 2]), int(cyc[i, 3])
        sg = op in ("div", "rem")
        an = (-n) & tg.M32 if sg and n >> 31 else n
        ad = (-d) & tg.M32 if sg and d >> 31 else d
        aq, ar = (an // ad, an % ad) if d else (1, an)
        honest = divider_row(op, n, d, aq, ar)
        # the builder above is the oracle's fill, and the fill satisfies the chip
        assert honest == [int(x) for x in main[:, r]], (op, hex(n), hex(d))
        assert _violations(oracle, zk, honest) == (0, [])
        forged = []
        if d:
            forged += [(aq + 1, ar - ad), (aq - 1, ar + ad), (aq + 1, ar), (aq, ar + 1), (0, an), (aq, (ar + ad) & tg.M32)]
            forged += [(aq, ar, 1 - honest[DV["SQ"]], None), (aq, ar, None, 1 - honest[DV["SR"]])]
            if aq:
                # wrap-around: |q'| |d| + r' = |n| + 2^32
                forged += [((an + (1 << 32)) // ad, (an + (1 << 32)) % ad)]
        else:
            forged += [(0, an), (1, 0), (1, (an + 1) & tg.M32), (2, an), (1, an, 0, None)]
        for f in forged:
            if f[0] < 0 or tuple(f) == (aq, ar):
                continue
            nviol, oor = _violations(oracle, zk, divider_row(op, n, d, *f))
            assert nviol or oor, (op, hex(n), hex(d), f)


def test_multiplier_rows_admit_one_signed_high_word(zk, toy, oracle):
    """mulh / mulhsu rows: R = P_hi - b31 C - [mulh] c31 B (mod 2^32) with two carries of two bits each.  For every other choice
    of the carries the limbs of R that satisfy the chip's equations leave the 16-bit range its lookups enforce, and any other R
    with the honest carries violates the equations."""
    client, pk, vk, t = toy
    chip = zk.MACHINE_CHIP_NAMES.index("mul")
    MU_SH, MU_SHU, MU_R, MU_K0, MU_K1, W = 161, 162, 163, 165, 167, 169
    _, main = oracle.machine_fill(t, chip)
    zero = np.zeros(W, np.uint32)
    seen = 0
    for r in range(main.shape[1]):
        row = main[:, r].copy()
        if not (row[MU_SH] or row[MU_SHU]):
            continue
        seen += 1
        assert not oracle.machine_constraints(chip, None, row, zero, 1, 0, 1).any()
        b = sum(int(row[2 + i]) << i for i in range(32))
        c = sum(int(row[34 + i]) << i for i in range(32))
        want = tg.semantics("mulh" if row[MU_SH] else "mulhsu", b, c)
        assert int(row[MU_R]) + (int(row[MU_R + 1]) << 16) == want
        k0, k1 = int(row[MU_K0] + row[MU_K0 + 1]), int(row[MU_K1] + row[MU_K1 + 1])
        for a0 in range(3):
            for a1 in range(3):
                if (a0, a1) == (k0, k1):
                    continue
                f = row.copy()
                f[MU_K0], f[MU_K0 + 1] = int(a0 > 0), int(a0 > 1)
                f[MU_K1], f[MU_K1 + 1] = int(a1 > 0), int(a1 > 1)
                lo = (int(row[MU_R]) + 65536 * (a0 - k0)) % P
                hi = (int(row[MU_R + 1]) - (a0 - k0) + 65536 * (a1 - k1)) % P
                f[MU_R], f[MU_R + 1] = lo, hi
                assert not oracle.machine_constraints(chip, None, f, zero, 1, 0, 1).any()  # the equations alone allow it
                assert lo >= 1 << 16 or hi >= 1 << 16  # the range lookups of R do not
        for col in (MU_R, MU_R + 1):
            f = row.copy()
            f[col] = (int(f[col]) + 1) % P
            assert oracle.machine_constraints(chip, None, f, zero, 1, 0, 1).any()
        f = row.copy()  # a row cannot be mulh and mulhsu (or mulhu) at once
        f[MU_SH] = f[MU_SHU] = 1
        assert oracle.machine_constraints(chip, None, f, zero, 1, 0, 1).any()
    assert seen == 2 * len(CASES)
